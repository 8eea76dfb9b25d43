#!/bin/bash
# kernel-level A/B of two library builds: bash tools/gpu_lib_kbench.sh <libA> <libB> <kbench kernel> ...
cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; shift 2
for k in "$@"; do for rep in 1 2 3; do for lib in $A $B; do
  echo -n "$lib  "; SITK_LIB=$PWD/$lib timeout -k 10 100 python tools/kbench.py $k 2>&1 | tail -1
done; done; done
