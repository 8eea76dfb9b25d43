#!/bin/bash
# step-time A/B of two builds of the library: bash tools/gpu_lib_ab.sh <libA> <libB> [bench args]
cd "$GRAFT_REPO_ROOT"
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for lib in $A $B; do
    r=$(SITK_LIB=$PWD/$lib timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-also --no-probe --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    echo "$lib: $r ms"
  done
done
