#!/bin/bash
# all-reduce stand-in sweep (tools/dp_cu_budget.py), ONE configuration per process (engines created later in a process measured up
# to 2 ms slower: streams and hardware queues of the earlier ones are still alive), three repetitions.
# usage: bash tools/gpu_dp_sweep.sh "8:2:42:16 6:2:42:16" [extra dp_cu_budget.py arguments, e.g. --own-stream 0 --per-bucket 2]
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
cfgs=${1:-8:2:42:16}; shift
: > gpurun_out/dp_sweep.txt
for rep in 1 2 3; do
  for cfg in $cfgs; do
    timeout -k 10 200 python tools/dp_cu_budget.py --configs "$cfg" --steps 40 --own-stream 0 "$@" 2>&1 | grep "ms per step" | cut -c1-190 >> gpurun_out/dp_sweep.txt || exit 1
  done
done
cat gpurun_out/dp_sweep.txt
