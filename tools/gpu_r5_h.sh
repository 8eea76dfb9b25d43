#!/bin/bash
# round 5: all-reduce stand-in, ONE configuration per process (engines created later in a process measured up to 2 ms slower:
# streams and queues of the earlier ones are still alive), three repetitions, default priority
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
: > gpurun_out/r5_dp_budget_h.txt
for rep in 1 2 3; do
  for cfg in ${1:-8:2:42:16 7:2:42:16 6:2:42:16 6:2:56:16 6:2:42:32}; do
    timeout -k 10 200 python tools/dp_cu_budget.py --configs "$cfg" --steps 40 --prio ${PRIO:-0} 2>&1 | grep "ms per step" | cut -c1-150 >> gpurun_out/r5_dp_budget_h.txt || exit 1
  done
done
cat gpurun_out/r5_dp_budget_h.txt
