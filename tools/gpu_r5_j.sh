#!/bin/bash
# round 5: the step on a HIGH-priority stream, the all-reduce path at normal priority, the side stream lowest
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
: > gpurun_out/r5_dp_budget_j.txt
for rep in 1 2; do
  for cfg in 8:2:42:16 8:2:42:1 6:2:42:16; do
    timeout -k 10 200 python tools/dp_cu_budget.py --configs "$cfg" --steps 40 --main-prio -1 2>&1 | grep "ms per step" | cut -c1-150 >> gpurun_out/r5_dp_budget_j.txt || exit 1
  done
  echo "16 workgroups, 1 us each" >> gpurun_out/r5_dp_budget_j.txt
  timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --standin-us 1 --steps 40 --main-prio -1 2>&1 | grep "ms per step" | cut -c1-150 >> gpurun_out/r5_dp_budget_j.txt || exit 1
done
cat gpurun_out/r5_dp_budget_j.txt
