#!/bin/bash
# round 5: is the stand-in's cost a matter of CUs or of scheduling?  1 channel (one workgroup) for the full wire time; 16 channels for 1 us
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
: > gpurun_out/r5_dp_budget_i.txt
for rep in 1 2; do
  echo "one workgroup, full wire time" >> gpurun_out/r5_dp_budget_i.txt
  timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:1" --steps 40 2>&1 | grep "ms per step" | cut -c1-150 >> gpurun_out/r5_dp_budget_i.txt || exit 1
  echo "16 workgroups, 1 us each" >> gpurun_out/r5_dp_budget_i.txt
  timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --standin-us 1 --steps 40 2>&1 | grep "ms per step" | cut -c1-150 >> gpurun_out/r5_dp_budget_i.txt || exit 1
  echo "2 workgroups, full wire time" >> gpurun_out/r5_dp_budget_i.txt
  timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:2" --steps 40 2>&1 | grep "ms per step" | cut -c1-150 >> gpurun_out/r5_dp_budget_i.txt || exit 1
done
cat gpurun_out/r5_dp_budget_i.txt
