#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
: > gpurun_out/r5_dp_budget_k.txt
for rep in 1 2; do
  for cfg in 8:2:42:16 6:2:42:16 8:2:42:1; do
    timeout -k 10 200 python tools/dp_cu_budget.py --configs "$cfg" --steps 40 --own-stream 0 2>&1 | grep "ms per step" | cut -c1-150 >> gpurun_out/r5_dp_budget_k.txt || exit 1
  done
  echo "16 workgroups, 1 us each" >> gpurun_out/r5_dp_budget_k.txt
  timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --standin-us 1 --steps 40 --own-stream 0 2>&1 | grep "ms per step" | cut -c1-150 >> gpurun_out/r5_dp_budget_k.txt || exit 1
done
cat gpurun_out/r5_dp_budget_k.txt
