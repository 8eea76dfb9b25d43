#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
: > gpurun_out/r5_dp_budget_l.txt
for rep in 1 2; do
  for v in "0 16" "1 1" "1 16" "70 16"; do
    set -- $v
    echo "stand-in $1 us x $2 workgroups, on the bucket's stream" >> gpurun_out/r5_dp_budget_l.txt
    timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:$2" --standin-us $1 --steps 40 --own-stream 0 2>&1 | grep "ms per step" | cut -c1-150 >> gpurun_out/r5_dp_budget_l.txt || exit 1
  done
done
cat gpurun_out/r5_dp_budget_l.txt
