#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x > gpurun_out/r5_tests_m.log 2>&1; rc=$?
tail -4 gpurun_out/r5_tests_m.log
[ $rc -ne 0 ] && { grep -n "Error\|error" gpurun_out/r5_tests_m.log | head -20; exit $rc; }
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
: > gpurun_out/r5_dp_budget_m.txt
for rep in 1 2; do
  for pb in 1 2 4; do
    for ch in 16 8; do
      echo "side launches per bucket $pb, stand-in on the bucket's stream" >> gpurun_out/r5_dp_budget_m.txt
      timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:$ch" --per-bucket $pb --steps 40 --own-stream 0 2>&1 | grep "ms per step" | cut -c1-190 >> gpurun_out/r5_dp_budget_m.txt || exit 1
    done
  done
done
cat gpurun_out/r5_dp_budget_m.txt
