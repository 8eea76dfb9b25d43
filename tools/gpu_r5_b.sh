#!/bin/bash
# round 5, second call: targeted tests (Q-resident attention backward, the data-parallel tests that failed), step A/B of the
# Q-resident plan (diagnostic build, SITK_ATTN_QRES=0/1), the all-reduce stand-in over the CU-budget options
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_dp_gpu.py -m gpu -q -x -k "attention or bench_ or side_stream or rccl" > gpurun_out/r5_tests_b.log 2>&1; rc=$?
tail -6 gpurun_out/r5_tests_b.log
[ $rc -ne 0 ] && { grep -n "Error\|error" gpurun_out/r5_tests_b.log | head -20; exit $rc; }
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "SITK_ATTN_QRES=$v: "; SITK_ATTN_QRES=$v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --no-also 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(d['ms_per_step'])" || exit 1
  done
done
timeout -k 10 500 python tools/dp_cu_budget.py --configs "8:2:42:16,8:2:42:32,9:3:30:12,9:3:30:8,8:3:30:12,6:2:20:16" > gpurun_out/r5_dp_budget_b.txt 2> gpurun_out/r5_dp_budget_b.err || { tail -5 gpurun_out/r5_dp_budget_b.err; exit 1; }
cat gpurun_out/r5_dp_budget_b.txt
echo "--- SITK_BWD_ROWS128=1 (128-row backward workgroups: 161 instead of 214; no pair launch)"
SITK_BWD_ROWS128=1 timeout -k 10 400 python tools/dp_cu_budget.py --configs "8:2:42:16,8:2:42:32,8:2:56:32" > gpurun_out/r5_dp_budget_b128.txt 2> gpurun_out/r5_dp_budget_b128.err || { tail -5 gpurun_out/r5_dp_budget_b128.err; exit 1; }
cat gpurun_out/r5_dp_budget_b128.txt
