#!/bin/bash
# A/B of the 12-wave (16-token waves) variants of the fused kernels: SITK_MLP_TT1=1, SITK_LG_TT1=1, SITK_MLP_PRIO
cd "$GRAFT_REPO_ROOT"
SITK_MLP_TT1=1 SITK_LG_TT1=1 timeout -k 10 600 python -m pytest tests/test_mlp_fused_gpu.py tests/test_ln_gemm_fused_gpu.py -m gpu -x -q > gpurun_out/tt1_tests.log 2>&1; rc=$?
tail -5 gpurun_out/tt1_tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
for v in 0 1 0 1; do
  echo -n "LG_TT1=$v "; SITK_LG_TT1=$v timeout -k 10 120 python tools/kbench.py lnqkv_bwd --reps 20 2>/dev/null || exit 1
done
for k in mlp_bwd proj_mlp_next_fwd; do
  for v in 0 1 2 3; do
    echo -n "PRIO=$v "; SITK_MLP_TT1=1 SITK_MLP_PRIO=$v timeout -k 10 120 python tools/kbench.py $k --reps 20 2>/dev/null || exit 1
  done
done
for v in "0 0 0" "1 0 0" "1 1 0" "1 1 1" "1 1 2" "1 1 3"; do
  set -- $v
  echo "bench MLP_TT1=$1 LG_TT1=$2 PRIO=$3"; SITK_MLP_TT1=$1 SITK_LG_TT1=$2 SITK_MLP_PRIO=$3 timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1
done
