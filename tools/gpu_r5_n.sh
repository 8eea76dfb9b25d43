#!/bin/bash
# round 5: kernel tables of the default step with the classic / Q-resident plan of the merged attention backward (diagnostic build),
# and the data-parallel tests + bench lines
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
for v in 0 1; do
  SITK_ATTN_QRES=$v bash tools/gpu_profile.sh r5_qres$v --no-also > gpurun_out/r5_prof_qres$v.log 2>&1 || { tail -5 gpurun_out/r5_prof_qres$v.log; exit 1; }
  grep -E "attn_bwd_res|attn_fwd_res|ln_gemm_mlp_bwd|mlp_kernel" gpurun_out/prof_r5_qres$v.md | head -8
done
unset SITK_LIB
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x > gpurun_out/r5_tests_n.log 2>&1; rc=$?
tail -4 gpurun_out/r5_tests_n.log
[ $rc -ne 0 ] && { grep -n "Error\|error" gpurun_out/r5_tests_n.log | head -20; exit $rc; }
for rep in 1 2; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --no-also --dp-form | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('dp-form', d['ms_per_step'])" || exit 1
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --no-also | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('default', d['ms_per_step'])" || exit 1
done
