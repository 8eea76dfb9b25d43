cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_mlp_fused_gpu.py tests/test_ln_gemm_fused_gpu.py tests/test_kernels_gpu.py -m gpu -q -x > gpurun_out/r5_tests_rows.log 2>&1; rc=$?
tail -3 gpurun_out/r5_tests_rows.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|FAILED" gpurun_out/r5_tests_rows.log | head; exit $rc; }
for b in 64 128 192 256 512; do
  timeout -k 10 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-also 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('B', d['config']['global_batch'], d['ms_per_step'], d['value'], d['step_mfma_frac'], d['config']['hip_graph'], d['config']['wgrad_overlap_layers'])"
done
