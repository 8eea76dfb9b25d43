cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r5_tests_final2.log 2>&1; rc=$?
tail -3 gpurun_out/r5_tests_final2.log
for b in 128 256 512; do
  timeout -k 10 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-also 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('B', d['config']['global_batch'], d['ms_per_step'], d['value'], d['step_mfma_frac'], d['config']['hip_graph'], d['config']['wgrad_overlap_layers'])"
done
