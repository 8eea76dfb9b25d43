cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_models_gpu.py tests/test_dp_gpu.py -m gpu -q -x > gpurun_out/r4_small_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4_small_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-also --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('default', d['ms_per_step'], d['value'])" || exit 1
done
