#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "wgrad or bitwise or bench_config" > gpurun_out/r3_wgrad_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r3_wgrad_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('default', d['ms_per_step'], d['value'])"
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --graph --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('graph  ', d['ms_per_step'], d['value'])"
done
