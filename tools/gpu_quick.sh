#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py tests/test_dp_gpu.py -m gpu -q -x -k "engine or head or layernorm or bench" > gpurun_out/r3_quick_tests.log 2>&1; rc=$?
tail -4 gpurun_out/r3_quick_tests.log
[ $rc -eq 124 ] && exit 124
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('default', d['ms_per_step'], d['value'])"
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --graph --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('graph  ', d['ms_per_step'], d['value'])"
done
