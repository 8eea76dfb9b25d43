#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "wgrad" > gpurun_out/r3_wgrad_tests.log 2>&1; rc=$?
tail -15 gpurun_out/r3_wgrad_tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
timeout -k 10 200 python bench.py --model small --patches 1280 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('cfg3', d['ms_per_step'], d['value'])"
timeout -k 10 200 python bench.py --model base --patches 1280 --batch 32 --task mpp --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('cfg5', d['ms_per_step'], d['value'])"
done
