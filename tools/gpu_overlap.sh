#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -k "engine or wgrad" > gpurun_out/r3_overlap_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r3_overlap_tests.log
[ $rc -eq 124 ] && exit 124
for ov in 0 3 6 8 9 10 12; do
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --wgrad-overlap $ov --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('overlap $ov', d['ms_per_step'], d['value'], d['config']['loss_after'])" || exit 1
  done
done
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --wgrad-overlap 9 --no-graph --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('overlap 9 eager', d['ms_per_step'], d['value'])"
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --wgrad-overlap 0 --no-graph --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('overlap 0 eager', d['ms_per_step'], d['value'])"
