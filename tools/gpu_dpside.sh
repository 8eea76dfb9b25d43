#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x > gpurun_out/r3_dpside_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_dpside_tests.log
[ $rc -eq 124 ] && exit 124
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --dp-form --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('dp-form auto', d['ms_per_step'], d['value'], d['config']['hip_graph'], d['config']['wgrad_overlap_layers'])"
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --dp-form --graph --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('dp-form graph', d['ms_per_step'], d['value'], d['config']['hip_graph'], d['config']['wgrad_overlap_layers'])"
done
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>&1 | tail -1 | cut -c1-300
