#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('single default', d['ms_per_step'], d['value'])"
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --dp-form --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('dp-form auto', d['ms_per_step'], d['value'], d['config']['hip_graph'], d['config']['wgrad_overlap_layers'])"
done
bash tools/gpu_dpside_trace.sh
