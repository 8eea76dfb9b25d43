#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3 4; do
timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('prefetch   ', d['ms_per_step'], d['value'])"
timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probe --no-prefetch 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('no prefetch', d['ms_per_step'], d['value'])"
done
