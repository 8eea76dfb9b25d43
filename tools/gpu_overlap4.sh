#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for ov in 0 6 7 8; do
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --wgrad-overlap $ov --no-graph --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('eager overlap $ov', d['ms_per_step'], d['value'])" || exit 1
  done
done
