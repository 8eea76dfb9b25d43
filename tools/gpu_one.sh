#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
echo "--- new"; timeout -k 10 200 python tools/attn_bench.py --batch 64 --tokens 321 --heads 3 2>/dev/null
echo "--- old"; SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_old.so timeout -k 10 200 python tools/attn_bench.py --batch 64 --tokens 321 --heads 3 2>/dev/null
done
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('new', d['ms_per_step'], d['value'])"
SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_old.so timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('old', d['ms_per_step'], d['value'])"
done
