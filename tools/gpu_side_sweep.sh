#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
for grp in 2 3 4; do
for ov in 7 8 9 10 12; do
export SITK_SIDE_GROUP=$grp
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --wgrad-overlap $ov 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('group $grp overlap $ov', d['ms_per_step'], d['value'])"
done
done
