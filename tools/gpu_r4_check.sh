#!/bin/bash
# round 4: whole GPU suite in one process, then the default bench line three times and a kernel table of the step
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r4_tests_all.log 2>&1; rc=$?
tail -8 gpurun_out/r4_tests_all.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('default', d['ms_per_step'], d['value'])" || exit 1
done
bash tools/gpu_profile.sh ${1:-r4_check}
