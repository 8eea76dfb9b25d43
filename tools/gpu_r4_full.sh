#!/bin/bash
# round 4: whole GPU suite in one process, smoke, then the driver's bench command (with the `also` object)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/r4_tests_all.log 2>&1; rc=$?
tail -6 gpurun_out/r4_tests_all.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
t0=$(date +%s)
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_driver_${1:-x}.json 2> gpurun_out/r4_bench_driver_${1:-x}.err; rc=$?
echo "bench wall: $(( $(date +%s) - t0 )) s, rc $rc"
python - "$1" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r4_bench_driver_{sys.argv[1] or 'x'}.json") if l.startswith("{")][0])
print("headline", d["ms_per_step"], d["value"], d["step_mfma_frac"])
for k, v in d.get("also", {}).items():
    print("  also", k, v)
PY
exit $rc
