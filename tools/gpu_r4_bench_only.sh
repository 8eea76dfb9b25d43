#!/bin/bash
cd "$GRAFT_REPO_ROOT"
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
