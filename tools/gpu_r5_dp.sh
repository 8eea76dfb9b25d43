#!/bin/bash
# round 5: the whole GPU suite (recording run for the new bf16 cases), the default bench line, the data-parallel form on a one-rank
# RCCL group, and the all-reduce stand-in sweep (tools/dp_cu_budget.py, diagnostic build)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SITK_PARITY_RECORD=1 timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r5_tests_all.log 2>&1; rc=$?
tail -15 gpurun_out/r5_tests_all.log
[ $rc -eq 124 ] && exit 124
[ $rc -eq 137 ] && exit 137
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --no-also > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err || { tail -5 gpurun_out/r5_bench_default.err; exit 1; }
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-probe --no-also --dp-form > gpurun_out/r5_bench_dpform.json 2> gpurun_out/r5_bench_dpform.err || { tail -5 gpurun_out/r5_bench_dpform.err; exit 1; }
python - <<'PY'
import json
for f in ("default", "dpform"):
    d = json.loads([l for l in open(f"gpurun_out/r5_bench_{f}.json") if l.startswith("{")][0])
    print(f, d["ms_per_step"], d["value"], d["config"]["parallelism"])
PY
SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so timeout -k 10 400 python tools/dp_cu_budget.py --channels ${DP_CH:-8,16} --side-cus ${DP_SIDE:-42,34,26} > gpurun_out/r5_dp_budget_raw.txt 2> gpurun_out/r5_dp_budget.err || { tail -5 gpurun_out/r5_dp_budget.err; exit 1; }
cat gpurun_out/r5_dp_budget_raw.txt
