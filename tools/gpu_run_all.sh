#!/bin/bash
# full check on the GPU box: GPU tests (SITK_PARITY_RECORD=1 to record bf16 errors), bench lines of every BASELINE configuration
cd "$GRAFT_REPO_ROOT"
tag=${1:-x}
step() { "$@"; rc=$?; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "STEP KILLED ($rc): $*"; exit $rc; fi; return 0; }
step timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_tests_$tag.log 2>&1
tail -4 gpurun_out/r3_tests_$tag.log
step timeout -k 10 300 python bench.py --steps 30 --warmup 5 > gpurun_out/r3_bench_tiny_$tag.json 2> gpurun_out/r3_bench_tiny_$tag.err
step timeout -k 10 200 python bench.py --steps 20 --warmup 3 --task mpp --no-cpu-baseline --no-probe > gpurun_out/r3_bench_tiny_mpp_$tag.json 2>&1
step timeout -k 10 300 python bench.py --steps 10 --warmup 3 --model small --patches 1280 --batch 32 --no-cpu-baseline --no-probe > gpurun_out/r3_bench_cfg3_$tag.json 2>&1
step timeout -k 10 300 python bench.py --steps 10 --warmup 3 --model base --patches 1280 --batch 32 --task mpp --no-cpu-baseline --no-probe > gpurun_out/r3_bench_cfg5_$tag.json 2>&1
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("tiny", "tiny_mpp", "cfg3", "cfg5"):
    try:
        l = [x for x in open(f"gpurun_out/r3_bench_{f}_{tag}.json") if x.startswith("{")][0]
        d = json.loads(l)
        print(f"{f:9s} {d['ms_per_step']:8.3f} ms  {d['value']:9.1f} surfaces/s  step MFMA {d['step_mfma_frac']:.4f}")
    except Exception as e:
        print(f, "ERR", e)
PY
