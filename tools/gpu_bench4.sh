#!/bin/bash
# the four bench lines (tiny, tiny MPP, config 3, config 5); tag = $1
cd "$GRAFT_REPO_ROOT"
tag=${1:-x}
timeout -k 10 300 python bench.py --steps 40 --warmup 5 > gpurun_out/r4_bench_tiny_$tag.json 2> gpurun_out/r4_bench_tiny_$tag.err || tail -3 gpurun_out/r4_bench_tiny_$tag.err
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --dtype f16 --no-cpu-baseline --no-probe > gpurun_out/r4_bench_tiny_f16_$tag.json 2>&1
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --graph --no-cpu-baseline --no-probe > gpurun_out/r4_bench_tiny_graph_$tag.json 2>&1
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --task mpp --no-cpu-baseline --no-probe > gpurun_out/r4_bench_tiny_mpp_$tag.json 2>&1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --model small --patches 1280 --batch 32 --no-cpu-baseline --no-probe > gpurun_out/r4_bench_cfg3_$tag.json 2>&1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --model base --patches 1280 --batch 32 --task mpp --no-cpu-baseline --no-probe > gpurun_out/r4_bench_cfg5_$tag.json 2>&1
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for f in ("tiny", "tiny_f16", "tiny_graph", "tiny_mpp", "cfg3", "cfg5"):
    try:
        l = [x for x in open(f"gpurun_out/r4_bench_{f}_{tag}.json") if x.startswith("{")][0]
        d = json.loads(l)
        print(f"{f:11s} {d['ms_per_step']:8.3f} ms  {d['value']:9.1f} surfaces/s  step MFMA {d['step_mfma_frac']:.4f}  graph {d['config']['hip_graph']} overlap {d['config']['wgrad_overlap_layers']}")
    except Exception as e:
        print(f, "ERR", e)
PY
