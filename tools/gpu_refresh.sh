#!/bin/bash
# refresh of the round-3 artifacts after the double-tile weight-gradient kernel: profile set, bench lines, timeline
cd "$GRAFT_REPO_ROOT"
bash tools/gpu_r3_profiles.sh > gpurun_out/r3_profiles.log 2>&1
bash tools/gpu_bench4.sh final
bash tools/gpu_full_timeline.sh > /dev/null 2>&1
bash tools/gpu_pmc2.sh r3_wgrad_cfg5 "wgrad_x2" bench.py --model base --patches 1280 --batch 32 --task mpp --steps 3 --warmup 1 --no-cpu-baseline --no-probe --no-graph > /dev/null 2>&1
bash tools/gpu_pmc2.sh r3_wgrad_tiny "wgrad_x2" bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe --graph > /dev/null 2>&1
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --dp-form --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('dp-form', d['ms_per_step'], d['value'])"
timeout -k 10 200 python tools/gemm_bench.py --model small > gpurun_out/r3_gemm_bench_small.txt 2>/dev/null
timeout -k 10 200 python tools/gemm_bench.py --model base > gpurun_out/r3_gemm_bench_base.txt 2>/dev/null
tail -9 gpurun_out/r3_gemm_bench_small.txt
