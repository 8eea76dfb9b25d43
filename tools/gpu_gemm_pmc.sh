#!/bin/bash
# counters of single encoder GEMMs (config 3 shapes): bash tools/gpu_gemm_pmc.sh
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python tools/gemm_bench.py --model small > gpurun_out/gemm_small.txt 2>&1 || { tail -5 gpurun_out/gemm_small.txt; exit 1; }
cat gpurun_out/gemm_small.txt
for i in 0 3; do
  bash tools/gpu_pmc2.sh gemm_small_$i "n192" tools/gemm_bench.py --model small --only $i --reps 4 > /dev/null
  echo "== row $i"; cat gpurun_out/pmc_gemm_small_$i.txt
done
