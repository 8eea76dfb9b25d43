#!/bin/bash
# counters of single encoder GEMMs (config 3 shapes): bash tools/gpu_gemm_pmc.sh
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python tools/gemm_bench.py --model small > gpurun_out/gemm_small.txt 2>&1 || { tail -5 gpurun_out/gemm_small.txt; exit 1; }
cat gpurun_out/gemm_small.txt
for i in 0 3; do
  sed -i 's/"FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"/"FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"/' tools/gpu_pmc2.sh
  bash tools/gpu_pmc2.sh gemm_small_$i "n192" tools/gemm_bench.py --model small --only $i --reps 4 > /dev/null
  echo "== row $i"; cat gpurun_out/pmc_gemm_small_$i.txt
done
