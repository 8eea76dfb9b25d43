#!/bin/bash
cd "$GRAFT_REPO_ROOT"
bash tools/gpu_profile.sh r3w_cfg5 --model base --patches 1280 --batch 32 --task mpp | head -12
bash tools/gpu_pmc2.sh r3w_wgrad_cfg5 "wgrad_wide" bench.py --model base --patches 1280 --batch 32 --task mpp --steps 3 --warmup 1 --no-cpu-baseline --no-probe --no-graph > /dev/null
cat gpurun_out/pmc_r3w_wgrad_cfg5.txt
