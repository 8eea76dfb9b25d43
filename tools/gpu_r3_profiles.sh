#!/bin/bash
# round-3 profile set: kernel tables of the tiny step (default form: side stream; and --graph), tiny f16, tiny MPP, config 3 / 5;
# PMC passes of the dominant kernel (block-tail forward) and of the fused MLP backward, attention dq / dkv
cd "$GRAFT_REPO_ROOT"
bash tools/gpu_profile.sh r3_tiny > /dev/null 2>&1; head -16 gpurun_out/prof_r3_tiny.md
bash tools/gpu_profile.sh r3_tiny_f16 --dtype f16 > /dev/null 2>&1
bash tools/gpu_profile.sh r3_tiny_nooverlap --wgrad-overlap 0 > /dev/null 2>&1
bash tools/gpu_profile.sh r3_tiny_mpp --task mpp > /dev/null 2>&1
bash tools/gpu_profile.sh r3_cfg3 --model small --patches 1280 --batch 32 > /dev/null 2>&1
bash tools/gpu_profile.sh r3_cfg5 --model base --patches 1280 --batch 32 --task mpp > /dev/null 2>&1
bash tools/gpu_pmc2.sh r3_block_tail_fwd "mlp_kernel<false, 0, 6, true, true" tools/kbench.py proj_mlp_next_fwd --reps 5 > /dev/null 2>&1
bash tools/gpu_pmc2.sh r3_mlp_bwd "mlp_kernel<true" tools/kbench.py mlp_bwd --reps 5 > /dev/null 2>&1
ls gpurun_out/ | grep -E "r3_|pmc_r3"
