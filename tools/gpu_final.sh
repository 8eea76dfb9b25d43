#!/bin/bash
# end-of-round collection on the GPU box: full check + kernel tables + PMC of the dominant kernels
cd "$GRAFT_REPO_ROOT"
tag=${1:-final}
bash tools/gpu_run_all.sh $tag || exit 1
bash tools/gpu_profile.sh r2_tiny_$tag > /dev/null || exit 1
bash tools/gpu_profile.sh r2_mpp_tiny_$tag --task mpp > /dev/null || exit 1
bash tools/gpu_profile.sh r2_cfg3_$tag --model small --patches 1280 --batch 32 > /dev/null || exit 1
bash tools/gpu_pmc2.sh block_tail_12w "mlp_kernel<false" tools/kbench.py proj_mlp_next_fwd --reps 5 > /dev/null 2>&1
bash tools/gpu_pmc2.sh mlp_bwd_12w "mlp_kernel<true" tools/kbench.py mlp_bwd --reps 5 > /dev/null 2>&1
bash tools/gpu_pmc2.sh attn_fwd_res "attn_fwd_res" tools/kbench.py attn_fwd --reps 5 > /dev/null 2>&1
echo DONE
