#!/bin/bash
# round-4 profile set: kernel tables (tiny default form, tiny f16, configs 3 / 5), counters of the dominant kernel (block-tail
# forward), of the fused MLP backward and of the resident attention kernels, one step as a timeline
cd "$GRAFT_REPO_ROOT"
bash tools/gpu_profile.sh r4_tiny > /dev/null 2>&1; head -14 gpurun_out/prof_r4_tiny.md
bash tools/gpu_profile.sh r4_tiny_f16 --dtype f16 > /dev/null 2>&1
bash tools/gpu_profile.sh r4_cfg3 --model small --patches 1280 --batch 32 > /dev/null 2>&1
bash tools/gpu_profile.sh r4_cfg5 --model base --patches 1280 --batch 32 --task mpp > /dev/null 2>&1
bash tools/gpu_pmc2.sh r4_block_tail_fwd "mlp_kernel<false, 0, 6, true, true" tools/kbench.py proj_mlp_next_fwd --reps 5 > /dev/null 2>&1
bash tools/gpu_pmc2.sh r4_mlp_bwd "mlp_kernel<true" tools/kbench.py mlp_bwd --reps 5 > /dev/null 2>&1
bash tools/gpu_pmc2.sh r4_attn_fwd_res "attn_fwd_res" tools/attn_bench.py --batch 64 --tokens 321 --heads 3 --only fwd --reps 6 > /dev/null 2>&1
bash tools/gpu_full_timeline.sh > /dev/null 2>&1
ls gpurun_out/ | grep -E "r4_|pmc_r4|full_timeline"
grep -E "FETCH|WRITE_SIZE|WAIT_INST_ANY|WAVE_CYCLES|MFMA_BUSY|GRBM" gpurun_out/pmc_r4_block_tail_fwd.txt
