#!/bin/bash
# round 6: two de-phased half-batch chains against the whole-batch chain (tools/dual_chain.py) + the placement probe with the
# release inside the chain
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python tools/dual_chain.py > gpurun_out/r6_dual_chain.txt 2>&1 || { tail -20 gpurun_out/r6_dual_chain.txt; exit 1; }
cat gpurun_out/r6_dual_chain.txt
timeout -k 10 300 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -k "picked_by_measurement or one_rank_rccl_group_with_side" 2>&1 | tail -3
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 2>&1 | grep "ms per step\|^#\|placement" | cut -c1-400
