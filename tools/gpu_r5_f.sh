#!/bin/bash
# round 5: stand-in sweep over the number of side layers (slack on the side stream), default- and high-priority all-reduce path
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
{
echo "=== stand-in stream priority 0"
timeout -k 10 500 python tools/dp_cu_budget.py --configs "8:2:42:16,7:2:42:16,6:2:42:16,5:2:42:16,4:2:42:16,6:2:56:16,6:2:42:32" --steps 30 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d" || exit 1
echo "=== stand-in stream priority -1 (high)"
timeout -k 10 500 python tools/dp_cu_budget.py --configs "8:2:42:16,6:2:42:16,4:2:42:16" --steps 30 --prio -1 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d" || exit 1
} > gpurun_out/r5_dp_budget_f.txt
cat gpurun_out/r5_dp_budget_f.txt
