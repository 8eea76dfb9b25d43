#!/bin/bash
# round 5: the whole all-reduce path (engine's bucket stream, process group, stand-in) at HIGH priority = hardware queues of its
# own (main: normal, side: lowest), repeated to see the run-to-run spread
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
{
for rep in 1 2; do
echo "=== high priority, repetition $rep"
timeout -k 10 500 python tools/dp_cu_budget.py --configs "${1:-8:2:42:16,7:2:42:16,6:2:42:16,8:2:42:8,8:2:42:32}" --steps 30 --prio -1 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d" || exit 1
done
echo "=== default priority"
timeout -k 10 500 python tools/dp_cu_budget.py --configs "${1:-8:2:42:16,7:2:42:16,6:2:42:16}" --steps 30 --prio 0 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d" || exit 1
} > gpurun_out/r5_dp_budget_g.txt
cat gpurun_out/r5_dp_budget_g.txt
