#!/bin/bash
# round 5, third call: (a) event timeline of the data-parallel step with the all-reduce stand-in -- default HW queues, 8 HW queues,
# stand-in on a high-priority stream; (b) counters of the merged attention backward, classic plan against the Q-resident one
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
{
echo "=== default queues, stand-in 16 channels, default priority"
timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --timeline --steps 20 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" || exit 1
echo "=== GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --timeline --steps 20 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" || exit 1
echo "=== stand-in on a HIGH-priority stream"
timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --timeline --steps 20 --prio -1 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" || exit 1
} > gpurun_out/r5_dp_timeline.txt
cat gpurun_out/r5_dp_timeline.txt
for v in 0 1; do
  SITK_ATTN_QRES=$v bash tools/gpu_pmc2.sh attn_bwd_qres$v "attn_bwd_res_kernel" bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-probe --no-also > gpurun_out/r5_pmc_qres$v.log 2>&1 || { tail -5 gpurun_out/r5_pmc_qres$v.log; exit 1; }
  grep -E "FETCH_SIZE|WRITE_SIZE|GRBM_GUI|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_INSTS_VMEM|SQ_INSTS_LDS|TCC_" gpurun_out/pmc_attn_bwd_qres$v.txt
done
