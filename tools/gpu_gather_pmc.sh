#!/bin/bash
# round 4: the patch gather on distinct batches -- tests, time, counters (FETCH / WRITE_SIZE, L2 hit rate)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gather" 2>&1 | tail -2 || exit 1
timeout -k 10 120 python tools/gather_bench.py 2>&1 | grep gather | tee gpurun_out/r4_gather_bench.txt
bash tools/gpu_pmc2.sh gather "gather_tokens" tools/gather_bench.py --reps 16 > /dev/null 2>&1
grep -E "kernels|GRBM|WAIT|WAVE_CYCLES|SQ_WAVES|INSTS_VALU|INSTS_VMEM|FETCH|WRITE|TCC" gpurun_out/pmc_gather.txt
