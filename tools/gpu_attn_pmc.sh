#!/bin/bash
# counters of the LDS-ring attention kernels at config 3's shape (B 32, N 1281, 6 heads)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python tools/attn_bench.py > gpurun_out/attn_ring.txt 2>&1; cat gpurun_out/attn_ring.txt
for k in fwd bwd_dq bwd_dkv; do
  bash tools/gpu_pmc2.sh attn_ring_$k "attn_.*ring" tools/attn_bench.py --only $k --reps 4 > /dev/null 2>&1
  echo "== $k"; grep -E "kernels|GRBM|MFMA|LDS_BANK|LDS_IDX|WAIT|WAVE_CYCLES|SQ_WAVES|INSTS_VALU|INSTS_LDS|FETCH|WRITE|TCC_REQ|TCC_HIT" gpurun_out/pmc_attn_ring_$k.txt
done
