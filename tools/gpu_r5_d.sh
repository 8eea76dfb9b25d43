#!/bin/bash
# round 5: rocprofv3 kernel-trace timelines of the data-parallel step on a one-rank RCCL group, without and with the all-reduce
# stand-in (every kernel of the last complete step, per hardware queue)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
for ch in 0 16; do
  rm -rf gpurun_out/prof_dpt$ch
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_dpt$ch -- python tools/dp_cu_budget.py --configs "8:2:42:$ch" --steps 6 > gpurun_out/prof_dpt$ch.log 2>&1 || { tail -5 gpurun_out/prof_dpt$ch.log; exit 1; }
  f=$(ls gpurun_out/prof_dpt$ch/*/*_kernel_trace.csv | head -1)
  python tools/trace_step_timeline.py "$f" "." > gpurun_out/r5_dp_trace_ch$ch.txt
  grep "ms per step" gpurun_out/prof_dpt$ch.log
  rm -rf gpurun_out/prof_dpt$ch
done
grep -E "step:|wgrad|occupy|nccl|Nccl|rccl|sgd_dev|ln_finalize|colsum|gather|stage_w" gpurun_out/r5_dp_trace_ch16.txt
