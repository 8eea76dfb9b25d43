#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ov in 9 0; do
rm -rf gpurun_out/prof_ov$ov
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ov$ov -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-probe --no-graph --wgrad-overlap $ov > gpurun_out/prof_ov$ov.log 2>&1 || { tail -5 gpurun_out/prof_ov$ov.log; exit 1; }
f=$(ls gpurun_out/prof_ov$ov/*/*_kernel_trace.csv | head -1)
python tools/trace_step_timeline.py "$f" "wgrad|mlp_kernel<true|ln_finalize|attn_bwd_dq" > gpurun_out/ov_timeline_$ov.txt
rm -rf gpurun_out/prof_ov$ov
done
