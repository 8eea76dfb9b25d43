#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_dps
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_dps -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-probe --dp-form > gpurun_out/prof_dps.log 2>&1 || { tail -5 gpurun_out/prof_dps.log; exit 1; }
f=$(ls gpurun_out/prof_dps/*/*_kernel_trace.csv | head -1)
python tools/trace_step_timeline.py "$f" "wgrad|ln_finalize|nccl|Nccl|rccl|mlp_kernel<true|sgd_dev|colsum|elementwise|cast_rows" > gpurun_out/dps_timeline.txt
rm -rf gpurun_out/prof_dps
