#!/bin/bash
# every kernel of one default step as a timeline -> gpurun_out/full_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_full
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_full -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-probe --no-also "$@" > gpurun_out/prof_full.log 2>&1 || { tail -5 gpurun_out/prof_full.log; exit 1; }
f=$(ls gpurun_out/prof_full/*/*_kernel_trace.csv | head -1)
python tools/trace_step_timeline.py "$f" "." > gpurun_out/full_timeline.txt
rm -rf gpurun_out/prof_full
head -30 gpurun_out/full_timeline.txt; echo ...; tail -22 gpurun_out/full_timeline.txt
