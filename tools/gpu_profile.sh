#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace profile of the default bench, summary into gpurun_out/.
# usage: bash tools/gpu_profile.sh <tag> [bench args...]
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probe --no-graph --no-also "$@" > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
f=$(ls gpurun_out/prof_$tag/*/*_kernel_stats.csv | head -1)
python tools/prof_summary.py "$f" --steps 23 --title "bench.py $* (rocprofv3 --kernel-trace --stats, eager launches)" > gpurun_out/prof_$tag.md
rm -f gpurun_out/prof_$tag/*/*_kernel_trace.csv
head -30 gpurun_out/prof_$tag.md
