#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for form in "" "--dp-form"; do
  tag=r6_cfg5${form:+_dpform}
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-probe --no-also --model base --patches 1280 --batch 32 --task mpp $form > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
  f=$(ls gpurun_out/prof_$tag/*/*_kernel_stats.csv | head -1)
  python tools/prof_summary.py "$f" --steps 11 --title "cfg5 $form" > gpurun_out/prof_$tag.md
  rm -f gpurun_out/prof_$tag/*/*_kernel_trace.csv
  head -18 gpurun_out/prof_$tag.md | cut -c1-150
done
