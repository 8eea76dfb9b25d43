#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for ov in 0 5 7 8; do
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --wgrad-overlap $ov --no-graph --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('eager overlap $ov', d['ms_per_step'], d['value'])" || exit 1
  done
done
rm -rf gpurun_out/prof_ovg
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ovg -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-probe --wgrad-overlap 7 > gpurun_out/prof_ovg.log 2>&1 || { tail -5 gpurun_out/prof_ovg.log; exit 1; }
f=$(ls gpurun_out/prof_ovg/*/*_kernel_trace.csv | head -1)
python tools/trace_step_timeline.py "$f" "wgrad|ln_finalize" > gpurun_out/ov_timeline_graph7.txt
rm -rf gpurun_out/prof_ovg
