#!/bin/bash
# end-of-round check: full GPU test suite, smoke, the bench lines, one kernel-trace timeline of the default step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_tests_final.log 2>&1; rc=$?
tail -3 gpurun_out/r3_tests_final.log
[ $rc -eq 124 ] && exit 124
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke.log 2>&1; tail -3 gpurun_out/r3_smoke.log
bash tools/gpu_bench4.sh final
rm -rf gpurun_out/prof_ovf
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ovf -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-probe > gpurun_out/prof_ovf.log 2>&1 || { tail -5 gpurun_out/prof_ovf.log; exit 1; }
f=$(ls gpurun_out/prof_ovf/*/*_kernel_trace.csv | head -1)
python tools/trace_step_timeline.py "$f" "wgrad|ln_finalize|mlp_kernel<true|stage_weights|gather_tokens|colsum|sgd_dev|head_" > gpurun_out/r03_overlap_timeline.txt
rm -rf gpurun_out/prof_ovf
