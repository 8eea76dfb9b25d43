#!/bin/bash
# round 5 end-of-round batch: whole GPU suite, the driver's bench command (with `also`), the kernel table of the default step,
# the data-parallel stand-in at the engine's defaults + its kernel-trace timeline, counters of the ring attention kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r5_tests_final.log 2>&1; rc=$?
tail -5 gpurun_out/r5_tests_final.log
[ $rc -eq 124 ] && exit 124
[ $rc -eq 137 ] && exit 137
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_bench_driver.json 2> gpurun_out/r5_bench_driver.err || { tail -5 gpurun_out/r5_bench_driver.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_bench_driver.json") if l.startswith("{")][0])
print("headline", d["ms_per_step"], d["value"], d["step_mfma_frac"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic"))
for k, v in d.get("also", {}).items():
    print("  also", k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "value", "step_mfma_frac", "error")})
print("  cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
bash tools/gpu_profile.sh r5_tiny --no-also > gpurun_out/r5_prof_tiny.log 2>&1 || { tail -5 gpurun_out/r5_prof_tiny.log; exit 1; }
head -22 gpurun_out/prof_r5_tiny.md
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
: > gpurun_out/r5_dp_final.txt
for rep in 1 2 3; do
  timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --own-stream 0 2>&1 | grep "ms per step" | cut -c1-190 >> gpurun_out/r5_dp_final.txt || exit 1
done
cat gpurun_out/r5_dp_final.txt
rm -rf gpurun_out/prof_dpf
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_dpf -- python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 6 --own-stream 0 > gpurun_out/prof_dpf.log 2>&1 || { tail -5 gpurun_out/prof_dpf.log; exit 1; }
f=$(ls gpurun_out/prof_dpf/*/*_kernel_trace.csv | head -1)
python tools/trace_step_timeline.py "$f" "wgrad|occupy|nccl|Nccl|rccl|sgd_dev|ln_finalize|colsum|gather|stage_w|ln_gemm_bwd_kernel|head_" > gpurun_out/r5_dp_trace_final.txt
rm -rf gpurun_out/prof_dpf
cat gpurun_out/r5_dp_trace_final.txt
unset SITK_LIB
bash tools/gpu_attn_pmc.sh > gpurun_out/r5_attn_ring_pmc.txt 2>&1
tail -60 gpurun_out/r5_attn_ring_pmc.txt
