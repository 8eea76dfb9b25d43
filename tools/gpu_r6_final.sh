#!/bin/bash
# round 6 closing batch: whole GPU suite, smoke, the driver's bench command (with `also`), counters of the dominant kernel,
# the stand-in study at the engine's defaults (placement probed against the main AND the side stream), three repetitions
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r6_tests_final.log 2>&1; rc=$?
tail -4 gpurun_out/r6_tests_final.log
[ $rc -eq 124 ] && exit 124
[ $rc -eq 137 ] && exit 137
cp gpurun_out/parity_measured.json gpurun_out/parity_measured_final.json 2>/dev/null
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_driver.json 2> gpurun_out/r6_bench_driver.err || { tail -5 gpurun_out/r6_bench_driver.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6_bench_driver.json") if l.startswith("{")][0])
print("headline", d["ms_per_step"], d["value"], d["step_mfma_frac"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("traffic"))
print("  probes", d["config"].get("stream_probe"))
for k, v in d.get("also", {}).items():
    print("  also", k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "value", "step_mfma_frac", "error")})
print("  cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
PY
bash tools/gpu_pmc2.sh r6_ln_gemm_mlp_bwd "ln_gemm_mlp_bwd_kernel" tools/kbench.py lnqkv_mlp_bwd > gpurun_out/r6_pmc.log 2>&1
grep "FETCH_SIZE\|WRITE_SIZE\|SQ_INSTS_MFMA \|SQ_WAIT_INST_ANY\|SQ_WAVE_CYCLES" gpurun_out/pmc_r6_ln_gemm_mlp_bwd.txt
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
O=gpurun_out/r6_dp_final.txt
: > $O
for rep in 1 2 3; do
  timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 2>&1 | grep "ms per step\|placement\|^#" | cut -c1-420 >> $O || exit 1
done
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 2>&1 | grep "ms per step\|placement\|^#" | cut -c1-420 >> $O
cat $O
