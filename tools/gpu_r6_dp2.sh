#!/bin/bash
# round 6, second batch: (1) tools/micro/blocked_queue with the hardware queues of the X streams created 0 - 3 queues LATER (does the
# toxic stream index move with the creation order?), (2) engine + data-parallel tests with the measured stream placement,
# (3) the stand-in study with the placement probes printed, both arrangements, 4 and 8 hardware queues, (4) the headline bench,
# (5) the new oracle tests of the benchmarked config-3 / config-5 engine forms in recording mode
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_dp2
: > $O.txt
for d in 0 1 2 3; do
  echo "=== blocked_queue, default queues, $d dummy low-priority queues first" >> $O.txt
  timeout -k 10 120 ./build/blocked_queue 8 1 $d 2 >> $O.txt 2>&1 || exit 1
done
echo "=== blocked_queue, chain on a created stream" >> $O.txt
timeout -k 10 120 ./build/blocked_queue 8 0 0 2 >> $O.txt 2>&1 || exit 1
cat $O.txt
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "not benchmarked_form" > gpurun_out/r6_dp2_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r6_dp2_tests.log
[ $rc -ne 0 ] && exit $rc
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d'
run() { echo "--- $*" >> $O.txt; timeout -k 10 200 "$@" 2>&1 | grep -v "$F" | grep "ms per step\|^#\|placement" | cut -c1-400 >> $O.txt; }
for rep in 1 2; do
  run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40
done
run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --collective group
for sk in 1 2 3; do
  run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --skip-streams $sk
done
export GPU_MAX_HW_QUEUES=8
run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40
run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --skip-streams 2
unset GPU_MAX_HW_QUEUES
unset SITK_LIB
tail -40 $O.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_a.json 2> gpurun_out/r6_bench_a.err || { tail -5 gpurun_out/r6_bench_a.err; exit 1; }
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r6_bench_a.json") if l.startswith("{")][0])
print("headline", d["ms_per_step"], d["value"], d["step_mfma_frac"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
print("  probes", d["config"].get("stream_probe"))
for k, v in d.get("also", {}).items():
    print("  also", k, {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "value", "step_mfma_frac", "error")})
PY
SITK_PARITY_RECORD=1 timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "benchmarked_form" > gpurun_out/r6_oracle_forms.log 2>&1; rc=$?
grep "parity\|worst\|passed\|failed\|Error" gpurun_out/r6_oracle_forms.log | tail -30
cp gpurun_out/parity_measured.json gpurun_out/parity_measured_forms.json 2>/dev/null
exit $rc
