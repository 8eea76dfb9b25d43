#!/bin/bash
# round-2 run 1: GPU tests in recording mode, bench lines for every BASELINE configuration, kernel tables
cd "$GRAFT_REPO_ROOT"
step() {  # run a step; stop the whole script if it was killed by its timeout
  "$@"; rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "STEP KILLED ($rc): $*"; exit $rc; fi
  return 0
}
SITK_PARITY_RECORD=1 step timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r2_tests1.log 2>&1
tail -5 gpurun_out/r2_tests1.log
step timeout -k 10 300 python bench.py --steps 30 --warmup 5 > gpurun_out/r2_bench_tiny.json 2> gpurun_out/r2_bench_tiny.err
step timeout -k 10 200 python bench.py --steps 10 --warmup 3 --dtype f32 --no-cpu-baseline --no-probe > gpurun_out/r2_bench_tiny_f32.json 2>&1
step timeout -k 10 200 python bench.py --steps 20 --warmup 3 --task mpp --no-cpu-baseline --no-probe > gpurun_out/r2_bench_tiny_mpp.json 2>&1
step timeout -k 10 300 python bench.py --steps 10 --warmup 3 --model small --patches 1280 --batch 32 --no-cpu-baseline > gpurun_out/r2_bench_cfg3.json 2>&1
step timeout -k 10 300 python bench.py --steps 10 --warmup 3 --model base --patches 1280 --batch 32 --task mpp --no-cpu-baseline > gpurun_out/r2_bench_cfg5.json 2>&1
step bash tools/gpu_profile.sh r2_tiny_v0
step bash tools/gpu_profile.sh r2_mpp_tiny_v0 --task mpp
step bash tools/gpu_profile.sh r2_cfg5_v0 --model base --patches 1280 --batch 32 --task mpp
echo ALL DONE
