#!/bin/bash
# round 6 mid-round batch: whole GPU suite on the current build, the stand-in study with the fixed placement probe (shipped
# arrangement x3, control x1, 8 hardware queues x2), the rocprofv3 trace of where a collective lands, kernel tables of the three
# benchmarked configurations
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r6_tests_mid.log 2>&1; rc=$?
tail -5 gpurun_out/r6_tests_mid.log
[ $rc -eq 124 ] && exit 124
[ $rc -eq 137 ] && exit 137
cp gpurun_out/parity_measured.json gpurun_out/parity_measured_mid.json 2>/dev/null
O=gpurun_out/r6_dp3
: > $O.txt
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d'
run() { echo "--- $*" >> $O.txt; timeout -k 10 200 "$@" 2>&1 | grep -v "$F" | grep "ms per step\|^#\|placement" | cut -c1-400 >> $O.txt; }
for rep in 1 2 3; do
  run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40
done
run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --collective group
run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --gbs 44
export GPU_MAX_HW_QUEUES=8
run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40
run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --skip-streams 2
unset GPU_MAX_HW_QUEUES
cat $O.txt
rm -rf gpurun_out/prof_cs
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_cs -- python tools/dp_collective_stream.py > gpurun_out/r6_collective_stream.log 2>&1 || { tail -5 gpurun_out/r6_collective_stream.log; exit 1; }
grep "^torch\|^[ABC]:" gpurun_out/r6_collective_stream.log > gpurun_out/r6_collective_stream.txt
python tools/dp_collective_stream.py --read gpurun_out/prof_cs | grep -v "at::native\|fillBuffer" >> gpurun_out/r6_collective_stream.txt
rm -rf gpurun_out/prof_cs
unset SITK_LIB
bash tools/gpu_profile.sh r6_tiny > gpurun_out/r6_prof_tiny.log 2>&1 || { tail -5 gpurun_out/r6_prof_tiny.log; exit 1; }
bash tools/gpu_profile.sh r6_cfg3 --model small --patches 1280 --batch 32 > gpurun_out/r6_prof_cfg3.log 2>&1 || { tail -5 gpurun_out/r6_prof_cfg3.log; exit 1; }
bash tools/gpu_profile.sh r6_cfg5 --model base --patches 1280 --batch 32 --task mpp > gpurun_out/r6_prof_cfg5.log 2>&1 || { tail -5 gpurun_out/r6_prof_cfg5.log; exit 1; }
head -16 gpurun_out/prof_r6_tiny.md
