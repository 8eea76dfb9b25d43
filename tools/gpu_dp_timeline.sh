#!/bin/bash
# round 5: un-profiled timelines of the data-parallel step (HIP events on the streams + the library's per-launch marks on the
# main stream with the side stream running), without and with the all-reduce stand-in
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
SITK_TIMELINE_SIDE=1 timeout -k 10 300 python tools/dp_cu_budget.py --configs "${1:-8:2:42:16}" --timeline --steps 20 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d" > gpurun_out/r5_dp_timeline2.txt || exit 1
cat gpurun_out/r5_dp_timeline2.txt
