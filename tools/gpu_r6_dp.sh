#!/bin/bash
# round 6, review item 1: the data-parallel step's collectives on the stream they are measured on.
#  1. the data-parallel GPU tests with the shipped arrangement (synchronous collectives on the bucket stream)
#  2. tools/micro/blocked_queue: what a stream blocked behind an event costs a dependent chain, per stream and queue count
#  3. rocprofv3 trace: on which stream a one-rank collective lands when issued the engine's way / the old way
#  4. the stand-in study in the shipped arrangement (default) against the control (async_op=True + a stream of its own), and the
#     control with the stand-in's stream at other positions of torch's stream pool / with 8 hardware queues
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_dp
: > $O.txt
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -m gpu -q -x > gpurun_out/r6_dp_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r6_dp_tests.log
[ $rc -ne 0 ] && exit $rc
echo "=== blocked_queue, default queues" >> $O.txt
timeout -k 10 120 ./build/blocked_queue 8 1 >> $O.txt 2>&1 || exit 1
echo "=== blocked_queue, GPU_MAX_HW_QUEUES=8" >> $O.txt
GPU_MAX_HW_QUEUES=8 timeout -k 10 120 ./build/blocked_queue 8 1 >> $O.txt 2>&1 || exit 1
echo "=== blocked_queue, GPU_MAX_HW_QUEUES=2" >> $O.txt
GPU_MAX_HW_QUEUES=2 timeout -k 10 120 ./build/blocked_queue 8 1 >> $O.txt 2>&1 || exit 1
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
rm -rf gpurun_out/prof_cs
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_cs -- python tools/dp_collective_stream.py > gpurun_out/r6_collective_stream.log 2>&1 || { tail -5 gpurun_out/r6_collective_stream.log; exit 1; }
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d" gpurun_out/r6_collective_stream.log > gpurun_out/r6_collective_stream.txt
python tools/dp_collective_stream.py --read gpurun_out/prof_cs >> gpurun_out/r6_collective_stream.txt
rm -rf gpurun_out/prof_cs
cat gpurun_out/r6_collective_stream.txt
F='^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d'
run() { echo "--- $*" >> $O.txt; timeout -k 10 200 "$@" 2>&1 | grep -v "$F" | grep "ms per step\|^#" | cut -c1-230 >> $O.txt; }
for rep in 1 2; do
  run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40
  run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --collective group
done
for sk in 1 2 3 4; do
  run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --collective group --skip-streams $sk
done
export GPU_MAX_HW_QUEUES=8
run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40
run python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 40 --collective group
unset GPU_MAX_HW_QUEUES
cat $O.txt
