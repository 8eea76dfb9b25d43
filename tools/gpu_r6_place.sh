#!/bin/bash
# round 6: which positions of torch's stream pool are toxic for the bucket stream IN THE STEP (not only in the probe)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
O=gpurun_out/r6_place.txt
: > $O
for sk in 0 1 2 3 4 5 6; do
  echo "--- skip $sk" >> $O
  timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 30 --skip-streams $sk 2>&1 | grep "ms per step\|placement" | cut -c1-330 >> $O
done
cat $O
echo "=== timeline of the configuration at skip ${1:-0}" >> $O
SITK_TIMELINE_SIDE=1 timeout -k 10 300 python tools/dp_cu_budget.py --configs "8:2:42:16" --timeline --steps 20 --skip-streams ${1:-0} 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|c10d" | cut -c1-3000 > gpurun_out/r6_place_timeline.txt
tail -45 gpurun_out/r6_place_timeline.txt | cut -c1-400
