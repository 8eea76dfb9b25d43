#!/bin/bash
# round 6: A/B of the final bucket's stream (bucket stream vs main stream) with the side stream probed as a victim of the parked
# bucket stream
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
O=gpurun_out/r6_place2.txt
: > $O
for fo in bucket main bucket main; do
  for sk in 0 1; do
    echo "--- final-on $fo skip $sk" >> $O
    timeout -k 10 200 python tools/dp_cu_budget.py --configs "8:2:42:16" --steps 30 --skip-streams $sk --final-on $fo 2>&1 | grep "ms per step\|placement" | cut -c1-400 >> $O
  done
done
cat $O
