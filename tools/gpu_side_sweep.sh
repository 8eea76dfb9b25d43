#!/bin/bash
# side-stream sweep of the default step: layers whose weight gradients run on the side stream x workgroups of one side launch
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/side_sweep.txt
: > $out
for L in ${LAYERS:-6 7 8 9}; do
  for C in ${CUS:-32 42 56 72}; do
    r=$(timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-also --no-probe --no-cpu-baseline --wgrad-overlap $L --overlap-cus $C 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    echo "layers $L workgroups $C : $r ms" | tee -a $out
  done
done
