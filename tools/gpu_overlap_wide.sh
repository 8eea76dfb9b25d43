#!/bin/bash
# round 4: weight gradients of the wide models (configs 3 / 5) on the side stream beside the backward chain?
cd "$GRAFT_REPO_ROOT"
run() { timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-probe --no-also "$@" 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]);print(d['ms_per_step'], d['step_mfma_frac'])"; }
for cfg in "--model small --patches 1280 --batch 32" "--model base --patches 1280 --batch 32 --task mpp"; do
  echo "== $cfg"
  echo -n "graph (default): "; run $cfg
  echo -n "eager, no side : "; run $cfg --no-graph --wgrad-overlap 0
  for ov in 6 10; do for cus in 64 128 256; do
    echo -n "side $ov layers, $cus workgroups: "; run $cfg --no-graph --wgrad-overlap $ov --overlap-cus $cus || echo failed
  done; done
done 2>&1 | tee gpurun_out/r4_overlap_wide.txt
