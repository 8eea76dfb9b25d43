#!/bin/bash
# round 6: engine.SplitTrainEngine -- tests, the config-3 oracle test in its new (split) form, bench A/B whole vs split
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -s -k "split_engine or config3_benchmarked or optimizer_skips or graph_follows" > gpurun_out/r6_split_tests.log 2>&1; rc=$?
grep "parity\|worst\|passed\|failed\|Error\|error" gpurun_out/r6_split_tests.log | tail -20
[ $rc -ne 0 ] && { tail -30 gpurun_out/r6_split_tests.log; exit $rc; }
O=gpurun_out/r6_split_bench.txt
: > $O
for rep in 1 2; do
  for f in "" "--whole-batch"; do
    for cfg in "--model small --patches 1280 --batch 32" "--model small --patches 320 --batch 64"; do
      timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-probe --no-cpu-baseline --no-also $cfg $f 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$cfg $f'.ljust(60), d['ms_per_step'], d['value'], d['step_mfma_frac'], 'parts', d['config']['batch_parts'], 'loss', d['config']['loss_after'])
" >> $O
    done
  done
done
cat $O
