#!/bin/bash
# f16 bring-up: the f16 kernel / model / engine tests, then bf16 vs f16 bench lines
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests -m gpu -q -k "f16 or fused or colsum or head" > gpurun_out/r3_f16_tests.log 2>&1; rc=$?
tail -25 gpurun_out/r3_f16_tests.log
[ $rc -eq 124 ] && exit 124
grep -h "^parity f16" gpurun_out/r3_f16_tests.log | sort -u > gpurun_out/r3_f16_parity.txt
for dt in bf16 f16; do
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --dtype $dt --no-cpu-baseline --no-probe > gpurun_out/r3_bench_tiny_$dt.json 2> gpurun_out/r3_bench_tiny_$dt.err || { tail -5 gpurun_out/r3_bench_tiny_$dt.err; }
  python -c "import json;d=json.loads([l for l in open('gpurun_out/r3_bench_tiny_$dt.json') if l.startswith('{')][0]);print('$dt',d['ms_per_step'],d['value'],d['config']['loss_after'])"
done
