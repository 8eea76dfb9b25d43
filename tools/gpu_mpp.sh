#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_models_gpu.py tests/test_dp_gpu.py -m gpu -q -x -k "mpp" > gpurun_out/r3_mpp_tests.log 2>&1; rc=$?
tail -8 gpurun_out/r3_mpp_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --model base --patches 1280 --batch 32 --task mpp --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('cfg5', d['ms_per_step'], d['value'])"
timeout -k 10 200 python bench.py --task mpp --steps 20 --warmup 3 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json;d=json.loads([l for l in sys.stdin if l.startswith('{')][0]);print('tiny mpp', d['ms_per_step'], d['value'])"
