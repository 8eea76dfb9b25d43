#!/bin/bash
# round 4: the two off-chain changes of the backward's first launches (cast folded into the first MLP backward; the head's
# gradient sums on the side stream): targeted tests, then A/B of the step
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_mlp_fused_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x > gpurun_out/r4_small_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r4_small_tests.log
[ $rc -ne 0 ] && exit $rc
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
for rep in 1 2 3; do
  for cfg in "0 --no-head-deferred" "1 --no-head-deferred" "0" "1"; do
    set -- $cfg
    r=$(SITK_BWD_CAST_FOLD=$1 timeout -k 10 120 python bench.py --steps 40 --warmup 5 --no-also --no-probe --no-cpu-baseline $2 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    echo "cast fold $1 ${2:-head deferred}: $r ms"
  done
done
