#!/bin/bash
# round 5 experiment: QT query tiles per wave in the sequence-resident attention kernels (diagnostic build)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export SITK_LIB=$PWD/surface-vision-transformers_amd/libsitk_ab.so
for v in ${1:-82 83 122}; do
  SITK_ATTN_FWD_QT=$v SITK_ATTN_BWD_QT=$v timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention" > gpurun_out/attn_qt_tests_$v.log 2>&1 || { tail -15 gpurun_out/attn_qt_tests_$v.log; exit 1; }
  tail -1 gpurun_out/attn_qt_tests_$v.log
done
for rep in 1 2; do
  for v in 0 ${1:-82 83 122}; do
    echo -n "QT=$v: "; SITK_ATTN_FWD_QT=$v SITK_ATTN_BWD_QT=$v timeout -k 10 200 python tools/attn_bench.py --batch 64 --tokens 321 --heads 3 --sets 12 --reps 24 ${ONLY:+--only $ONLY} 2>/dev/null | tr '\n' ' '; echo
  done
done
