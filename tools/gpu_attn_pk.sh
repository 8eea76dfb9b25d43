#!/bin/bash
# round 4: the unit-packed attention kernels (N = 321) -- tests, then timing against the sequence-resident ones
# (SITK_ATTN_PK=0/1 in the diagnostic build) at the benchmark shape and at the widths of small / base
cd "$GRAFT_REPO_ROOT"
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention" > gpurun_out/attn_pk_tests.log 2>&1; rc=$?
tail -5 gpurun_out/attn_pk_tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
export SITK_LIB=$GRAFT_REPO_ROOT/surface-vision-transformers_amd/libsitk_ab.so
for shape in "64 3" "32 6" "16 12"; do
  set -- $shape
  for pk in 0 1; do
    echo "== B=$1 H=$2 SITK_ATTN_PK=$pk"
    SITK_ATTN_PK=$pk timeout -k 10 120 python tools/attn_bench.py --batch $1 --tokens 321 --heads $2 --sets 6 --reps 12 ${ATTN_ONLY:+--only $ATTN_ONLY} || exit 1
  done
done 2>&1 | tee gpurun_out/attn_pk_bench.txt
