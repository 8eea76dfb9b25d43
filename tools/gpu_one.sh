#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm" > gpurun_out/r3_one.log 2>&1; rc=$?
tail -4 gpurun_out/r3_one.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/gemm_bench.py --model small | tail -10
timeout -k 10 200 python tools/gemm_bench.py --model base | tail -10
