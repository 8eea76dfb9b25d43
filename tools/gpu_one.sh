#!/bin/bash
cd "$GRAFT_REPO_ROOT"
SITK_PARITY_RECORD=1 timeout -k 10 600 python -m pytest tests/test_models_gpu.py -m gpu -q -k "narrow or wide_heads or dropout" > gpurun_out/r3_one.log 2>&1; rc=$?
tail -25 gpurun_out/r3_one.log
grep "parity bf16 dropout" gpurun_out/r3_one.log
exit $rc
