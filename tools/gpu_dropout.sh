#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_models_gpu.py -m gpu -q -x -k "dropout" > gpurun_out/r3_dropout_tests.log 2>&1; rc=$?
tail -30 gpurun_out/r3_dropout_tests.log
exit $rc
