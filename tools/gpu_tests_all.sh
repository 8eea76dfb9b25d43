#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r3_tests_all.log 2>&1; rc=$?
tail -12 gpurun_out/r3_tests_all.log
exit $rc
