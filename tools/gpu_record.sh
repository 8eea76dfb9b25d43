#!/bin/bash
# ONE recording run of the GPU tests (bf16 errors -> gpurun_out/parity_measured.json), then the bench lines
cd "$GRAFT_REPO_ROOT"
SITK_PARITY_RECORD=1 timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r3_tests_record.log 2>&1; rc=$?
tail -4 gpurun_out/r3_tests_record.log
[ $rc -eq 124 ] && exit 124
bash tools/gpu_bench4.sh e
