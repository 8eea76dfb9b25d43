#!/bin/bash
# usage (on the GPU box): bash tools/gpu_ab.sh "<test files>" VAR "v1 v2 ..." [bench args]   -- step-time A/B over an env switch
cd "$GRAFT_REPO_ROOT"
tests=$1; var=$2; vals=$3; shift 3
if [ -n "$tests" ]; then
  timeout -k 10 600 python -m pytest $tests -m gpu -x -q > gpurun_out/ab_tests.log 2>&1; rc=$?
  tail -3 gpurun_out/ab_tests.log
  if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
fi
for rep in ${AB_REPS:-1 2}; do
  for v in $vals; do
    echo -n "$var=$v: "; env $var=$v timeout -k 10 200 python bench.py --steps ${AB_STEPS:-30} --warmup 5 --no-cpu-baseline --no-probe "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" || exit 1
  done
done
