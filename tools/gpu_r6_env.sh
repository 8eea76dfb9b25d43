#!/bin/bash
# round 6: do any of the HIP / HSA runtime knobs move the headline step?  (the chain pays ~2.5 us per dependent launch, ~110 launches)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r6_env.txt
: > $O
run() { local tag="$1"; shift; ms=$(env "$@" timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-probe --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(d['ms_per_step'], d['config']['loss_after'])
"); echo "$tag: $ms" | tee -a $O; }
for rep in 1 2; do
run "default" A=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "GPU_MAX_HW_QUEUES=2" GPU_MAX_HW_QUEUES=2
run "GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
run "HSA_ENABLE_INTERRUPT=0" HSA_ENABLE_INTERRUPT=0
run "HSA_ENABLE_SDMA=0" HSA_ENABLE_SDMA=0
run "AMD_DIRECT_DISPATCH=0" AMD_DIRECT_DISPATCH=0
run "HIP_USE_RUNTIME_UNBUNDLER... none; DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run "GPU_STREAMOPS_CP_WAIT=1" GPU_STREAMOPS_CP_WAIT=1
run "HIP_SKIP_ABORT_ON_GPU_ERROR none; ROC_SIGNAL_POOL_SIZE=128" ROC_SIGNAL_POOL_SIZE=128
run "ROC_AQL_QUEUE_SIZE=65536" ROC_AQL_QUEUE_SIZE=65536
done
