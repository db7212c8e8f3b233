#!/bin/bash
TAG=${1:-r5batch15}
mkdir -p gpurun_out/$TAG
ICPGPU_FLAVOUR=dev ICPGPU_BATCH_TRACE=1 REPS=14 timeout 900 python scripts/r5/r5_batch_probe.py 4x8x8 2>&1 | grep "call of\|pairs/s" > gpurun_out/$TAG/trace.txt
cat gpurun_out/$TAG/trace.txt
