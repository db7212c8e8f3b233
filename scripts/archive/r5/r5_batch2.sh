#!/bin/bash
TAG=${1:-r5batch2}; shift
mkdir -p gpurun_out/$TAG
timeout 900 python scripts/r5/r5_batch_probe.py "$@" > gpurun_out/$TAG/probe.txt 2>&1
cat gpurun_out/$TAG/probe.txt
ICPGPU_FLAVOUR=dev ICPGPU_BATCH_TIMING=1 timeout 300 python scripts/r5/r5_batch_probe.py 4x8x8 > gpurun_out/$TAG/timing.txt 2>&1
grep "batch thread 0" gpurun_out/$TAG/timing.txt | tail -3
