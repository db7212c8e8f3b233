#!/bin/bash
TAG=${1:-r5batch3}
mkdir -p gpurun_out/$TAG
for q in 4 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q" >> gpurun_out/$TAG/probe.txt
  GPU_MAX_HW_QUEUES=$q timeout 300 python scripts/r5/r5_batch_probe.py 4x8x4 4x8x8 1x8x8 8x8x8 >> gpurun_out/$TAG/probe.txt 2>&1
done
cat gpurun_out/$TAG/probe.txt | grep -v amdgpu.ids
for g in 4 8; do
ICPGPU_FLAVOUR=dev ICPGPU_BATCH_TIMING=1 timeout 300 python scripts/r5/r5_batch_probe.py 4x8x$g > gpurun_out/$TAG/timing$g.txt 2>&1
grep "batch thread" gpurun_out/$TAG/timing$g.txt | tail -4
done
