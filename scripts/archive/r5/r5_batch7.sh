#!/bin/bash
TAG=${1:-r5batch7}
mkdir -p gpurun_out/$TAG
for f in 1 0; do
  echo "== ICPGPU_BATCH_LOW_PRIORITY=$f" >> gpurun_out/$TAG/probe.txt
  ICPGPU_FLAVOUR=dev ICPGPU_BATCH_LOW_PRIORITY=$f timeout 300 python scripts/r5/r5_batch_probe.py 4x8x4 4x8x8 2x8x8 1x8x8 4x6x8 4x8x16 >> gpurun_out/$TAG/probe.txt 2>&1
done
cat gpurun_out/$TAG/probe.txt | grep -v amdgpu.ids
ICPGPU_FLAVOUR=dev ICPGPU_BATCH_TRACE=1 REPS=2 timeout 300 python scripts/r5/r5_batch_probe.py 4x8x8 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/$TAG/trace.txt
cat gpurun_out/$TAG/trace.txt
