#!/bin/bash
TAG=${1:-r5batch5}
mkdir -p gpurun_out/$TAG
for f in 1 0; do
  echo "== ICPGPU_BATCH_STAGGER=$f" >> gpurun_out/$TAG/probe.txt
  ICPGPU_FLAVOUR=dev ICPGPU_BATCH_STAGGER=$f timeout 300 python scripts/r5/r5_batch_probe.py 4x8x4 4x8x8 2x8x8 1x8x8 4x6x8 4x8x12 4x8x16 4x6x12 4x12x8 >> gpurun_out/$TAG/probe.txt 2>&1
done
cat gpurun_out/$TAG/probe.txt | grep -v amdgpu.ids
