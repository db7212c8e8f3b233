#!/bin/bash
TAG=${1:-r5batch4}
mkdir -p gpurun_out/$TAG
for f in 2 4 8; do
  echo "== ICPGPU_BATCH_FIRST=$f" >> gpurun_out/$TAG/probe.txt
  ICPGPU_FLAVOUR=dev ICPGPU_BATCH_FIRST=$f timeout 300 python scripts/r5/r5_batch_probe.py 4x8x4 4x8x8 2x8x8 1x8x8 4x6x8 4x4x8 >> gpurun_out/$TAG/probe.txt 2>&1
done
cat gpurun_out/$TAG/probe.txt | grep -v amdgpu.ids
