#!/bin/bash
TAG=${1:-r5batch6}
mkdir -p gpurun_out/$TAG
for c in 4x8x8 4x6x8 4x8x4; do
  echo "== $c" >> gpurun_out/$TAG/trace.txt
  ICPGPU_FLAVOUR=dev ICPGPU_BATCH_TRACE=1 REPS=2 timeout 300 python scripts/r5/r5_batch_probe.py $c 2>&1 | grep -v amdgpu.ids | tail -14 >> gpurun_out/$TAG/trace.txt
done
cat gpurun_out/$TAG/trace.txt
