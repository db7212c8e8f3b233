#!/bin/bash
TAG=${1:-r5batch11}
mkdir -p gpurun_out/$TAG
C="4x8x4 4x8x8 4x7x8 4x6x8 4x5x12 4x4x16 1x8x8"
for x in 1 0; do
echo "== ICPGPU_BATCH_XCD_PAIRS=$x, 512 pairs in one call" >> gpurun_out/$TAG/probe.txt
ICPGPU_FLAVOUR=dev ICPGPU_BATCH_XCD_PAIRS=$x PAIRS=512 REPS=4 timeout 900 python scripts/r5/r5_batch_probe.py $C >> gpurun_out/$TAG/probe.txt 2>&1
done
echo "== ICPGPU_BATCH_XCD_PAIRS=1, 64 pairs" >> gpurun_out/$TAG/probe.txt
timeout 900 python scripts/r5/r5_batch_probe.py $C >> gpurun_out/$TAG/probe.txt 2>&1
cat gpurun_out/$TAG/probe.txt | grep -v amdgpu.ids
