#!/bin/bash
TAG=${1:-r5batch10}
mkdir -p gpurun_out/$TAG
C="4x8x4 4x8x8 4x7x8 4x6x8 4x5x12 4x4x16 4x8x6 2x7x8 1x7x8 1x8x8"
timeout 900 python scripts/r5/r5_batch_probe.py $C >> gpurun_out/$TAG/probe.txt 2>&1
echo "== 512 pairs in one call" >> gpurun_out/$TAG/probe.txt
PAIRS=512 REPS=4 timeout 900 python scripts/r5/r5_batch_probe.py $C >> gpurun_out/$TAG/probe.txt 2>&1
cat gpurun_out/$TAG/probe.txt | grep -v amdgpu.ids
