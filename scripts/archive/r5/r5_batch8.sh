#!/bin/bash
TAG=${1:-r5batch8}
mkdir -p gpurun_out/$TAG
timeout 600 python scripts/r5/r5_batch_probe.py 4x5x8 4x6x8 4x7x8 2x6x8 1x6x8 4x6x6 4x6x10 2x6x6 1x6x6 4x8x4 >> gpurun_out/$TAG/probe.txt 2>&1
echo "== 512 pairs in one call" >> gpurun_out/$TAG/probe.txt
PAIRS=512 REPS=4 timeout 600 python scripts/r5/r5_batch_probe.py 4x8x4 4x8x8 4x6x8 2x8x8 1x8x8 >> gpurun_out/$TAG/probe.txt 2>&1
cat gpurun_out/$TAG/probe.txt | grep -v amdgpu.ids
