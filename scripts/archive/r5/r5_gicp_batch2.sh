#!/bin/bash
TAG=${1:-r5gicpb2}
mkdir -p gpurun_out/$TAG
for q in 4 8; do
echo "== GPU_MAX_HW_QUEUES=$q" >> gpurun_out/$TAG/probe.txt
GPU_MAX_HW_QUEUES=$q CHECK=0 timeout 1200 python scripts/r5_gicp_batch_probe.py 1x8 2x4 1x12 1x4 1x6 >> gpurun_out/$TAG/probe.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/$TAG/probe.txt | tail -20
