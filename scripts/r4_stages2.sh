#!/bin/bash
mkdir -p gpurun_out/r4s
export ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1
for a in 1 2 3; do
  timeout 300 python scripts/pipeline_breakdown.py 43 >> gpurun_out/r4s/stages2.txt 2>&1
done
