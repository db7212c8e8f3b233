#!/bin/bash
# Dev tool (round 4): host wall per stage of align_gicp on the reference's pipeline (development flavour, ICPGPU_GICP_TIMING=1)
mkdir -p gpurun_out/r4s
export ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1
timeout 300 python scripts/pipeline_breakdown.py 43 > gpurun_out/r4s/stages.txt 2>&1
echo "rc=$?" >> gpurun_out/r4s/stages.txt
