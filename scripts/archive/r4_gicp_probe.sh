#!/bin/bash
# round 4: the reference's per-scan pipeline (VoxelGrid 0.2 m + GICP) with the device solver, the evaluation server, single launches
mkdir -p gpurun_out/r4
{
for mode in "ICPGPU_GICP_DEVICE=1" "ICPGPU_GICP_DEVICE=0" "ICPGPU_GICP_DEVICE=0 ICPGPU_GICP_SERVER=0"; do
  echo "== $mode"
  env $mode ICPGPU_DEBUG=0 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v "^\[icpgpu\] grid"
done
} > gpurun_out/r4/gicp_probe.log 2>&1
cat gpurun_out/r4/gicp_probe.log
