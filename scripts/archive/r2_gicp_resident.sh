#!/bin/bash
# Dev tool (round 2): GICP server with the lane's correspondences resident in registers, by problem size
for m in 0 65536; do
  echo "== ICPGPU_GICP_RESIDENT_MAX=$m"
  ICPGPU_GICP_RESIDENT_MAX=$m python scripts/gicp_timing.py 5000x5000 20000x20000 50000x50000 2>&1 | grep -v amdgpu.ids | cut -c1-170
  ICPGPU_GICP_RESIDENT_MAX=$m python scripts/pipeline_breakdown.py 2>&1 | grep -v amdgpu.ids | head -2
  ICPGPU_GICP_RESIDENT_MAX=$m python scripts/pipeline_breakdown.py 2>&1 | grep -v amdgpu.ids | head -1
done
timeout 600 python -m pytest tests/test_gpu_gicp.py -x -q 2>&1 | tail -2
