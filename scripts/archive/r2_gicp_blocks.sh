#!/bin/bash
# Dev tool (round 2): how many workgroups should a GICP cost evaluation use?  (64 until round 2)
for b in 64 128 256; do
  echo "== ICPGPU_GICP_BLOCKS=$b"
  ICPGPU_GICP_BLOCKS=$b python scripts/gicp_timing.py 2>&1 | grep -v amdgpu.ids
done
