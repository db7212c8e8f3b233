#!/bin/bash
# round 6: correspondences per workgroup of the GICP evaluation server (development switch ICPGPU_GICP_PER_BLOCK), reference pipeline
TAG=${1:-r6perblock}
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do for pb in 1024 512 768 1536 2048; do echo "== ICPGPU_GICP_PER_BLOCK=$pb"; ICPGPU_FLAVOUR=dev ICPGPU_GICP_PER_BLOCK=$pb timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep "scans of"; done; done > $O/perblock.txt 2>&1
cat $O/perblock.txt
