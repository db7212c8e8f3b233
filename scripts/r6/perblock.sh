#!/bin/bash
# round 6: correspondences per workgroup of the GICP evaluation server (development switch ICPGPU_GICP_PER_BLOCK), reference pipeline
TAG=${1:-r6perblock3}
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2 3; do for pb in 512 256 384 640; do echo "== ICPGPU_GICP_PER_BLOCK=$pb"; ICPGPU_FLAVOUR=dev ICPGPU_GICP_PER_BLOCK=$pb timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep "scans of\|host wall"; done; done > $O/perblock.txt 2>&1
cat $O/perblock.txt
