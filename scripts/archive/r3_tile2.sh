#!/bin/bash
O=gpurun_out/r3t; mkdir -p $O
for S in 1 4 8; do echo "## splits $S"; ICPGPU_TILE_SPLITS=$S ICPGPU_TILE_SEARCH=2 timeout 600 python scripts/one_align.py 200000x200000 2>&1 | grep "tile search" | head -4; done
