#!/bin/bash
O=gpurun_out/r3t; mkdir -p $O
for S in 1 4; do ICPGPU_TILE_SPLITS=$S timeout 900 python -m pytest tests/test_gpu_tile_search.py -m gpu -x -q 2>&1 | tail -1; done
for S in 1 2 4; do echo "## splits $S"; ICPGPU_TILE_SPLITS=$S ICPGPU_TILE_SEARCH=2 timeout 600 python scripts/one_align.py 200000x200000 2>&1 | grep "tile search" | head -3 | cut -c1-260
ICPGPU_TILE_SPLITS=$S ICPGPU_TILE_SEARCH=1 timeout 600 python scripts/tile_check.py $O/x.npz 2>&1 | grep -E "^200k:|^50k:|^scan"; done
