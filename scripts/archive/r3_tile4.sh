#!/bin/bash
for S in 1; do echo "## splits $S, no exact path"; ICPGPU_TILE_NO_EXACT=1 ICPGPU_TILE_SPLITS=$S ICPGPU_TILE_SEARCH=2 timeout 600 python scripts/one_align.py 200000x200000 2>&1 | grep "tile search" | head -3 | cut -c1-260; done
