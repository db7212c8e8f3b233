#!/bin/bash
# Dev tool (round 2): matrix-core brute-force kernel: correctness, then variants.
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_grid.py tests/test_gpu_fullsize.py tests/test_gpu_map.py tests/test_gpu_errors.py -x -q -m gpu 2>&1 | tail -4
for g in 2 4; do for w in 8 16 32; do
  echo "G=$g WAVES=$w: $(ICPGPU_MFMA_G=$g ICPGPU_MFMA_WAVES=$w python scripts/brute_timing.py 200000x200000 2>&1 | grep 'matrix cores')"
done; done
python scripts/brute_timing.py 50000x50000 200000x1000000 2>&1 | grep -v amdgpu.ids
