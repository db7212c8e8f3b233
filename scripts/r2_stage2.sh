#!/bin/bash
# Dev tool (round 2): correctness + per-iteration timing of the grid search for the cube-start modes.
O=gpurun_out/r2e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_grid.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py -x -q -m gpu -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|within|flavour|same outer" $O/pytest.log | tail -20
for rep in 1 2; do
for r in 0 1 3; do
  echo "200k CUBE_START=$r: $(ICPGPU_CUBE_START=$r python scripts/iter_profile.py 200000x200000 2>&1 | grep per-iter)"
done; done
for r in 0 1; do
  echo "50k CUBE_START=$r: $(ICPGPU_CUBE_START=$r python scripts/iter_profile.py 50000x50000 2>&1 | grep per-iter)"
  echo "1M CUBE_START=$r: $(ICPGPU_CUBE_START=$r python scripts/iter_profile.py 200000x1000000 2>&1 | grep per-iter)"
done
for d in 3 5 6; do
  echo "div $d: $(ICPGPU_GRID_DIV=$d python scripts/iter_profile.py 200000x200000 2>&1 | grep per-iter)"
done
