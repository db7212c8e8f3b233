#!/bin/bash
# Dev tool (round 2): correctness + per-iteration timing of the grid search for the cold-sweep start caps.
O=gpurun_out/r2h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_grid.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2; do
for r in 1 2 3; do
  echo "200k COLD_CAP=$r: $(ICPGPU_CUBE_COLD_CAP=$r python scripts/iter_profile.py 200000x200000 2>&1 | grep per-iter)"
done; done
for r in 1 2; do
  echo "50k COLD_CAP=$r: $(ICPGPU_CUBE_COLD_CAP=$r python scripts/iter_profile.py 50000x50000 2>&1 | grep per-iter)"
  echo "1M COLD_CAP=$r: $(ICPGPU_CUBE_COLD_CAP=$r python scripts/iter_profile.py 200000x1000000 2>&1 | grep per-iter)"
done
