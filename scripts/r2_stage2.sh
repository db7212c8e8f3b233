#!/bin/bash
# Dev tool (round 2): correctness of the stage-2 search + per-iteration timing with and without it.
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_grid.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for rep in 1 2; do
for r in 0 64; do
  echo "200k S2=$r: $(ICPGPU_S2_ROWS=$r python scripts/iter_profile.py 200000x200000 2>&1 | grep per-iter)"
done; done
for r in 0 64; do
  echo "50k S2=$r: $(ICPGPU_S2_ROWS=$r python scripts/iter_profile.py 50000x50000 2>&1 | grep per-iter)"
  echo "1M S2=$r: $(ICPGPU_S2_ROWS=$r python scripts/iter_profile.py 200000x1000000 2>&1 | grep per-iter)"
done
for d in 3 5 6; do
  echo "div $d S2=64: $(ICPGPU_GRID_DIV=$d ICPGPU_S2_ROWS=64 python scripts/iter_profile.py 200000x200000 2>&1 | grep per-iter)"
done
