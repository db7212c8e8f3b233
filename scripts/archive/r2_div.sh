#!/bin/bash
# Dev tool (round 2): cell-size sweep of the grid search (prints the cell size the build actually chose).
for d in 2 2.5 3 3.5 4 4.5 5; do
  h=$(ICPGPU_DEBUG=1 ICPGPU_GRID_DIV=$d python scripts/iter_profile.py 200000x200000 2>&1 | grep -m1 "grid n=" | sed 's/.*h=\([0-9.]*\).*pop=\([0-9.]*\).*/h=\1 pop=\2/')
  for rep in 1 2; do
  echo "200k div $d ($h): $(ICPGPU_GRID_DIV=$d python scripts/iter_profile.py 200000x200000 2>&1 | grep per-iter)"
  done
done
for d in 3 4; do
  echo "50k div $d: $(ICPGPU_GRID_DIV=$d python scripts/iter_profile.py 50000x50000 2>&1 | grep per-iter)"
  echo "1M div $d: $(ICPGPU_GRID_DIV=$d python scripts/iter_profile.py 200000x1000000 2>&1 | grep per-iter)"
done
