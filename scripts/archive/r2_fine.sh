#!/bin/bash
# Dev tool (round 2): would a FINE second grid pay on the bounded (late) sweeps?  Per-iteration kernel time with much smaller cells.
for d in 4.5 6 8 10 14 20; do
  h=$(ICPGPU_DEBUG=1 ICPGPU_GRID_DIV=$d python scripts/iter_profile.py 200000x200000 2>&1 | grep -m1 "grid n=" | sed 's/.*h=\([0-9.]*\).*pop=\([0-9.]*\).*/h=\1 pop=\2/')
  echo "200k div $d ($h): $(ICPGPU_GRID_DIV=$d python scripts/iter_profile.py 200000x200000 2>&1 | grep per-iter)"
done
for d in 4.5 8 14; do
  echo "50k div $d: $(ICPGPU_GRID_DIV=$d python scripts/iter_profile.py 50000x50000 2>&1 | grep per-iter)"
done
