#!/bin/bash
for q in 16 12 8; do for rep in 1 2; do echo "QPW=$q: $(ICPGPU_QPW=$q python scripts/iter_profile.py 200000x200000 2>&1 | grep per-iter)"; done; done
for q in 16 8 4; do echo "50k QPW=$q: $(ICPGPU_QPW=$q python scripts/iter_profile.py 50000x50000 2>&1 | grep per-iter)"; done
