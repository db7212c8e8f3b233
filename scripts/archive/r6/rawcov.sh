#!/bin/bash
# round 6: the covariance pass on RAW (unfiltered) clouds, selecting against streaming kernel
TAG=${1:-r6rawcov}
O=gpurun_out/$TAG; mkdir -p $O
for sel in 1 0; do echo "== ICPGPU_COV_SELECT=$sel"; ICPGPU_FLAVOUR=dev ICPGPU_COV_SELECT=$sel ICPGPU_COV_STATS=$STATS timeout 300 python scripts/gicp_timing.py 5000x5000 50000x50000 200000x200000 2>&1 | grep -v amdgpu.ids | cut -c1-400; done > $O/rawcov.txt 2>&1
cat $O/rawcov.txt
