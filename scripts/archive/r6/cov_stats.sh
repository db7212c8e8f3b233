#!/bin/bash
TAG=${1:-r6covstats}
O=gpurun_out/$TAG; mkdir -p $O
for pop in 32 16 8; do echo "== knn_pop $pop"; ICPGPU_FLAVOUR=dev ICPGPU_KNN_POP=$pop ICPGPU_COV_STATS=1 python scripts/pipeline_breakdown.py 8 2>&1 | grep -v amdgpu.ids | tail -9; done > $O/stats.txt 2>&1
cat $O/stats.txt
