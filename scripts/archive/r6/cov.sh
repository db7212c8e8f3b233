#!/bin/bash
# round 6: the selecting covariance kernel -- its tests, the views' tests, kernel times at three cell populations
TAG=${1:-r6cov}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_views.py tests/test_gpu_voxel.py tests/test_cpp_shim.py tests/test_gpu_parity_golden.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log
for pop in 32 16 8 4; do for sel in 1 0; do echo "== knn_pop $pop select $sel"; ICPGPU_FLAVOUR=dev ICPGPU_KNN_POP=$pop ICPGPU_COV_SELECT=$sel python scripts/pipeline_breakdown.py 23 2>&1 | grep -v amdgpu.ids; done; done > $O/cov_pop.txt 2>&1
cat $O/cov_pop.txt
export TMPDIR=/tmp; R=$PWD
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/scripts/pipeline_breakdown.py 23 > $R/$O/prof.log 2>&1
cd $R; find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_pipeline.csv \; ; rm -rf $O/prof
head -25 $O/kernel_stats_pipeline.csv | cut -c1-150
