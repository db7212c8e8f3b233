#!/bin/bash
# round 6: the selecting covariance kernel -- tests, statistics, kernel times (every step under its own timeout)
TAG=${1:-r6cov2}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py tests/test_gpu_widened_fullsize.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -8 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
for pop in 32 16; do echo "== knn_pop $pop"; ICPGPU_FLAVOUR=dev ICPGPU_KNN_POP=$pop ICPGPU_COV_STATS=1 timeout 120 python scripts/pipeline_breakdown.py 6 2>&1 | grep -v amdgpu.ids | tail -5; done > $O/stats.txt 2>&1
cat $O/stats.txt
export TMPDIR=/tmp; R=$PWD
for pop in 32 16; do
cd /tmp && ICPGPU_FLAVOUR=dev ICPGPU_KNN_POP=$pop timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/scripts/pipeline_breakdown.py 23 > $R/$O/prof_$pop.log 2>&1
cd $R; find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_pipeline_pop$pop.csv \; ; rm -rf $O/prof
echo "== pop $pop"; grep "gicp_cov\|grid_\|scan_" $O/kernel_stats_pipeline_pop$pop.csv | sed -E 's/\(anonymous namespace\):://; s/\(HIP[^"]*"/"/' | cut -c1-110
tail -3 $O/prof_$pop.log
done
