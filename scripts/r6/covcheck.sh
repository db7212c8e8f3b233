#!/bin/bash
# round 6: the covariance kernels after a change -- tests, campaign slice, kernel times
TAG=${1:-r6covcheck}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py tests/test_gpu_widened_fullsize.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
timeout 900 python scripts/cov_campaign.py 0 300 2>&1 | grep -v amdgpu.ids | tail -2
export TMPDIR=/tmp; R=$PWD
cd /tmp && timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/scripts/pipeline_breakdown.py 43 > $R/$O/prof.log 2>&1
cd $R; find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/prof
grep "gicp_cov" $O/kernel_stats.csv | sed -E 's/\(anonymous namespace\):://; s/\(HIP[^"]*"/"/; s/\(int[^"]*"/"/' | cut -c1-110
grep "scans of\|device counters" $O/prof.log
