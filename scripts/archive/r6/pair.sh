#!/bin/bash
# round 6: two points per wave in the selecting covariance kernel -- tests, A/B timing
TAG=${1:-r6pair}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py tests/test_gpu_widened_fullsize.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -8 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
timeout 900 python scripts/cov_campaign.py 0 300 2>&1 | grep -v amdgpu.ids | tail -4 > $O/cov_campaign.txt; cat $O/cov_campaign.txt
export TMPDIR=/tmp; R=$PWD
for pair in 1 0; do
cd /tmp && ICPGPU_FLAVOUR=dev ICPGPU_COV_PAIR=$pair timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/scripts/pipeline_breakdown.py 43 > $R/$O/prof_$pair.log 2>&1
cd $R; find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_pair$pair.csv \; ; rm -rf $O/prof
echo "== ICPGPU_COV_PAIR=$pair"; grep "gicp_cov" $O/kernel_stats_pair$pair.csv | sed -E 's/\(anonymous namespace\):://; s/\(HIP[^"]*"/"/; s/\(int[^"]*"/"/' | cut -c1-110
grep "scans of\|device counters" $O/prof_$pair.log
done
for pair in 1 0; do echo "== raw clouds, ICPGPU_COV_PAIR=$pair"; ICPGPU_FLAVOUR=dev ICPGPU_COV_PAIR=$pair timeout 300 python scripts/gicp_timing.py 50000x50000 200000x200000 2>&1 | grep -v amdgpu.ids | cut -c1-200; done
