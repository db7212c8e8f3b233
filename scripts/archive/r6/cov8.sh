#!/bin/bash
# round 6: far-field kernel with key caps, no streaming launch for clouds up to 64k points, grid-less covariances; bbox init elided
TAG=${1:-r6cov8}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py tests/test_gpu_widened_fullsize.py tests/test_gpu_voxel.py tests/test_cpp_shim.py tests/test_gpu_recognition.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -8 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
timeout 900 python scripts/cov_campaign.py 0 300 2>&1 | grep -v amdgpu.ids | tail -4 > $O/cov_campaign.txt; cat $O/cov_campaign.txt
for i in 1 2; do timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v amdgpu.ids; done > $O/pipeline.txt; cat $O/pipeline.txt
ICPGPU_GICP_INNER=quadratic timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v amdgpu.ids > $O/pipeline_quadratic.txt; cat $O/pipeline_quadratic.txt
export TMPDIR=/tmp; R=$PWD
cd /tmp && timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/scripts/pipeline_breakdown.py 43 > $R/$O/prof.log 2>&1
cd $R; find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_pipeline.csv \; ; rm -rf $O/prof
grep "gicp_cov\|bbox\|voxel_\|fillBuffer" $O/kernel_stats_pipeline.csv | sed -E 's/\(anonymous namespace\):://; s/\(HIP[^"]*"/"/; s/\(int[^"]*"/"/' | cut -c1-110
