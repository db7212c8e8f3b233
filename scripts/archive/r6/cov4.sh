#!/bin/bash
# round 6: selecting covariance kernel with pruned re-collection -- tests, statistics, kernel times, A/B of the 8-wave build
TAG=${1:-r6cov4}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py tests/test_gpu_widened_fullsize.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -5 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
ICPGPU_FLAVOUR=dev ICPGPU_COV_STATS=1 timeout 120 python scripts/pipeline_breakdown.py 6 2>&1 | grep -v amdgpu.ids | tail -5 > $O/stats.txt 2>&1
cat $O/stats.txt
export TMPDIR=/tmp; R=$PWD
for v in main; do
[ $v = w8 ] && export ICPGPU_LIB_PATH=$R/icpslam_amd/csrc/build_ab/libicpgpu_w8.so
cd /tmp && ICPGPU_FLAVOUR=dev timeout 180 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- python $R/scripts/pipeline_breakdown.py 43 > $R/$O/prof_$v.log 2>&1
cd $R; find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_pipeline_$v.csv \; ; rm -rf $O/prof
echo "== $v"; grep "gicp_cov" $O/kernel_stats_pipeline_$v.csv | sed -E 's/\(anonymous namespace\):://; s/\(HIP[^"]*"/"/' | cut -c1-110
grep "scans of\|device counters" $O/prof_$v.log
done
cd /tmp && ICPGPU_FLAVOUR=dev timeout 180 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- python $R/scripts/pipeline_breakdown.py 14 > $R/$O/prof2.log 2>&1
cd $R; find $O/prof -name '*kernel_trace.csv' -exec cp {} $O/kernel_trace.csv \; ; rm -rf $O/prof
python - <<PY
import csv
for name in ('gicp_cov_select','gicp_cov_far','gicp_cov_kernel','gicp_cov_finish'):
    rows=[r for r in csv.DictReader(open('$O/kernel_trace.csv')) if name in r['Kernel_Name']]
    print(name, [round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1) for r in rows])
PY
