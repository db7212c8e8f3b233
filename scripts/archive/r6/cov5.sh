#!/bin/bash
TAG=${1:-r6cov5}
O=gpurun_out/$TAG; mkdir -p $O
ICPGPU_FLAVOUR=dev ICPGPU_COV_STATS=1 timeout 120 python scripts/pipeline_breakdown.py 12 2>&1 | grep -v amdgpu.ids > $O/stats.txt 2>&1
cat $O/stats.txt
export TMPDIR=/tmp; R=$PWD
cd /tmp && ICPGPU_FLAVOUR=dev timeout 180 rocprofv3 --kernel-trace --output-format csv -d $R/$O/prof -- python $R/scripts/pipeline_breakdown.py 12 > $R/$O/prof.log 2>&1
cd $R; find $O/prof -name '*kernel_trace.csv' -exec cp {} $O/kernel_trace.csv \; ; rm -rf $O/prof
python - <<PY
import csv
rows=[r for r in csv.DictReader(open('$O/kernel_trace.csv')) if 'gicp_cov_select' in r['Kernel_Name']]
print([round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,1) for r in rows])
PY
