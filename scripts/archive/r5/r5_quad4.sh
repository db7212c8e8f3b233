#!/bin/bash
TAG=${1:-r5quad4}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests/test_gpu_gicp_quadratic.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
grep -v amdgpu.ids gpurun_out/$TAG/tests.log | tail -5
for i in 1 2; do ICPGPU_GICP_INNER=quadratic timeout 600 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v amdgpu.ids >> gpurun_out/$TAG/pipeline.txt; done
cat gpurun_out/$TAG/pipeline.txt | cut -c1-300
export TMPDIR=/tmp
( cd /tmp && ICPGPU_GICP_INNER=quadratic timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -- python $GRAFT_REPO_ROOT/scripts/pipeline_breakdown.py 43 > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1 )
find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$TAG/kernel_stats_pipeline_quadratic.csv
cut -d, -f1-4 gpurun_out/$TAG/kernel_stats_pipeline_quadratic.csv | cut -c1-60,150- | head -6
rm -rf gpurun_out/$TAG/prof
