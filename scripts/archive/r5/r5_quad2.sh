#!/bin/bash
# round 5: quadratic inner solver -- tests incl. batches, kernel trace of the pipeline, GICP batches with it
TAG=${1:-r5quad2}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests/test_gpu_gicp_quadratic.py -m gpu -x -q -s > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
grep -v amdgpu.ids gpurun_out/$TAG/tests.log | tail -12
for mode in quadratic; do
  echo "## ICPGPU_GICP_INNER=$mode" >> gpurun_out/$TAG/pipeline.txt
  ICPGPU_GICP_INNER=$mode timeout 600 python scripts/pipeline_breakdown.py 43 >> gpurun_out/$TAG/pipeline.txt 2>&1
  ICPGPU_GICP_INNER=$mode ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1 timeout 600 python scripts/pipeline_breakdown.py 43 2>&1 | grep "GICP alignments" >> gpurun_out/$TAG/pipeline.txt
done
grep -v amdgpu.ids gpurun_out/$TAG/pipeline.txt | cut -c1-600
export TMPDIR=/tmp
( cd /tmp && ICPGPU_GICP_INNER=quadratic timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof -- python $GRAFT_REPO_ROOT/scripts/pipeline_breakdown.py 43 > $GRAFT_REPO_ROOT/gpurun_out/$TAG/prof.log 2>&1 )
find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$TAG/kernel_stats_pipeline_quadratic.csv
cut -d, -f1-7 gpurun_out/$TAG/kernel_stats_pipeline_quadratic.csv | head -16
find gpurun_out/$TAG/prof -name "*.csv" ! -name "*stats*" -delete
for mode in exact quadratic; do
  echo "## GICP batches, ICPGPU_GICP_INNER=$mode" >> gpurun_out/$TAG/batch.txt
  ICPGPU_GICP_INNER=$mode timeout 900 python scripts/r5_gicp_batch_probe.py 0x0 1x8 1x16 2x8 4x4 >> gpurun_out/$TAG/batch.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/$TAG/batch.txt
