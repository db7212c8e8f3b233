#!/bin/bash
# round 5: GICP's quadratic inner solver -- tests, then the reference pipeline with and without it
TAG=${1:-r5quad}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests/test_gpu_gicp_quadratic.py -m gpu -x -q -s > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
grep -v amdgpu.ids gpurun_out/$TAG/tests.log | tail -25
for mode in exact quadratic; do
  echo "## ICPGPU_GICP_INNER=$mode" >> gpurun_out/$TAG/pipeline.txt
  ICPGPU_GICP_INNER=$mode timeout 600 python scripts/pipeline_breakdown.py 43 >> gpurun_out/$TAG/pipeline.txt 2>&1
  ICPGPU_GICP_INNER=$mode ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1 timeout 600 python scripts/pipeline_breakdown.py 43 >> gpurun_out/$TAG/pipeline.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/$TAG/pipeline.txt | cut -c1-600
