#!/bin/bash
# Dev tool (round 4): GICP tests + pipeline stages with the covariance grid adopted for the search (and, A/B, without)
mkdir -p gpurun_out/r4s
timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py tests/test_gpu_sequence.py tests/test_gpu_map.py tests/test_cpp_shim.py -x -q -m gpu > gpurun_out/r4s/adopt_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4s/adopt_tests.log
export ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1
for a in 1 0 1 0; do
  echo "== ICPGPU_GICP_ADOPT_GRID=$a" >> gpurun_out/r4s/adopt.txt
  ICPGPU_GICP_ADOPT_GRID=$a timeout 300 python scripts/pipeline_breakdown.py 43 >> gpurun_out/r4s/adopt.txt 2>&1
done
