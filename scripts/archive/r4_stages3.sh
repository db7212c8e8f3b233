#!/bin/bash
# Dev tool (round 4): GICP / voxel / map / shim tests, then the pipeline's stages with and without the covariance-grid hint
mkdir -p gpurun_out/r4s
timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py tests/test_gpu_sequence.py tests/test_gpu_map.py tests/test_cpp_shim.py tests/test_gpu_voxel.py tests/test_gpu_grid.py tests/test_gpu_recognition.py -x -q -m gpu > gpurun_out/r4s/tests3.log 2>&1
echo "rc=$?" >> gpurun_out/r4s/tests3.log
export ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1
for k in 1 0 1 0; do
  echo "== ICPGPU_KNN_HINT=$k" >> gpurun_out/r4s/stages3.txt
  ICPGPU_KNN_HINT=$k timeout 300 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v "amdgpu.ids\|icpgpu\]   \|per workgroup\|GICP evaluations" >> gpurun_out/r4s/stages3.txt
done
