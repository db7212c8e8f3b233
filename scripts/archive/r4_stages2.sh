#!/bin/bash
# Dev tool (round 4): A/B of two builds of the development library on the reference's pipeline (same box, alternating)
mkdir -p gpurun_out/r4s
timeout 600 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py -x -q -m gpu > gpurun_out/r4s/gicp_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r4s/gicp_tests.log
export ICPGPU_GICP_TIMING=1
for k in new base new base new base; do
  echo "== $k" >> gpurun_out/r4s/stages2.txt
  if [ $k = base ]; then export ICPGPU_LIB_PATH=$PWD/icpslam_amd/libicpgpu_base_dev.so; else export ICPGPU_LIB_PATH=$PWD/icpslam_amd/libicpgpu_dev.so; fi
  timeout 300 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v "amdgpu.ids\|device counters\|host wall per scan\|icpgpu\]   \|per workgroup" >> gpurun_out/r4s/stages2.txt
done
