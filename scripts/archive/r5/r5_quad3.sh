#!/bin/bash
TAG=${1:-r5quad3}
mkdir -p gpurun_out/$TAG
timeout 1500 python scripts/r5/quadratic_campaign.py 400 61 > gpurun_out/$TAG/campaign.txt 2>&1
grep -v amdgpu.ids gpurun_out/$TAG/campaign.txt
timeout 600 python -m pytest tests/test_cpp_shim.py tests/test_gpu_gicp.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
grep -v amdgpu.ids gpurun_out/$TAG/tests.log | tail -5
