#!/bin/bash
TAG=${1:-r5lines}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_mailbox.py tests/test_gpu_parity_golden.py tests/test_cpp_shim.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
tail -4 gpurun_out/$TAG/tests.log
for i in 1 2 3; do ICPGPU_GICP_DEVICE=0 python scripts/pipeline_breakdown.py 43 2>&1 | grep "scans of"; done | tee gpurun_out/$TAG/pipeline_host.txt
for i in 1 2; do python scripts/pipeline_breakdown.py 43 2>&1 | grep "scans of"; done | tee gpurun_out/$TAG/pipeline_auto.txt
ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1 ICPGPU_GICP_DEVICE=0 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v "amdgpu.ids" | tail -12 > gpurun_out/$TAG/stages.txt; cat gpurun_out/$TAG/stages.txt | cut -c1-400
