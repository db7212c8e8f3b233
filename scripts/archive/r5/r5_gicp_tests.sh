#!/bin/bash
TAG=${1:-r5gicpt}
mkdir -p gpurun_out/$TAG
timeout 1800 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity.py tests/test_gpu_parity_golden.py tests/test_gpu_sequence.py tests/test_gpu_threads.py tests/test_cpp_shim.py tests/test_gpu_recognition.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
tail -4 gpurun_out/$TAG/tests.log
CHECK=1 timeout 600 python scripts/r5_gicp_batch_probe.py 0x0 2x4 1x8 > gpurun_out/$TAG/probe.txt 2>&1; grep -v amdgpu gpurun_out/$TAG/probe.txt | tail -4
