#!/bin/bash
# round 4: the device solver (gicp_solve_kernel) against the oracle, the host fallback, and the map's approx mode after the key fix
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_map.py tests/test_gpu_gicp.py tests/test_gpu_parity.py tests/test_gpu_parity_golden.py tests/test_cpp_shim.py tests/test_gpu_sequence.py -m gpu -x -q --durations=8 > gpurun_out/r4/gicp.log 2>&1
echo "rc=$?" >> gpurun_out/r4/gicp.log
tail -30 gpurun_out/r4/gicp.log
