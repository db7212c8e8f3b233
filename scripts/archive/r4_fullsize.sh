#!/bin/bash
# round 4: configs 4 / 5 at full size under -m gpu (tests/test_gpu_configs_fullsize.py), with durations
mkdir -p gpurun_out/r4
nproc > gpurun_out/r4/nproc.txt
timeout 1500 python -m pytest tests/test_gpu_configs_fullsize.py -m gpu -x -q --durations=10 > gpurun_out/r4/fullsize.log 2>&1
echo "rc=$?" >> gpurun_out/r4/fullsize.log
tail -30 gpurun_out/r4/fullsize.log
