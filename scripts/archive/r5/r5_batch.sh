#!/bin/bash
# round 5: config-4 batch: parity tests of the batch path, then threads x depth x groups
TAG=${1:-r5batch}
mkdir -p gpurun_out/$TAG
timeout 900 python -m pytest tests/test_gpu_configs_fullsize.py tests/test_gpu_threads.py tests/test_gpu_multi.py tests/test_gpu_parity.py -m gpu -x -q -k "batch or config4 or multi or thread" > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
tail -4 gpurun_out/$TAG/tests.log
timeout 900 python scripts/r5/r5_batch_probe.py > gpurun_out/$TAG/probe.txt 2>&1
cat gpurun_out/$TAG/probe.txt
