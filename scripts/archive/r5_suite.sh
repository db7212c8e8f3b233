#!/bin/bash
# round 5: the whole GPU suite (what the driver runs at round end) + smoke
TAG=${1:-r5suite}
mkdir -p gpurun_out/$TAG
( time timeout 2400 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/$TAG/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/gpu_tests.log
tail -8 gpurun_out/$TAG/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; tail -2 gpurun_out/$TAG/smoke.log
