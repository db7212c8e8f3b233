#!/bin/bash
# round 6: the whole GPU suite (what the driver runs at round end) + smoke
TAG=${1:-r6suite}
mkdir -p gpurun_out/$TAG
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/$TAG/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/gpu_tests.log
tail -30 gpurun_out/$TAG/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$TAG/smoke.log 2>&1; tail -2 gpurun_out/$TAG/smoke.log
