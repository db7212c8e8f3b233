#!/bin/bash
# round 4: the whole GPU suite with durations (after the unit split, the two library flavours and the correctly rounded trig contract)
mkdir -p gpurun_out/r4
timeout 1700 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r4/suite.log 2>&1
echo "rc=$?" >> gpurun_out/r4/suite.log
tail -40 gpurun_out/r4/suite.log
