#!/bin/bash
# round 6: the whole GPU suite WITHOUT -x (every failure listed), then the default bench line
TAG=${1:-r6suite}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 2400 python -m pytest tests/ -q -m gpu --durations=15 ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
tail -30 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
