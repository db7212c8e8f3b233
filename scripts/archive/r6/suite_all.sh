#!/bin/bash
# round 6: the whole GPU suite WITHOUT -x (every failure listed), smoke, then the default bench line
TAG=${1:-r6suite}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests/ -q -m gpu --durations=15 ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
tail -30 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; tail -3 $O/bench.err
