#!/bin/bash
# round 6, first GPU call: the new full-size tests of the widened rows, a bench line and the pipeline's stage timers (baseline for the round)
TAG=${1:-r6first}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 1500 python -m pytest tests/test_gpu_widened_fullsize.py -x -q -m gpu -s --durations=10 ) > $O/widened.log 2>&1; echo "rc=$?" >> $O/widened.log
tail -25 $O/widened.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
python scripts/pipeline_breakdown.py 43 > $O/pipeline.txt 2>&1; cat $O/pipeline.txt
ICPGPU_GICP_INNER=quadratic python scripts/pipeline_breakdown.py 43 > $O/pipeline_quadratic.txt 2>&1; cat $O/pipeline_quadratic.txt
