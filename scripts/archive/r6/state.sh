#!/bin/bash
# round 6: state of the tree on the GPU (whole suite + smoke, a bench line, the pipeline's stage timers in both inner modes)
TAG=${1:-r6state}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 ) > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log
tail -30 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
python scripts/pipeline_breakdown.py 43 > $O/pipeline.txt 2>&1; cat $O/pipeline.txt
ICPGPU_GICP_INNER=quadratic python scripts/pipeline_breakdown.py 43 > $O/pipeline_quadratic.txt 2>&1; cat $O/pipeline_quadratic.txt
