#!/bin/bash
# Dev tool: the standard GPU-box sequence (tests, bench, rocprofv3 kernel trace). Outputs under gpurun_out/<tag>/.
TAG=${1:-run}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 5 --no-cpu-baseline > $O/prof.log 2>&1
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
cat $O/kernel_stats.csv | cut -c1-160
