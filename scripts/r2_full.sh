#!/bin/bash
# Dev tool (round 2): the whole GPU suite, the default bench, the kernel trace of the same command, issue counters.
TAG=${1:-r2full}
O=gpurun_out/$TAG; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 1500 $O/bench.json; echo
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
head -12 $O/kernel_stats.csv | cut -c1-200
bash scripts/pmc_issue.sh $TAG/issue 200000x200000 50000x50000
