#!/bin/bash
# round 3: the suite + the bench line on the final build (the long version: scripts/r3_final.sh)
O=gpurun_out/r3final3; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench.json; echo
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
for f in $O/prof/*/*kernel_stats.csv; do
  grep -q "nn_quad_kernel<false, true" $f && grep -q nn_brute_bf16_kernel $f && cp $f $O/kernel_stats.csv
done
head -8 $O/kernel_stats.csv | cut -c1-160
