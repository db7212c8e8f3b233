#!/bin/bash
# Dev tool (round 2): the whole GPU suite, the default bench, the kernel trace of the same command, issue + traffic counters,
# the five configs, the batch workload and the self-launched two-rank run.
TAG=${1:-r2full}
O=gpurun_out/$TAG; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 600 $O/bench.json; echo
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
bash scripts/pmc_issue.sh $TAG/issue 200000x200000 50000x50000 > $O/issue.log 2>&1; tail -3 $O/issue.log
bash scripts/pmc_round.sh $TAG 200000x200000 > $O/traffic.log 2>&1; tail -5 $O/traffic.log
python scripts/configs_timing.py > $O/configs.txt 2>&1; grep "^C" $O/configs.txt
python bench.py --workload batch50k --steps 10 --warmup 3 > $O/bench_batch50k.json 2> $O/bench_batch50k.err; echo "batch bench rc=$?"; head -c 900 $O/bench_batch50k.json; echo
ICPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks.err; echo "2-rank rc=$?"; head -c 400 $O/bench_2ranks_gloo.json; echo
python bench.py --workload 50kx50k --no-cpu-baseline > $O/bench_50k.json 2>/dev/null; head -c 300 $O/bench_50k.json; echo
python bench.py --workload 200kx1M --no-cpu-baseline > $O/bench_1M.json 2>/dev/null; head -c 300 $O/bench_1M.json; echo
python scripts/hbm_kernels.py > $O/hbm_kernels.txt 2>&1; tail -12 $O/hbm_kernels.txt
