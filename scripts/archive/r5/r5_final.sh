#!/bin/bash
# Round 5: everything behind profiles/r05_* and DESIGN.md's round-5 numbers in one GPU call.
TAG=${1:-r5final}
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 3000 python -m pytest tests -x -q -m gpu --durations=12 ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench.json; echo
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
for f in $O/prof/*/*kernel_stats.csv; do  # the bench process's own file (it spawns the C++ pipeline harness, traced too)
  grep -q "nn_quad_kernel<false, true" $f && grep -q nn_brute_bf16_kernel $f && cp $f $O/kernel_stats.csv
  grep -q "nn_quad_kernel<false, true" $f || cp $f $O/kernel_stats_shim_pipeline.csv
done
rm -rf $O/prof
head -6 $O/kernel_stats.csv | cut -c1-170
python bench.py --workload 50kx50k --no-cpu-baseline > $O/bench_50k.json 2>/dev/null; head -c 200 $O/bench_50k.json; echo
python bench.py --workload 200kx1M --no-cpu-baseline > $O/bench_1M.json 2>/dev/null; head -c 200 $O/bench_1M.json; echo
python bench.py --workload batch50k --steps 20 --warmup 3 > $O/bench_batch50k.json 2> $O/bench_batch50k.err; echo "batch bench rc=$?"; head -c 300 $O/bench_batch50k.json; echo
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/profb -- python $GRAFT_REPO_ROOT/bench.py --workload batch50k --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/profb.log 2>&1)
for f in $O/profb/*/*kernel_stats.csv; do cp $f $O/kernel_stats_batch50k.csv; done; rm -rf $O/profb
head -4 $O/kernel_stats_batch50k.csv | cut -c1-170
python bench.py --multi-entry --gpus 8 --workload batch50k --steps 5 --warmup 1 > $O/bench_multi_entry_8_on_one_gpu.json 2> $O/bench_multi.err; echo "multi-entry rc=$?"; head -c 300 $O/bench_multi_entry_8_on_one_gpu.json; echo
ICPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-extras > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks.err; echo "2-rank rc=$?"; head -c 300 $O/bench_2ranks_gloo.json; echo
python scripts/configs_timing.py > $O/configs.txt 2>&1; grep "^C" $O/configs.txt
