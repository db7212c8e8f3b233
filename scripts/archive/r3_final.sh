#!/bin/bash
# Round 3: everything behind profiles/r03_* and DESIGN.md's round-3 numbers in one GPU call.
TAG=${1:-r3final}
O=gpurun_out/$TAG; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 400 $O/bench.json; echo
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
for f in $O/prof/*/*kernel_stats.csv; do  # the bench process's own file (it spawns the C++ pipeline harness, traced too)
  grep -q "nn_quad_kernel<false, true" $f && grep -q nn_brute_bf16_kernel $f && cp $f $O/kernel_stats.csv
  grep -q "nn_quad_kernel<false, true" $f || cp $f $O/kernel_stats_shim_pipeline.csv
done
bash scripts/pmc_issue.sh $TAG/issue 200000x200000 50000x50000 > $O/issue.log 2>&1; tail -3 $O/issue.log
bash scripts/pmc_round.sh $TAG 200000x200000 > $O/traffic.log 2>&1; tail -5 $O/traffic.log
python scripts/configs_timing.py > $O/configs.txt 2>&1; grep "^C" $O/configs.txt
python bench.py --workload batch50k --steps 10 --warmup 3 > $O/bench_batch50k.json 2> $O/bench_batch50k.err; echo "batch bench rc=$?"; head -c 300 $O/bench_batch50k.json; echo
ICPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-extras > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks.err; echo "2-rank rc=$?"; head -c 300 $O/bench_2ranks_gloo.json; echo
python bench.py --workload 50kx50k --no-cpu-baseline > $O/bench_50k.json 2>/dev/null; head -c 200 $O/bench_50k.json; echo
python bench.py --workload 200kx1M --no-cpu-baseline > $O/bench_1M.json 2>/dev/null; head -c 200 $O/bench_1M.json; echo
python scripts/hbm_kernels.py > $O/hbm_kernels.txt 2>&1; tail -12 $O/hbm_kernels.txt
python scripts/reference_pipeline_probe.py 31 > $O/refpipe.txt 2>&1; tail -8 $O/refpipe.txt
ICPGPU_GICP_TIMING=1 python scripts/pipeline_breakdown.py 31 > $O/breakdown.txt 2>&1; tail -5 $O/breakdown.txt
python scripts/map_timing.py > $O/map.txt 2>&1; tail -8 $O/map.txt
python scripts/brute_timing.py > $O/brute.txt 2>&1; tail -6 $O/brute.txt
