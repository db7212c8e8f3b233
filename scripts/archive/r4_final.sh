#!/bin/bash
# Round 4: everything behind profiles/r04_* and DESIGN.md's round-4 numbers in one GPU call.
TAG=${1:-r4final}
O=gpurun_out/$TAG; mkdir -p $O
timeout 3000 python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 300 $O/bench.json; echo
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1)
for f in $O/prof/*/*kernel_stats.csv; do  # the bench process's own file (it spawns the C++ pipeline harness, traced too)
  grep -q "nn_quad_kernel<false, true" $f && grep -q nn_brute_bf16_kernel $f && cp $f $O/kernel_stats.csv
  grep -q "nn_quad_kernel<false, true" $f || cp $f $O/kernel_stats_shim_pipeline.csv
done
head -6 $O/kernel_stats.csv | cut -c1-170
python bench.py --workload 50kx50k --no-cpu-baseline > $O/bench_50k.json 2>/dev/null; head -c 200 $O/bench_50k.json; echo
python bench.py --workload 200kx1M --no-cpu-baseline > $O/bench_1M.json 2>/dev/null; head -c 200 $O/bench_1M.json; echo
python bench.py --workload batch50k --steps 20 --warmup 3 > $O/bench_batch50k.json 2> $O/bench_batch50k.err; echo "batch bench rc=$?"; head -c 300 $O/bench_batch50k.json; echo
python bench.py --multi-entry --gpus 8 --workload batch50k --steps 5 --warmup 1 > $O/bench_multi_entry_8_on_one_gpu.json 2> $O/bench_multi.err; echo "multi-entry rc=$?"; head -c 300 $O/bench_multi_entry_8_on_one_gpu.json; echo
ICPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-extras > $O/bench_2ranks_gloo.json 2> $O/bench_2ranks.err; echo "2-rank rc=$?"; head -c 300 $O/bench_2ranks_gloo.json; echo
python scripts/configs_timing.py > $O/configs.txt 2>&1; grep "^C" $O/configs.txt
{ for mode in "ICPGPU_GICP_DEVICE=0" "ICPGPU_GICP_DEVICE=1" "ICPGPU_GICP_DEVICE=0 ICPGPU_GICP_SERVER=0"; do echo "== $mode"; env $mode python scripts/pipeline_breakdown.py 43 2>&1 | grep -v "amdgpu.ids\|grid n="; done; } > $O/gicp_modes.txt 2>&1; grep "scans of\|==" $O/gicp_modes.txt
{ timeout 120 scripts/probes/solve_probe_dev 22000 5; timeout 100 scripts/probes/solve_probe_dev 5000 3; timeout 100 scripts/probes/solve_probe_dev 200000 1; } > $O/solve_probe.txt 2>&1; grep -A1 "rep 2\|rep 5" $O/solve_probe.txt | head -12
timeout 100 scripts/probes/granule_probe 200 > $O/granule_probe.txt 2>&1; tail -4 $O/granule_probe.txt
python scripts/host_scaling_probe.py 2>&1 | grep -v amdgpu.ids > $O/host_scaling.txt; tail -3 $O/host_scaling.txt
{ ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v "amdgpu.ids"; } > $O/stages.txt 2>&1; grep "scans of\|host wall per align" $O/stages.txt | cut -c1-300
