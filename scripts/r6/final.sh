#!/bin/bash
# round 6: everything behind profiles/r06_* in one GPU call (every step under its own timeout).  Outputs under gpurun_out/<tag>/.
TAG=${1:-r6final}
R=$PWD
O=$R/gpurun_out/$TAG; mkdir -p $O /tmp/shim
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -q -m gpu --durations=10 ) > $O/gpu_suite.txt 2>&1; echo "rc=$?" >> $O/gpu_suite.txt
tail -4 $O/gpu_suite.txt
timeout 600 python bench.py > $O/bench_grid_200k.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 > $O/bench_grid_200k_steps20.json 2>> $O/bench.err
timeout 300 python bench.py --workload 50kx50k --iters 30 > $O/bench_50k.json 2>> $O/bench.err
timeout 300 python bench.py --workload 200kx1M --iters 30 --steps 10 --warmup 3 > $O/bench_200k_1M.json 2>> $O/bench.err
timeout 300 python bench.py --workload batch50k > $O/bench_batch50k.json 2>> $O/bench.err
timeout 300 python bench.py --workload batch50k --multi-entry --gpus 8 > $O/bench_multi_entry_8_entries_one_gpu.json 2>> $O/bench.err
ICPGPU_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --no-extras > $O/bench_2ranks_gloo_one_gpu.json 2>> $O/bench.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a -- python $R/bench.py --steps 20 --no-cpu-baseline --no-extras > $O/prof_a.log 2>&1
find $O/prof_a -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_grid_200k.csv \; ; rm -rf $O/prof_a
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -- python $R/scripts/pipeline_breakdown.py 43 > $O/prof_b.log 2>&1
find $O/prof_b -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_pipeline.csv \; ; rm -rf $O/prof_b
ICPGPU_GICP_INNER=quadratic timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c -- python $R/scripts/pipeline_breakdown.py 43 > $O/prof_c.log 2>&1
find $O/prof_c -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_pipeline_quadratic.csv \; ; rm -rf $O/prof_c
cd $R
g++ -std=c++14 -O2 -DICPGPU_SHIM_TIMING -I include tests/cpp/odometer_pipeline_demo.cpp -o /tmp/shim/demo -L icpslam_amd -licpgpu -Wl,-rpath,$PWD/icpslam_amd -Wl,-rpath,/opt/rocm/lib -pthread
python - <<'PY'
import sys; sys.path.insert(0, '.')
from icpslam_amd import synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
a.tofile('/tmp/shim/a.bin'); b.tofile('/tmp/shim/b.bin')
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d -- /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 54 0.2 10 4 4 > $O/prof_d.log 2>&1
find $O/prof_d -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_shim_pipeline.csv \; ; rm -rf $O/prof_d
cd $R
{
echo "== scripts/pipeline_breakdown.py 43 (exact inner solver), twice"; for i in 1 2; do timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v amdgpu.ids; done
echo "== the same, ICPGPU_GICP_INNER=quadratic"; ICPGPU_GICP_INNER=quadratic timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v amdgpu.ids
echo "== development flavour, the round-5 ways: ICPGPU_COV_SELECT=0 (streaming covariance kernel), ICPGPU_VOXEL_PLANNED=0 (filter waits for the box), ICPGPU_STAGE_DIRECT=0 (copy engine), ICPGPU_COV_GRID_UNCHECKED=0"
ICPGPU_FLAVOUR=dev ICPGPU_COV_SELECT=0 ICPGPU_VOXEL_PLANNED=0 ICPGPU_STAGE_DIRECT=0 ICPGPU_COV_GRID_UNCHECKED=0 timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v amdgpu.ids
echo "== development flavour, ICPGPU_GICP_TIMING=1: host wall per stage of align_gicp (after the warm-up), resident loop on the bench pair"
ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1 timeout 120 python scripts/r5_pipeline_on_bench_pair.py 2>&1 | grep "GICP alignments\|scans/s"
echo "== development flavour, ICPGPU_COV_STATS=1: what the selecting covariance kernel did with three clouds"
ICPGPU_FLAVOUR=dev ICPGPU_COV_STATS=1 timeout 120 python scripts/pipeline_breakdown.py 6 2>&1 | grep "covariances of\|handed over\|cube radius" | tail -9
echo "== the callback as integrated (tests/cpp/odometer_pipeline_demo.cpp, -DICPGPU_SHIM_TIMING), 4 threads then 1"
for th in 4 4 1 1; do ICPGPU_DEMO_TIMING=1 timeout 60 /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 104 0.2 10 $th 4 2>&1 | grep "TIMING\|STAGES\|SHIM"; done
echo "== scripts/r5_shim_breakdown.py (the shim's C-ABI calls from Python, one by one)"; SCANS=100 timeout 120 python scripts/r5_shim_breakdown.py 2>&1 | grep -v amdgpu.ids
} > $O/gicp_pipeline.txt 2>&1
timeout 600 python scripts/hbm_kernels.py > $O/hbm_kernels.txt 2>&1
timeout 900 python scripts/configs_timing.py > $O/configs.txt 2>&1
echo done
