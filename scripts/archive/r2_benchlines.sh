#!/bin/bash
# round 2: the bench lines of every workload (un-profiled) -> gpurun_out/<tag>/
TAG=${1:-r2lines}; O=gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --workload batch50k --steps 10 --warmup 3 > $O/bench_batch50k.json 2>/dev/null; echo "batch rc=$?"
ICPGPU_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_2ranks_gloo.json 2>/dev/null; echo "2-rank rc=$?"
python bench.py --workload 50kx50k --no-cpu-baseline > $O/bench_50k.json 2>/dev/null
python bench.py --workload 200kx1M --no-cpu-baseline > $O/bench_1M.json 2>/dev/null
for f in bench bench_batch50k bench_2ranks_gloo bench_50k bench_1M; do tail -1 $O/$f.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['value']), d['unit'], round(d['ms_per_step'],4), d.get('scan_pairs_per_sec'), d.get('scan_pairs_per_sec_e2e'), (d.get('gicp') or {}).get('scan_pairs_per_sec_e2e'), (d.get('gicp') or {}).get('reference_pipeline_scans_per_sec'), d['n_gpus'])"; done
python scripts/pipeline_breakdown.py > $O/pipeline_breakdown.txt 2>&1; grep -v amdgpu.ids $O/pipeline_breakdown.txt
python scripts/reference_pipeline_probe.py 31 > $O/reference_pipeline.txt 2>&1; grep -v amdgpu.ids $O/reference_pipeline.txt
python scripts/gicp_timing.py > $O/gicp_timing.txt 2>&1; grep -v amdgpu.ids $O/gicp_timing.txt
