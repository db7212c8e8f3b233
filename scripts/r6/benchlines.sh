#!/bin/bash
# round 6: the bench lines behind profiles/r06_bench_*.json (the round's last bench.py)
TAG=${1:-r6lines}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python bench.py > $O/bench_grid_200k.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 > $O/bench_grid_200k_steps20.json 2>> $O/bench.err
timeout 300 python bench.py --workload 50kx50k --iters 30 > $O/bench_50k.json 2>> $O/bench.err
timeout 300 python bench.py --workload 200kx1M --iters 30 --steps 10 --warmup 3 > $O/bench_200k_1M.json 2>> $O/bench.err
timeout 300 python bench.py --workload batch50k > $O/bench_batch50k.json 2>> $O/bench.err
timeout 300 python bench.py --workload batch50k --multi-entry --gpus 8 > $O/bench_multi_entry_8_entries_one_gpu.json 2>> $O/bench.err
ICPGPU_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --no-extras > $O/bench_2ranks_gloo_one_gpu.json 2>> $O/bench.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print(f.split('/')[-1], round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'frac', r.get('frac'), 'steady', (r.get('steady_state') or {}).get('iterations_per_sec'))
    g=d.get('gicp')
    if g: print('   pipeline', round(g['reference_pipeline_scans_per_sec'],1), 'shim', g['shim_pipeline_scans_per_sec'], 'one thread', g['shim_pipeline'].get('scans_per_sec_one_thread'), 'quadratic', round(g['quadratic_inner']['reference_pipeline_scans_per_sec'],1), 'shim q', g['shim_pipeline'].get('scans_per_sec_quadratic_inner'), 'e2e', round(d['scan_pairs_per_sec_e2e'],1))
PY
