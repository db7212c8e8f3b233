#!/bin/bash
TAG=${1:-r5batch17}
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests/test_gpu_configs_fullsize.py tests/test_gpu_threads.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_sequence.py tests/test_gpu_errors.py tests/test_gpu_grid.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
tail -4 gpurun_out/$TAG/tests.log
for i in 1 2; do
python bench.py --workload batch50k --steps 20 --warmup 3 > gpurun_out/$TAG/bench_batch50k_$i.json 2> gpurun_out/$TAG/bench_batch50k_$i.err
python -c "
import json
b = json.loads(open('gpurun_out/$TAG/bench_batch50k_$i.json').read().strip().splitlines()[-1])
print('batch50k', round(b['value']), b['unit'], b['ms_per_step'], b.get('scan_pairs_per_sec'), b.get('step_ms'))"
done
ICPGPU_BATCH_THREADS=1 python bench.py --workload batch50k --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch50k one thread', round(b['value']), b['unit'], b['ms_per_step'], b.get('scan_pairs_per_sec'))"
