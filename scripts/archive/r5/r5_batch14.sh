#!/bin/bash
TAG=${1:-r5batch14}
mkdir -p gpurun_out/$TAG
C="4x8x8 2x8x8 1x8x8"
for st in 1 0; do
echo "== ICPGPU_BATCH_STAGGER=$st" >> gpurun_out/$TAG/probe.txt
ICPGPU_FLAVOUR=dev ICPGPU_BATCH_STAGGER=$st timeout 900 python scripts/r5/r5_batch_probe.py $C >> gpurun_out/$TAG/probe.txt 2>&1
done
echo "== GPU_MAX_HW_QUEUES=8" >> gpurun_out/$TAG/probe.txt
GPU_MAX_HW_QUEUES=8 timeout 900 python scripts/r5/r5_batch_probe.py $C >> gpurun_out/$TAG/probe.txt 2>&1
cat gpurun_out/$TAG/probe.txt | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_configs_fullsize.py tests/test_gpu_threads.py tests/test_gpu_multi.py tests/test_gpu_parity.py tests/test_gpu_sequence.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1; echo "rc=$?" >> gpurun_out/$TAG/tests.log
tail -4 gpurun_out/$TAG/tests.log
python bench.py --workload batch50k > gpurun_out/$TAG/bench_batch50k.json 2> gpurun_out/$TAG/bench_batch50k.err
python -c "
import json
b = json.loads(open('gpurun_out/$TAG/bench_batch50k.json').read().strip().splitlines()[-1])
print('batch50k', round(b['value']), b['unit'], b['ms_per_step'], b.get('scan_pairs_per_sec'))"
