#!/bin/bash
# A/B of an experimental library build against the shipped one: bench.py headline + 50k + 200kx1M, two rounds each (alternating)
mkdir -p gpurun_out/r4
X=${1:-icpslam_amd/libicpgpu_rows4.so}
for round in 1 2; do
for lib in "" "$X"; do
  for wl in 200kx200k 50kx50k 200kx1M; do
    ICPGPU_LIB_PATH=$lib python bench.py --workload $wl --no-extras --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${lib:-shipped}', '$wl', 'it/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel us', round(1e3*d['roofline'].get('avg_launch_ms',0),2))"
  done
done
done
