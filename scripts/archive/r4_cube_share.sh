#!/bin/bash
# how much of a sweep is the cube stage, per workload?  (development flavour: ICPGPU_SKIP_UNCERT drops the uncertified points -- WRONG results, timing only)
for wl in 200kx200k 50kx50k 200kx1M; do
  for skip in 0 1; do
    if [ $skip = 1 ]; then export ICPGPU_SKIP_UNCERT=1; else unset ICPGPU_SKIP_UNCERT; fi
    ICPGPU_FLAVOUR=dev python bench.py --workload $wl --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl', 'skip_uncert=$skip', 'it/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'kernel us', round(1e3*d['roofline'].get('avg_launch_ms',0),2))"
  done
done
