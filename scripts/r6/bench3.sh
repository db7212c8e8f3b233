#!/bin/bash
# round 6: the driver's bench line (--steps 20) three times: timed region against the steady-state loop behind it
TAG=${1:-r6bench3}
O=gpurun_out/$TAG; mkdir -p $O
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err; done
python - <<PY
import json
for i in (1,2,3):
    d=json.loads(open('$O/bench_%d.json'%i).read().strip().splitlines()[-1]); r=d['roofline']
    print(i, 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'steady', round(r['steady_state']['iterations_per_sec'],1), 'dense kernel ms', round(r['avg_launch_ms'],5), 'sampled', round(r['sampled_in_timed_region']['avg_launch_ms'],5))
PY
