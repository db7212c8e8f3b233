#!/bin/bash
# Dev tool (round 2): the whole GPU suite after the scheduler refactor, search timing, batch throughput / jitter.
O=gpurun_out/r2g; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for w in 200000x200000 50000x50000 200000x1000000; do echo "$w: $(python scripts/iter_profile.py $w 2>&1 | grep per-iter)"; done
python scripts/batch_jitter.py auto,1x8,2x4,4x2,8x1,4x4 2>&1 | grep -v amdgpu.ids
echo "--- 8 processes sharing the box (2 CPUs each via taskset), auto setting, LOCAL_WORLD_SIZE=8"
for i in 0 1 2 3 4 5 6 7; do LOCAL_WORLD_SIZE=8 REPS=10 taskset -c $((2*i)),$((2*i+1)) python scripts/batch_jitter.py auto > $O/share_$i.log 2>&1 & done; wait
grep -h "median" $O/share_*.log | sed 's/.*| //'
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.pop('roofline'); c=d.pop('cpu_baseline'); print(json.dumps(d)); print({k:r[k] for k in r if k!='brute_force_kernel' and k!='note'}); print(r.get('brute_force_kernel',{}).get('frac')); print(c['value'])"
