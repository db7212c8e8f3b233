#!/bin/bash
# round 6: the staged result clouds (views) -- their tests, the shim's stage timers, the bench's pipeline figures
TAG=${1:-r6shim}
O=gpurun_out/$TAG; mkdir -p $O /tmp/shim
( time timeout 1500 python -m pytest tests/test_gpu_views.py tests/test_gpu_voxel.py tests/test_cpp_shim.py tests/test_gpu_recognition.py tests/test_abi.py tests/test_gpu_errors.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log
SCANS=100 python scripts/r5_shim_breakdown.py > $O/shim_breakdown.txt 2>&1; cat $O/shim_breakdown.txt
ICPGPU_FLAVOUR=dev ICPGPU_STAGE_DIRECT=0 SCANS=100 python scripts/r5_shim_breakdown.py > $O/shim_breakdown_old.txt 2>&1; cat $O/shim_breakdown_old.txt
g++ -std=c++14 -O2 -I include tests/cpp/odometer_pipeline_demo.cpp -o /tmp/shim/demo -L icpslam_amd -licpgpu -Wl,-rpath,$PWD/icpslam_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
python - <<'PY'
import sys; sys.path.insert(0, '.')
from icpslam_amd import synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
a.tofile('/tmp/shim/a.bin'); b.tofile('/tmp/shim/b.bin')
PY
for th in 4 1; do for i in 1 2 3; do ICPGPU_DEMO_TIMING=1 /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 104 0.2 10 $th 4 2>&1 | grep "TIMING\|STAGES"; done; done > $O/shim_stages.txt 2>&1
cat $O/shim_stages.txt
python bench.py --steps 20 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
g=d['gicp']; print({k:v for k,v in g.items() if not isinstance(v,(dict,list,str))}); print(g.get('shim_pipeline'))
PY
