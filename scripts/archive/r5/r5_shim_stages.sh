#!/bin/bash
# round 5: the callback-as-integrated harness (tests/cpp/odometer_pipeline_demo.cpp) with its stage timers, release library
mkdir -p gpurun_out/r5s /tmp/shim
g++ -std=c++14 -O2 -I include tests/cpp/odometer_pipeline_demo.cpp -o /tmp/shim/demo -L icpslam_amd -l:libicpgpu.so -Wl,-rpath,$PWD/icpslam_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
python - <<'PY'
import sys; sys.path.insert(0, '.')
from icpslam_amd import synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
a.tofile('/tmp/shim/a.bin'); b.tofile('/tmp/shim/b.bin')
PY
for th in 1 4; do for i in 1 2; do ICPGPU_DEMO_TIMING=1 /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 54 0.2 10 $th 4 2>&1 | grep "TIMING\|STAGES"; done; done > gpurun_out/r5s/shim_stages.txt 2>&1
cat gpurun_out/r5s/shim_stages.txt
python scripts/pipeline_breakdown.py > gpurun_out/r5s/pipeline_breakdown.txt 2>&1; tail -12 gpurun_out/r5s/pipeline_breakdown.txt
