#!/bin/bash
# Dev tool (round 4): the callback-as-integrated harness (tests/cpp/odometer_pipeline_demo.cpp) with its stage timers
mkdir -p gpurun_out/r4s /tmp/shim
g++ -std=c++14 -O2 -I include tests/cpp/odometer_pipeline_demo.cpp -o /tmp/shim/demo -L icpslam_amd -l:libicpgpu_dev.so -Wl,-rpath,$PWD/icpslam_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
python - <<'PY'
import sys; sys.path.insert(0, '.')
from icpslam_amd import synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
a.tofile('/tmp/shim/a.bin'); b.tofile('/tmp/shim/b.bin')
PY
for i in 1 2; do ICPGPU_GICP_TIMING=1 ICPGPU_DEMO_TIMING=1 /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 54 0.2 10 1 4 2>&1 | grep "TIMING\|STAGES\|GICP alignments"; done > gpurun_out/r4s/shim_stages.txt 2>&1
cat gpurun_out/r4s/shim_stages.txt
