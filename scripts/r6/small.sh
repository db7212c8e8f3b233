#!/bin/bash
# round 6: small measurements -- e2e loop's stages, the HBM-bound kernels, the callback harness with its four threads on ONE cpu
TAG=${1:-r6small}
O=gpurun_out/$TAG; mkdir -p $O /tmp/shim
timeout 200 python scripts/r6/e2e_breakdown.py 2>&1 | grep -v amdgpu.ids > $O/e2e.txt; cat $O/e2e.txt
timeout 600 python scripts/hbm_kernels.py 2>&1 | grep -v amdgpu.ids > $O/hbm_kernels.txt; cat $O/hbm_kernels.txt
g++ -std=c++14 -O2 -DICPGPU_SHIM_TIMING -I include tests/cpp/odometer_pipeline_demo.cpp -o /tmp/shim/demo -L icpslam_amd -licpgpu -Wl,-rpath,$PWD/icpslam_amd -Wl,-rpath,/opt/rocm/lib -pthread
python - <<'PY'
import sys; sys.path.insert(0, '.')
from icpslam_amd import synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
a.tofile('/tmp/shim/a.bin'); b.tofile('/tmp/shim/b.bin')
PY
{
nproc; taskset -p $$
for rep in 1 2 3; do
echo "== four rotating callback threads, free"; ICPGPU_DEMO_TIMING=1 timeout 60 /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 104 0.2 10 4 4 2>&1 | grep "TIMING\|STAGES"
CPU=$(taskset -cp $$ | sed 's/.*: //; s/[,-].*//')
echo "== four rotating callback threads, all on cpu $CPU (taskset)"; ICPGPU_DEMO_TIMING=1 timeout 60 taskset -c $CPU /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 104 0.2 10 4 4 2>&1 | grep "TIMING\|STAGES"
echo "== one callback thread, free"; ICPGPU_DEMO_TIMING=1 timeout 60 /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 104 0.2 10 1 4 2>&1 | grep "TIMING\|STAGES"
done
} > $O/shim_threads.txt 2>&1
cat $O/shim_threads.txt
