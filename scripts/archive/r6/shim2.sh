#!/bin/bash
# round 6: view tests + where the shim's align spends its time (shim phase timers, the library's GICP stage timers)
TAG=${1:-r6shim2}
O=gpurun_out/$TAG; mkdir -p $O /tmp/shim
( time timeout 1500 python -m pytest tests/test_gpu_views.py tests/test_gpu_voxel.py tests/test_cpp_shim.py tests/test_gpu_recognition.py tests/test_abi.py tests/test_gpu_errors.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log
g++ -std=c++14 -O2 -DICPGPU_SHIM_TIMING -I include tests/cpp/odometer_pipeline_demo.cpp -o /tmp/shim/demo -L icpslam_amd -licpgpu -Wl,-rpath,$PWD/icpslam_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
g++ -std=c++14 -O2 -DICPGPU_SHIM_TIMING -I include tests/cpp/odometer_pipeline_demo.cpp -o /tmp/shim/demo_dev -L icpslam_amd -l:libicpgpu_dev.so -Wl,-rpath,$PWD/icpslam_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
python - <<'PY'
import sys; sys.path.insert(0, '.')
from icpslam_amd import synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
a.tofile('/tmp/shim/a.bin'); b.tofile('/tmp/shim/b.bin')
PY
for th in 4 1; do for i in 1 2; do ICPGPU_DEMO_TIMING=1 /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 104 0.2 10 $th 4 2>&1 | grep "TIMING\|STAGES\|SHIM"; done; done > $O/shim_stages.txt 2>&1
cat $O/shim_stages.txt
ICPGPU_GICP_TIMING=1 ICPGPU_DEMO_TIMING=1 /tmp/shim/demo_dev /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 104 0.2 10 1 4 > $O/shim_dev.txt 2>&1; grep -v "^[0-9]" $O/shim_dev.txt | tail -20
ICPGPU_FLAVOUR=dev ICPGPU_GICP_TIMING=1 python scripts/r5_pipeline_on_bench_pair.py > $O/resident_dev.txt 2>&1; tail -20 $O/resident_dev.txt
