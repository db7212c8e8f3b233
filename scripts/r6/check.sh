#!/bin/bash
# round 6: tests around the voxel filter + the pipeline's stage timers (both inner modes) + the shim's stages
TAG=${1:-r6check}
O=gpurun_out/$TAG; mkdir -p $O /tmp/shim
( time timeout 900 python -m pytest tests/test_gpu_voxel.py tests/test_cpp_shim.py tests/test_gpu_recognition.py tests/test_gpu_sequence.py tests/test_gpu_views.py tests/test_gpu_errors.py tests/test_gpu_widened_fullsize.py -x -q -m gpu ) > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log
for i in 1 2; do timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v amdgpu.ids; done > $O/pipeline.txt 2>&1; cat $O/pipeline.txt
ICPGPU_GICP_INNER=quadratic timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v amdgpu.ids > $O/pipeline_quadratic.txt; cat $O/pipeline_quadratic.txt
ICPGPU_FLAVOUR=dev ICPGPU_VOXEL_PLANNED=0 timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep -v amdgpu.ids > $O/pipeline_unplanned.txt; cat $O/pipeline_unplanned.txt
g++ -std=c++14 -O2 -DICPGPU_SHIM_TIMING -I include tests/cpp/odometer_pipeline_demo.cpp -o /tmp/shim/demo -L icpslam_amd -licpgpu -Wl,-rpath,$PWD/icpslam_amd -Wl,-rpath,/opt/rocm/lib -pthread || exit 1
python - <<'PY'
import sys; sys.path.insert(0, '.')
from icpslam_amd import synth
a, b, _ = synth.make_pair(200000, 200000, seed=4)
a.tofile('/tmp/shim/a.bin'); b.tofile('/tmp/shim/b.bin')
PY
for th in 4 1; do for i in 1 2; do ICPGPU_DEMO_TIMING=1 timeout 60 /tmp/shim/demo /tmp/shim/a.bin 200000 /tmp/shim/b.bin 200000 104 0.2 10 $th 4 2>&1 | grep "TIMING\|STAGES\|SHIM"; done; done > $O/shim_stages.txt 2>&1
cat $O/shim_stages.txt
timeout 60 python scripts/r5_pipeline_on_bench_pair.py 2>&1 | tail -1
