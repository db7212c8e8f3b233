#!/bin/bash
# round 6: the host merge of an evaluation as vectors against the library of the commit before: same bits (exact GICP), then the rates
# library of the commit before: the same bits over a 60-scan drive and 64 random pairs; then the rates
TAG=${1:-r6mergeab}
O=gpurun_out/$TAG; mkdir -p $O
cat > /tmp/quad_drive.py <<'PY'
import sys, hashlib, numpy as np
sys.path.insert(0, '.')
from icpslam_amd import Context, GICP, GICP_INNER_QUADRATIC, synth
h = hashlib.sha256()
with Context(0) as c:
    c.set_params(c.default_params(), method=GICP, max_iterations=10)
    scene = synth.make_scene(5, extent=120.0); rng = np.random.default_rng(5); P = np.eye(4)
    c.set_source_voxel_filtered(synth.scan(scene, P, 100000, seed=8000), 0.2); c.promote_source_to_target()
    for k in range(1, 40):
        P = P @ synth.pose_matrix(0.25, 0, 0, 0, 0, np.deg2rad(rng.uniform(-3, 3)))
        c.set_source_voxel_filtered(synth.scan(scene, P, 100000, seed=8000 + k), 0.2)
        r = c.align(want_fitness=True); c.promote_source_to_target()
        h.update(r['T'].tobytes()); h.update(np.float64(r['fitness']).tobytes()); h.update(np.int32(r['iterations']).tobytes())
    for seed in range(48):
        s, t, _ = synth.make_pair(9000, 9500, seed=500 + seed)
        c.set_source(s); c.set_target(t); r = c.align(want_fitness=True)
        h.update(r['T'].tobytes()); h.update(np.float64(r['fitness']).tobytes()); h.update(np.int32(r['iterations']).tobytes())
print('results sha256', h.hexdigest())
PY
{
echo "== this build"; timeout 300 python /tmp/quad_drive.py 2>&1 | grep -v amdgpu.ids
echo "== the commit before (scalar accumulators)"; ICPGPU_LIB_PATH=$PWD/icpslam_amd/csrc/build_ab/libicpgpu_prev.so timeout 300 python /tmp/quad_drive.py 2>&1 | grep -v amdgpu.ids
grep -c avx2 /proc/cpuinfo | head -1
for rep in 1 2 3; do
echo "== this build"; timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep "scans of\|host wall"
echo "== the commit before"; ICPGPU_LIB_PATH=$PWD/icpslam_amd/csrc/build_ab/libicpgpu_prev.so timeout 120 python scripts/pipeline_breakdown.py 43 2>&1 | grep "scans of\|host wall"
done
} > $O/quad_ab.txt 2>&1
cat $O/quad_ab.txt
timeout 900 python -m pytest tests/test_gpu_gicp.py tests/test_gpu_parity_golden.py -q -m gpu 2>&1 | tail -2
