"""Dev tool: achieved bandwidth of the HBM-bound kernels (a6 transform, a3+a4 reduce over keys, f2 voxel filter, f4 map
insert) against the 8 TB/s peak -- algorithmic bytes / HIP-event time of the kernel(s)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, NN_BRUTE, synth
PEAK = 8000.0
with Context(0) as ctx:
    ctx.profile_sampling(1)
    for n in (200000, 1000000, 4000000, 16000000):
        rng = np.random.default_rng(n)
        src = np.ones((n, 4), np.float32); src[:, :3] = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
        small = src[:4096].copy()
        T = synth.pose_matrix(0.1, 0.2, 0.0, 0.0, 0.0, 0.05)
        ctx.set_params(ctx.default_params(), nn_mode=NN_BRUTE)
        ctx.set_source(src); ctx.set_target(small)
        ctx.transform(T); ctx.profile_reset()
        for _ in range(5): ctx.transform(T)
        p = ctx.profile(); ms = p.transform_ms / p.transform_launches
        if 16 * n <= (8 << 20):  # round 6: clouds up to 8 MB are written by the kernel straight into pinned HOST memory (icpgpu_transform's output)
            print(f"transform_kernel  n={n:8d}: {ms*1e3:8.1f} us, {16.0*n/ms/1e6:7.1f} GB/s of 16-byte stores into pinned host memory over PCIe (not an HBM figure: the aligned cloud's way to the caller)")
        else:
            print(f"transform_kernel  n={n:8d}: {ms*1e3:8.1f} us, {32.0*n/ms/1e6:7.0f} GB/s = {32.0*n/ms/1e6/PEAK*100:5.1f} % of HBM peak (32 B/point)")
        ctx.nn(T)                                        # keys for the reduce (brute force against a tiny target: cheap)
        ctx.reduce(T, 1e9); ctx.profile_reset()
        t0 = time.perf_counter()
        for _ in range(5): ctx.reduce(T, 1e9)
        wall = (time.perf_counter() - t0) / 5
        print(f"reduce (keys path) n={n:8d}: {wall*1e3:8.3f} ms per call incl. launch + 136 B readback; {40.0*n/wall/1e9:7.0f} GB/s "
              f"({40.0*n/wall/1e9/PEAK*100:5.1f} %) if all of it were the kernel (40 B/point)")
        if n <= 1000000 and not os.environ.get('ONLY_STREAM'):
            ctx.voxel_grid(src, 0.2); ctx.profile_reset()
            for _ in range(3): out = ctx.voxel_grid(src, 0.2)
            p = ctx.profile(); ms = p.voxel_ms / p.voxel_launches
            print(f"voxel filter      n={n:8d}: {ms*1e3:8.1f} us device time -> {len(out)} points; {p.voxel_bytes/p.voxel_launches/ms/1e6:7.0f} GB/s "
                  f"algorithmic (16 B in + 16 B out per point; was 88 / 212 us through the library sort)")
            ctx.map_reset(0.5); ctx.map_add_points(src); ctx.map_reset(0.5); ctx.profile_reset()
            ctx.map_add_points(src)
            ctx.map_reset(0.5); ctx.map_add_points(src[:10]); ctx.profile_reset(); ctx.map_add_points(src)
            p = ctx.profile()
            print(f"map insert        n={n:8d}: {p.map_insert_ms*1e3:8.1f} us, {(16+16+8+12)*n/p.map_insert_ms/1e6:7.0f} GB/s algorithmic (52 B/point: read, staged copy, probe, flags)")
