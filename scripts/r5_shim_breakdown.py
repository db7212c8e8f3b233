"""Dev tool (round 5): the calls the C++ shim makes per scan of the reference's pipeline (voxel filter with the result fetched to the
host, set_target of the previous filtered scan, set_source of the new one, align with the aligned cloud returned and the fitness),
timed one by one, against the resident loop (set_source_voxel_filtered + align + promote) on the same two scans."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, synth, _lib
a, b, _ = synth.make_pair(200000, 200000, seed=4)
raw = [a, b]
N = int(os.environ.get("SCANS", "60"))
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
    L = _lib.load()
    prev = ctx.voxel_grid(raw[1], 0.2)
    t = dict(filter=0.0, set_target=0.0, set_source=0.0, align=0.0, align_nocloud=0.0, copy=0.0)
    for k in range(N + 4):
        timed = k >= 4
        t0 = time.perf_counter()
        cur = ctx.voxel_grid(raw[k % 2], 0.2)
        t1 = time.perf_counter()
        ctx.set_target(prev)
        t2 = time.perf_counter()
        ctx.set_source(cur)
        t3 = time.perf_counter()
        r = ctx.align(want_fitness=True, want_cloud=(k % 2 == 0))
        t4 = time.perf_counter()
        prev = cur.copy()
        t5 = time.perf_counter()
        if timed:
            t["filter"] += t1 - t0; t["set_target"] += t2 - t1; t["set_source"] += t3 - t2
            t["align" if k % 2 == 0 else "align_nocloud"] += t4 - t3; t["copy"] += t5 - t4
    p = ctx.profile()
    print(f"shim-like sequence, us per scan: voxel_grid (H2D raw + filter + D2H result) {t['filter'] / N * 1e6:.0f} | set_target (recognised: {p.targets_recognised}) "
          f"{t['set_target'] / N * 1e6:.0f} | set_source (adopted: {p.sources_adopted}) {t['set_source'] / N * 1e6:.0f} | align + fitness + aligned cloud "
          f"{t['align'] / (N / 2) * 1e6:.0f}, without the cloud {t['align_nocloud'] / (N / 2) * 1e6:.0f} | host copy {t['copy'] / N * 1e6:.0f}")
    for name, env in (("adoption off", "0"),):
        pass
    # the resident loop
    ctx.set_source_voxel_filtered(raw[1], 0.2); ctx.promote_source_to_target()
    tt = dict(filter=0.0, align=0.0)
    for k in range(N + 4):
        t0 = time.perf_counter(); ctx.set_source_voxel_filtered(raw[k % 2], 0.2); t1 = time.perf_counter()
        ctx.align(want_fitness=True); t2 = time.perf_counter(); ctx.promote_source_to_target()
        if k >= 4:
            tt["filter"] += t1 - t0; tt["align"] += t2 - t1
    print(f"resident loop, us per scan: set_source_voxel_filtered {tt['filter'] / N * 1e6:.0f} | align + fitness {tt['align'] / N * 1e6:.0f}")
    fp = np.ascontiguousarray(prev)
    t0 = time.perf_counter()
    for _ in range(200): L.icpgpu_fingerprint(fp.ctypes.data_as(__import__('ctypes').POINTER(__import__('ctypes').c_float)), fp.shape[0])
    print(f"icpgpu_fingerprint of {fp.shape[0]} points: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
