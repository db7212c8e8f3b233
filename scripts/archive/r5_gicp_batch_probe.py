"""Dev tool (round 5): GICP through icpgpu_align_batch -- 64 voxel-filtered (0.2 m) pairs of 120k-point raw scans (~20k points each),
by host threads x resumable runs per thread; every result compared with a single icpgpu_align of the same pair (bit for bit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, synth
n_pairs = int(os.environ.get("PAIRS", "64"))
n_distinct = int(os.environ.get("DISTINCT", "16"))
with Context(0) as f:
    base = []
    for k in range(n_distinct):
        a, b, _ = synth.make_pair(120000, 120000, seed=2000 + k)
        base.append((f.voxel_grid(a, 0.2), f.voxel_grid(b, 0.2)))
srcs = [base[k % n_distinct][0] for k in range(n_pairs)]; tgts = [base[k % n_distinct][1] for k in range(n_pairs)]
print("points per cloud:", sorted(set(x.shape[0] for x in srcs))[:4], "...", flush=True)
singles = None
if os.environ.get("CHECK", "1") != "0":
    singles = []
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
        for k in range(n_distinct):
            ctx.set_source(base[k][0]); ctx.set_target(base[k][1])
            singles.append(ctx.align(want_fitness=True))
combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(8, 1), (1, 8), (2, 4), (4, 2), (1, 16), (2, 8)]
for threads, depth in combos:
    for name, v in (("ICPGPU_BATCH_THREADS", threads), ("ICPGPU_BATCH_DEPTH", depth)):   # 0x0: the library's own choice
        if v: os.environ[name] = str(v)
        else: os.environ.pop(name, None)
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
        res = ctx.align_batch(srcs, tgts, want_fitness=True)
        bad = 0
        if singles:
            for k, r in enumerate(res):
                s = singles[k % n_distinct]
                if not (np.array_equal(r["T"], s["T"]) and r["iterations"] == s["iterations"] and r["n_corr"] == s["n_corr"] and r["fitness"] == s["fitness"]):
                    bad += 1
        ms = []
        ctx.profile_reset()
        for _ in range(int(os.environ.get("REPS", "5"))):
            t0 = time.perf_counter()
            ctx.align_batch(srcs, tgts, want_fitness=True)
            ms.append(1e3 * (time.perf_counter() - t0))
        p = ctx.profile()
        ms.sort()
        print(f"threads={threads} runs/thread={depth}: ms per {n_pairs} pairs min {ms[0]:.1f} median {ms[len(ms) // 2]:.1f} -> {n_pairs * 1e3 / ms[len(ms) // 2]:.0f} pairs/s; "
              f"differing from single aligns: {bad}; device solves {p.gicp_device_solves}, host solves {p.gicp_host_solves}, "
              f"evaluation {p.gicp_eval_ms / max(1, p.gicp_cost_launches) * 1e3:.2f} us", flush=True)
