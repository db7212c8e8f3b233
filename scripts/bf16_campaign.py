"""Dev tool: the bf16-bound brute-force kernel (brute_variant 0) against the plain vector kernel (1) on random clouds of
8 192 - 60 000 points -- shapes, scales, offsets, duplicates, outliers, non-finite points -- with the kernel's test mode on:
every pair evaluated exactly against its bound.  Usage: python scripts/bf16_campaign.py FIRST LAST"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ICPGPU_MFMA_CHECK_BOUND"] = "1"
import numpy as np
from icpslam_amd import Context, NN_BRUTE, synth
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = viol = 0
worst = 0.0
t0 = time.time()
with Context(0) as ctx:
    for seed in range(first, last):
        rng = np.random.default_rng(90_000 + seed)
        ns, nt = int(rng.integers(8192, 60000)), int(rng.integers(8192, 60000))
        kind = int(rng.integers(0, 5))
        scale = float(10.0 ** rng.uniform(-3, 3))
        off = rng.uniform(-1, 1, 3) * float(10.0 ** rng.uniform(0, 4)) * (rng.random() < 0.5)
        if kind == 0:
            s, t, _ = synth.make_pair(ns, nt, seed=seed)
            s, t = s[:, :3], t[:, :3]
        elif kind == 1:
            s, t = rng.uniform(-30, 30, (ns, 3)), rng.uniform(-30, 30, (nt, 3))
        elif kind == 2:
            c = rng.uniform(-20, 20, (10, 3))
            s = c[rng.integers(0, 10, ns)] + rng.normal(0, 0.2, (ns, 3)); t = c[rng.integers(0, 10, nt)] + rng.normal(0, 0.2, (nt, 3))
        elif kind == 3:
            s, t = rng.uniform(-30, 30, (ns, 3)), rng.uniform(-30, 30, (nt, 3))
            s[:, 2] = rng.normal(0, 0.01, ns); t[:, 2] = rng.normal(0, 0.01, nt)           # sheets
        else:
            g = rng.integers(0, 40, (nt, 3)).astype(np.float64) * 0.5                       # a lattice: ties everywhere
            t = g; s = rng.integers(0, 40, (ns, 3)).astype(np.float64) * 0.5 + 0.25
        src = np.ones((ns, 4), np.float32); tgt = np.ones((nt, 4), np.float32)
        src[:, :3] = (s * scale + off).astype(np.float32); tgt[:, :3] = (t * scale + off).astype(np.float32)
        k = int(rng.integers(0, 40))
        if k:
            tgt[rng.integers(0, nt, k)] = tgt[rng.integers(0, nt)]                          # duplicates
            src[rng.integers(0, ns, k), :3] += rng.normal(0, 300.0 * scale, (k, 3)).astype(np.float32)   # outliers
        if rng.random() < 0.3:
            src[rng.integers(0, ns), :3] = np.nan; tgt[rng.integers(0, nt), :3] = np.inf; tgt[rng.integers(0, nt), :3] = np.nan
        T = synth.pose_matrix(*(rng.uniform(-0.5, 0.5, 3) * scale), *rng.uniform(-0.1, 0.1, 3)).astype(np.float32)
        got = {}
        for v in (0, 1):
            ctx.set_params(ctx.default_params(), nn_mode=NN_BRUTE, brute_variant=v)
            ctx.set_source(src); ctx.set_target(tgt)
            ctx.profile_reset()
            got[v] = ctx.nn(T)
            if v == 0:
                p = ctx.profile()
                viol += int(p.brute_bound_violations); worst = max(worst, p.brute_bound_worst)
        same = np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1].view(np.uint32), got[1][1].view(np.uint32))
        if not same:
            bad += 1
            print(f"MISMATCH seed {seed} kind {kind} scale {scale:.3g} ns {ns} nt {nt}", flush=True)
print(f"bf16 campaign {first}..{last}: {bad} clouds with different keys, {viol} bound violations, worst excess {worst:.3e} of (P^2 + |v|^2) (tau = {2.0**-13:.3e}), {time.time()-t0:.0f} s")
