"""Dev tool: brute-force sweep time, the matrix-core kernels (brute_variant 0: bound on the bf16 path, 2: bound in f32 MFMAs;
unseeded first sweep / seeded later sweeps) vs the plain-VALU kernel (1).  BRUTE_VARIANTS=0,2 restricts the list."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icpslam_amd import Context, synth, NN_BRUTE
for size in (sys.argv[1:] or ["200000x200000"]):
    ns, nt = (int(x) for x in size.split("x"))
    src, tgt, _ = synth.make_scan_vs_submap(ns, nt, seed=3) if nt > 300000 else synth.make_pair(ns, nt, seed=4)
    with Context(0) as ctx:
        ctx.profile_sampling(1)
        res = {}
        labels = {0: "matrix cores (bf16 bound)", 2: "matrix cores (f32 bound)", 1: "plain VALU"}
        wanted = [int(x) for x in os.environ.get("BRUTE_VARIANTS", "0,2,1").split(",")]
        for v, label in ((v, labels[v]) for v in wanted):
            t = {}
            for iters in (1, 4):
                tot = 0.0
                for rep in range(3):
                    ctx.set_source(src); ctx.set_target(tgt)      # new versions: the first sweep has no seed
                    ctx.set_params(ctx.default_params(), nn_mode=NN_BRUTE, brute_variant=v, max_iterations=iters, force_iterations=1)
                    if rep == 0:
                        ctx.align()
                        ctx.set_source(src); ctx.set_target(tgt)
                    ctx.profile_reset()
                    r = ctx.align()
                    p = ctx.profile()
                    tot += p.nn_ms
                t[iters] = tot / 3
                res[v] = r
            first, later = t[1], (t[4] - t[1]) / 3
            tf = lambda ms: 8.0 * ns * nt / (ms * 1e-3) / 1e12
            print(f"{size} {label}: first sweep {first:.3f} ms = {tf(first):.1f} TFLOP/s ({tf(first)/1.573:.1f} %), later sweeps "
                  f"{later:.3f} ms = {tf(later):.1f} TFLOP/s ({tf(later)/1.573:.1f} % of the f32 peak)", flush=True)
        first = res[wanted[0]]
        print("  same result:", all(r["n_corr"] == first["n_corr"] and (r["T"] == first["T"]).all() for r in res.values()))
