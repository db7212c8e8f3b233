"""Dev tool: target points evaluated per grid sweep (mean over the ten sweeps of an alignment), for profiles/pmc_issue.json."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icpslam_amd import Context, synth, NN_GRID
ns, nt = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "200000x200000").split("x"))
src, tgt, _ = synth.make_scan_vs_submap(ns, nt, seed=3) if nt > 300000 else synth.make_pair(ns, nt, seed=4)
with Context(0) as ctx:
    ctx.set_params(ctx.default_params(), max_iterations=10, force_iterations=1, nn_mode=NN_GRID)
    ctx.set_source(src); ctx.set_target(tgt)
    ctx.align()
    ctx.count_candidates(True)
    reps = 3
    for _ in range(reps):
        ctx.align()
    n = ctx.candidates()
    ctx.count_candidates(False)
    print(f"{ns}x{nt} candidates_per_launch {n / (10.0 * reps):.1f} per_source_point {n / (10.0 * reps) / ns:.1f}")
