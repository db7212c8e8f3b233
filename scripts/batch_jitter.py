"""Dev tool: run-to-run spread of icpgpu_align_batch (64 pairs of 50k), per worker count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth
base = [synth.make_pair(50000, 50000, seed=1000 + k)[:2] for k in range(8)]
pairs = [base[k % 8] for k in range(64)]
S, Tg = [p[0] for p in pairs], [p[1] for p in pairs]
for workers in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,8").split(",")]:
    os.environ["ICPGPU_BATCH_WORKERS"] = str(workers)
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), max_iterations=10)
        ctx.align_batch(S[:16], Tg[:16])
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); ctx.align_batch(S, Tg); ts.append((time.perf_counter() - t0) * 1e3)
        print(f"workers={workers}: " + " ".join(f"{t:.1f}" for t in ts) + f"  | median {np.median(ts):.1f} ms = {64e3/np.median(ts):.0f} pairs/s", flush=True)
