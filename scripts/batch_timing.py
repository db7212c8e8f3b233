"""Dev tool: BASELINE config 4 shape -- independent 50k x 50k pairs through icpgpu_align_batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
base = [synth.make_pair(n, n, seed=1000 + k)[:2] for k in range(8)]
pairs = [base[k % 8] for k in range(n_pairs)]
wlist = [int(x) for x in sys.argv[3].split(',')] if len(sys.argv) > 3 else (1, 2, 4, 8)
for workers in wlist:
    os.environ["ICPGPU_BATCH_WORKERS"] = str(workers)
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), max_iterations=10)
        ctx.align_batch([p[0] for p in pairs[:8]], [p[1] for p in pairs[:8]])
        t0 = time.perf_counter()
        res = ctx.align_batch([p[0] for p in pairs], [p[1] for p in pairs])
        dt = time.perf_counter() - t0
        its = sum(r["iterations"] for r in res)
        print(f"workers={workers}: {n_pairs} pairs of {n}x{n} in {dt*1e3:.1f} ms -> {n_pairs/dt:.0f} pairs/s, {its/dt:.0f} it/s (host->device copies and grid builds included)", flush=True)
