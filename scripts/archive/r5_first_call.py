"""Dev tool (round 5): the FIRST icpgpu_align_batch call of a process (worker contexts created, every buffer sized) against the
steady state, config 4's shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icpslam_amd import Context, synth
n_pairs, n = 64, 50000
base = [synth.make_pair(n, n, seed=1000 + k)[:2] for k in range(16)]
srcs = [base[k % 16][0] for k in range(n_pairs)]; tgts = [base[k % 16][1] for k in range(n_pairs)]
t0 = time.perf_counter()
ctx = Context(0)
t1 = time.perf_counter()
ctx.set_params(ctx.default_params(), max_iterations=10)
ms = []
for _ in range(12):
    t = time.perf_counter(); ctx.align_batch(srcs, tgts, want_fitness=True); ms.append(1e3 * (time.perf_counter() - t))
print(f"first context {1e3 * (t1 - t0):.1f} ms; batches in order [ms]: " + " ".join(f"{x:.1f}" for x in ms))
