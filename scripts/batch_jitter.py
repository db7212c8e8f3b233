"""Dev tool: run-to-run spread of icpgpu_align_batch (64 pairs of 50k, <= 10 iterations + fitness) per (threads, depth)
setting: "TxK,TxK,..." (default: the library's own choice).  One subprocess per setting (the switches are read per call,
but worker contexts are kept across calls)."""
import gc, os, sys, time
if not os.environ.get('KEEP_GC'): gc.disable()   # a gen-2 collection of CPython (~40 ms) is not the library's jitter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth
base = [synth.make_pair(50000, 50000, seed=1000 + k)[:2] for k in range(8)]
pairs = [base[k % 8] for k in range(64)]
S, Tg = [p[0] for p in pairs], [p[1] for p in pairs]
reps = int(os.environ.get("REPS", "20"))
for setting in (sys.argv[1] if len(sys.argv) > 1 else "auto").split(","):
    if setting != "auto":
        t, k = setting.split("x")
        os.environ["ICPGPU_BATCH_THREADS"], os.environ["ICPGPU_BATCH_DEPTH"] = t, k
    else:
        os.environ.pop("ICPGPU_BATCH_THREADS", None); os.environ.pop("ICPGPU_BATCH_DEPTH", None)
    with Context(0) as ctx:
        ctx.set_params(ctx.default_params(), max_iterations=10)
        ctx.align_batch(S[:16], Tg[:16], want_fitness=True)
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); ctx.align_batch(S, Tg, want_fitness=True); ts.append((time.perf_counter() - t0) * 1e3)
        med = np.median(ts)
        print(f"threads x depth = {setting}: " + " ".join(f"{t:.1f}" for t in ts) + f"  | median {med:.1f} ms = {64e3/med:.0f} pairs/s, "
              f"max/median {max(ts)/med:.2f}", flush=True)
