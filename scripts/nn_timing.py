"""Dev tool: time the NN kernel variants on the GPU box (not part of the product or the tests)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth

sizes = [(5000, 5000), (50000, 50000), (200000, 200000), (200000, 1000000)]
if len(sys.argv) > 1:
    sizes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for variant in os.environ.get("VARIANTS", "0,1").split(","):
    os.environ["ICPGPU_NN_VARIANT"] = variant
    with Context(0) as ctx:
        for ns, nt in sizes:
            rng = np.random.default_rng(0)
            src, tgt, _ = synth.make_pair(ns, min(nt, 300000), seed=4)
            if nt > tgt.shape[0]:
                reps = -(-nt // tgt.shape[0])
                tgt = np.concatenate([tgt + np.float32(0.01 * k) for k in range(reps)])[:nt]
                tgt[:, 3] = 1
            ctx.set_source(src); ctx.set_target(tgt)
            ctx.nn(np.eye(4))
            ctx.profile_reset()
            n = 5
            t0 = time.time()
            for _ in range(n):
                ctx.nn(np.eye(4))
            wall = (time.time() - t0) / n
            p = ctx.profile()
            ms = p.nn_ms / p.nn_launches
            pairs = ns * nt
            print(f"variant {variant} {ns}x{nt}: nn {ms:.3f} ms/launch  {pairs/ms/1e9:.2f} Gpairs/ms... "
                  f"{8*pairs/ms/1e9:.1f} TFLOP/s-equiv (8 flop/pair)  wall {wall*1e3:.2f} ms", flush=True)
