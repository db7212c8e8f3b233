"""Dev tool: time the NN kernel variants on the GPU box and check they agree bit for bit (not part of the product)."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, synth

sizes = [(5000, 5000), (50000, 50000), (200000, 200000), (200000, 1000000)]
args = [a for a in sys.argv[1:] if "x" in a]
if args:
    sizes = [tuple(int(x) for x in a.split("x")) for a in args]
data = {}
for ns, nt in sizes:
    src, tgt, _ = synth.make_pair(ns, min(nt, 300000), seed=4)
    if nt > tgt.shape[0]:
        reps = -(-nt // tgt.shape[0])
        tgt = np.concatenate([tgt + np.float32(0.01 * k) for k in range(reps)])[:nt]
        tgt[:, 3] = 1
    data[(ns, nt)] = (src, tgt)
ref = {}
for variant in os.environ.get("VARIANTS", "0,1,2,10,12").split(","):
    os.environ["ICPGPU_NN_VARIANT"] = variant
    with Context(0) as ctx:
        for (ns, nt), (src, tgt) in data.items():
            ctx.set_source(src); ctx.set_target(tgt)
            idx, d2 = ctx.nn(np.eye(4))
            h = hashlib.sha1(idx.tobytes() + d2.tobytes()).hexdigest()[:10]
            same = ref.setdefault((ns, nt), h) == h
            ctx.profile_reset()
            n = 5
            for _ in range(n):
                ctx.nn(np.eye(4))
            p = ctx.profile()
            ms = p.nn_ms / max(1, p.nn_timed)
            pairs = ns * nt
            print(f"variant {variant:>2} {ns}x{nt}: nn {ms:8.3f} ms  {pairs/ms/1e9:6.2f} Gpair/ms  "
                  f"{8*pairs/ms/1e9:6.1f} TFLOP/s  bitexact_vs_first={same}", flush=True)
