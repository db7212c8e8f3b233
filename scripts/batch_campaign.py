"""Dev tool (round 5): icpgpu_align_batch against single icpgpu_align calls, bit for bit, over random batches -- point-to-point
(lock-step groups on their own streams, several per host thread) and GICP (resumable runs).  usage: batch_campaign.py <first> <last>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from icpslam_amd import Context, GICP, synth
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = pairs_done = 0
with Context(0) as one, Context(0) as bat:
    for seed in range(first, last):
        rng = np.random.default_rng(seed)
        gicp = bool(seed & 1)
        n_pairs = int(rng.integers(3, 40))
        pairs = []
        for k in range(n_pairs):
            n = int(rng.integers(3000, 45000 if not gicp else 26000))
            s, t, _ = synth.make_pair(n, int(n * rng.uniform(0.8, 1.2)), seed=seed * 100 + k)
            pairs.append((s, t))
        kw = dict(max_iterations=int(rng.integers(3, 11)))
        if gicp: kw["method"] = GICP
        os.environ["ICPGPU_BATCH_THREADS"] = str(int(rng.integers(1, 5)))
        os.environ["ICPGPU_BATCH_DEPTH"] = str(int(rng.integers(1, 9)))
        os.environ["ICPGPU_BATCH_GROUPS"] = str(int(rng.integers(1, 13)))
        one.set_params(one.default_params(), **kw); bat.set_params(bat.default_params(), **kw)
        got = bat.align_batch([p[0] for p in pairs], [p[1] for p in pairs], want_fitness=True)
        for (s, t), g in zip(pairs, got):
            one.set_source(s); one.set_target(t)
            w = one.align(want_fitness=True)
            ok = (np.array_equal(g["T"], w["T"]) and g["iterations"] == w["iterations"] and g["n_corr"] == w["n_corr"] and g["state"] == w["state"]
                  and (g["fitness"] == w["fitness"] if n < 100000 else abs(g["fitness"] - w["fitness"]) <= 1e-12 * w["fitness"]))
            bad += 0 if ok else 1
            pairs_done += 1
        if (seed - first) % 10 == 9: print(f"seed {seed}: {pairs_done} pairs, {bad} differing", flush=True)
print(f"batch campaign seeds {first}..{last - 1}: {pairs_done} pairs through icpgpu_align_batch (P2P lock-step groups / GICP runs), {bad} differing from single aligns")
