"""Dev tool: why a fuzz seed's transforms differ between paths (conditioning of the cross-covariance)."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import oracle
from icpslam_amd import Context, NN_GRID, NN_BRUTE, synth
import test_gpu_grid as G
with Context(0) as ctx:
    for seed in [int(a) for a in sys.argv[1:]]:
        rng = np.random.default_rng(1000 + seed)
        kinds = ["uniform", "planes", "clusters", "line", "lattice"]
        ks, kt = kinds[seed % 5], kinds[(seed // 2) % 5]
        n_s, n_t = int(rng.integers(4500, 30000)), int(rng.integers(4500, 60000))
        src, tgt = G._fuzz_cloud(rng, n_s, ks), G._fuzz_cloud(rng, n_t, kt)
        if seed % 3 == 0:
            tgt[rng.integers(0, n_t, 20), :3] = np.nan
            src[rng.integers(0, n_s, 20), :3] = np.inf
        T = synth.pose_matrix(*rng.uniform(-1, 1, 3), *rng.uniform(-0.2, 0.2, 3))
        gate = float(rng.choice([0.07, 0.4, 1.0, 3.0, 12.0]))
        res = {}
        for mode in (NN_GRID, NN_BRUTE):
            ctx.set_params(ctx.default_params(), nn_mode=mode, max_correspondence_distance=gate, max_iterations=1, force_iterations=1)
            ctx.set_source(src); ctx.set_target(tgt)
            res[mode] = ctx.align(guess=T)
        tr = oracle.icp_align(src, tgt, oracle.default_params(max_correspondence_distance=gate, max_iterations=1, force_iterations=1), guess=T, want_trace=True)["trace"][0]
        sm = tr["sums"]
        S = sm[7:16].reshape(3, 3) / sm[0] - np.outer(sm[4:7] / sm[0], sm[1:4] / sm[0])
        sv = np.linalg.svd(S, compute_uv=False)
        print(f"seed {seed}: {ks}->{kt} gate {gate} n_corr {res[NN_GRID]['n_corr']} sv {sv} ratio {sv[1]/sv[0]:.2e} |coords| {np.abs(sm[1:7]/sm[0]).max():.1f}; "
              f"|T_grid-T_brute| {np.abs(res[NN_GRID]['T']-res[NN_BRUTE]['T']).max():.2e}, |T_grid-T_oracle| {np.abs(res[NN_GRID]['T']-tr['final']).max():.2e}, "
              f"|T_brute-T_oracle| {np.abs(res[NN_BRUTE]['T']-tr['final']).max():.2e}")
