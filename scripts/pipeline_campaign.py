"""Dev tool: the reference odometer's per-scan pipeline (VoxelGrid 0.2 m -> GICP, 10 iterations -> fitness) on random raw
scan pairs, GPU against the CPU oracle, stage by stage.  Usage: python scripts/pipeline_campaign.py FIRST LAST"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
import oracle
from icpslam_amd import Context, GICP, synth
first, last = int(sys.argv[1]), int(sys.argv[2])
oracle.build()
bad = 0
t0 = time.time()
with Context(0) as ctx:
    for seed in range(first, last):
        rng = np.random.default_rng(70_000 + seed)
        n = int(rng.integers(40_000, 90_000))
        src_raw, tgt_raw, _ = synth.make_pair(n, n, seed=seed)
        fs, ft = ctx.voxel_grid(src_raw, 0.2), ctx.voxel_grid(tgt_raw, 0.2)
        os_, ot = oracle.voxel_grid(src_raw, 0.2), oracle.voxel_grid(tgt_raw, 0.2)
        vox_ok = fs.shape == os_.shape and ft.shape == ot.shape and np.array_equal(fs.view(np.uint32), os_.view(np.uint32)) and \
            np.array_equal(ft.view(np.uint32), ot.view(np.uint32))
        ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)
        ctx.set_source(fs); ctx.set_target(ft)
        got = ctx.align(want_fitness=True)
        ref = oracle.icp_align(os_, ot, oracle.default_params(method=oracle.GICP, max_iterations=10), want_fitness=True)
        same = (np.array_equal(got["T"].view(np.uint32), np.asarray(ref["T"], np.float32).view(np.uint32)) and got["iterations"] == ref["iterations"]
                and got["n_corr"] == ref["n_corr"] and got["converged"] == ref["converged"])
        if not (vox_ok and same):
            bad += 1
            print(f"MISMATCH seed {seed}: voxel {vox_ok}, filtered {fs.shape[0]}/{os_.shape[0]}, iters {got['iterations']}/{ref['iterations']}, "
                  f"|dT| {np.abs(got['T'].astype(np.float64) - ref['T']).max():.2e}", flush=True)
print(f"pipeline {first}..{last}: {bad} pairs differ from the oracle in some bit, {time.time()-t0:.0f} s")
