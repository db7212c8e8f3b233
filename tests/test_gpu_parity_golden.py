"""GPU parity against the committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_gpu_matches_golden(ctx, path):
    g = np.load(path)
    ctx.set_params(ctx.default_params(), max_iterations=int(g["max_iterations"]),
                   max_correspondence_distance=float(g["max_corr"]))
    ctx.set_source(g["src"])
    ctx.set_target(g["tgt"])
    idx, d2 = ctx.nn(np.eye(4))
    assert np.array_equal(idx, g["nn_idx"])                                   # index work: bit exact
    assert np.array_equal(d2.view(np.uint32), g["nn_d2"].view(np.uint32))
    got = ctx.align(guess=g["guess"] if bool(g["has_guess"]) else None, want_fitness=True)
    assert got["converged"] == bool(g["converged"])
    assert got["iterations"] == int(g["iterations"])
    assert got["state"] == int(g["state"])
    assert got["n_corr"] == int(g["n_corr"])
    T, Tg = got["T"].astype(np.float64), g["T"].astype(np.float64)
    assert np.abs(T[:3, :3] - Tg[:3, :3]).max() <= 1e-4                       # BASELINE.json: 1e-4 (R)
    assert np.linalg.norm(T[:3, 3] - Tg[:3, 3]) <= 1e-3                       # BASELINE.json: 1e-3 m (t)
    assert abs(got["fitness"] - float(g["fitness"])) <= 1e-6 * max(1.0, float(g["fitness"]))
