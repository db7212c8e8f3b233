"""Generates tests/golden/*.npz.  Run here (needs scipy): python tests/golden/make_golden.py

The reference ships no tests or vectors for this path, so the fixtures are produced by the two
independent restatements of PCL's point-to-point ICP kept under oracle/:
  * expected transforms / iteration traces come from the NumPy/SciPy restatement (oracle/icp_oracle_np.py);
  * expected nearest-neighbour indices / squared distances at identity come from the C oracle's brute-force search
    (the bit-level arithmetic contract of DESIGN.md section 3).
tests/test_oracle.py checks the C oracle against these files; tests/test_gpu_parity_golden.py checks the HIP path.
A fixture is data only: inputs + expected outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from icpslam_amd import synth  # noqa: E402
from oracle import icp_oracle_np as onp  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def case(name, src, tgt, max_iterations, guess=None, max_corr=1.0):
    r = onp.icp_align(src, tgt, max_iterations=max_iterations, max_correspondence_distance=max_corr, guess=guess,
                      want_fitness=True)
    idx, d2 = oracle.nn(src, tgt, np.eye(4), nn_mode=oracle.NN_BRUTE)
    tr = r["trace"]
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), src=src, tgt=tgt, max_iterations=max_iterations, max_corr=max_corr,
        guess=np.eye(4) if guess is None else guess, has_guess=guess is not None,
        T=r["T"], converged=r["converged"], iterations=r["iterations"], state=r["state"], n_corr=r["n_corr"],
        mse=r["mse"], fitness=r["fitness"],
        trace_Tk=np.array([t["Tk"] for t in tr]).reshape(-1, 4, 4), trace_final=np.array([t["final"] for t in tr]).reshape(-1, 4, 4),
        trace_n_corr=np.array([t["n_corr"] for t in tr], np.int64), trace_mse=np.array([t["mse"] for t in tr]),
        nn_idx=idx, nn_d2=d2)
    print(name, "iters", r["iterations"], "state", r["state"], "n_corr", r["n_corr"], "fitness", r["fitness"])


def main():
    s, t, _ = synth.make_pair(2000, 2000, seed=101)
    case("pair_2k", s, t, 30)
    s, t, _ = synth.make_known_answer_pair(1500, seed=102)
    case("known_answer_1k5", s, t, 40)
    s, t, Tgt = synth.make_pair(1200, 3000, seed=103)
    g = Tgt.copy()
    g[:3, 3] += (0.08, -0.05, 0.02)
    case("ragged_guess", s, t, 10, guess=g)
    s, t, _ = synth.make_pair(1500, 1500, seed=104)
    s = s.copy()
    s[:, 0] += 0.9
    case("low_overlap", s, t, 10, max_corr=0.5)
    s, t, _ = synth.make_pair(800, 800, seed=105)
    s = s.copy()
    s[:, 1] += 300.0
    case("no_correspondences", s, t, 10)


if __name__ == "__main__":
    main()
