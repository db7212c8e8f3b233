"""The C++ shim (include/icpgpu_registration.hpp) driven exactly like the reference's call site."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from icpslam_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_demo(tmp_path):
    exe = tmp_path / "shim_demo"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_demo.cpp"), "-o", str(exe), "-L", libdir, "-licpgpu",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _run(exe, tmp_path, src, tgt, iters):
    a, b = tmp_path / "src.bin", tmp_path / "tgt.bin"
    src.tofile(a)
    tgt.tofile(b)
    return subprocess.run([str(exe), str(a), str(src.shape[0]), str(b), str(tgt.shape[0]), str(iters)],
                          capture_output=True, text=True)


def test_shim_compiles_and_fails_loudly_without_gpu(built, tmp_path):
    import torch
    exe = _build_demo(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    src, tgt, _ = synth.make_pair(100, 100, seed=1)
    r = _run(exe, tmp_path, src, tgt, 10)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_shim_matches_oracle(built, tmp_path):
    exe = _build_demo(tmp_path)
    src, tgt, _ = synth.make_pair(5000, 5000, seed=1)
    r = _run(exe, tmp_path, src, tgt, 10)
    assert r.returncode == 0, r.stderr
    vals = r.stdout.split()
    ok, iters, fit = int(vals[0]), int(vals[1]), float(vals[2])
    T = np.array([float(v) for v in vals[3:19]]).reshape(4, 4).T
    ref = oracle.icp_align(src, tgt, oracle.default_params(max_iterations=10), want_fitness=True, want_cloud=True)
    assert ok == 1 and iters == ref["iterations"]
    assert np.abs(T[:3, :3] - ref["T"][:3, :3]).max() <= 1e-4 and np.linalg.norm(T[:3, 3] - ref["T"][:3, 3]) <= 1e-3
    assert abs(fit - ref["fitness"]) <= 1e-9 * max(1.0, ref["fitness"])
    c = ref["cloud"].astype(np.float64)
    assert abs(float(vals[19]) - (c[:, 0] + 2 * c[:, 1] + 3 * c[:, 2] + c[:, 3]).sum()) <= 1e-2
