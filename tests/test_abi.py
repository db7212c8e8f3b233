"""CPU tests of the drop-in boundary: libicpgpu.so builds for gfx950, loads, and exports exactly what
include/icpgpu.h declares; the host-only entry points work; the device entry points fail loudly without a GPU."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
from icpslam_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "icpgpu.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^#define(.*\\\n)*.*$", "", text, flags=re.M)     # the three macros over the sized entry points: not symbols
    return sorted(set(re.findall(r"\b(icpgpu_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_all_exported(built):
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/icpgpu.h but not exported by libicpgpu.so"
    assert sorted(_lib.EXPORTS) == names
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r"\bT (icpgpu_[a-z_0-9]+)", out)))
    assert exported == names, "library exports symbols the header does not declare (or vice versa)"


def test_library_contains_gfx950_code_object(built):
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"nn_brute_kernel" in blob


def test_header_compiles_as_c_and_struct_sizes_match(built, tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "icpgpu.h"\n#include <stdio.h>\nint main(void){printf("%zu %zu %zu\\n", sizeof(icpgpu_params), '
                   'sizeof(icpgpu_result), sizeof(icpgpu_profile)); return 0;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)], text=True).split()]
    assert sizes == [C.sizeof(_lib.Params), C.sizeof(_lib.Result), C.sizeof(_lib.Profile)]


def test_struct_sizes_and_the_abi_rule(built):
    """include/icpgpu.h, "ABI rule": the library reports its structs' sizes (equal to the ctypes mirrors'), default_params_sz fills a
    SHORTER struct (an older 1.x header) without touching what follows it and zero-fills the tail of a LONGER one (a newer header);
    a context cannot be created for a header of another major version, nor with sizes no 1.x header ever had; and the unsized
    symbols of 0.x are gone, so a binary built against them fails to load instead of overrunning its structs."""
    lib = _lib.load()
    sizes = (C.c_size_t * 3)()
    lib.icpgpu_struct_sizes(sizes)
    assert list(sizes) == [C.sizeof(_lib.Params), C.sizeof(_lib.Result), C.sizeof(_lib.Profile)]
    assert lib.icpgpu_version() == _lib.HEADER_VERSION
    full = _lib.Params()
    lib.icpgpu_default_params(C.byref(full))
    n = C.sizeof(_lib.Params)
    buf = (C.c_ubyte * (n + 64))(*([0xAB] * (n + 64)))
    lib.icpgpu_default_params_sz(C.cast(buf, C.POINTER(_lib.Params)), n - 8)                 # a caller without the last fields
    assert bytes(buf[:n - 8]) == bytes(full)[:n - 8] and set(buf[n - 8:]) == {0xAB}
    lib.icpgpu_default_params_sz(C.cast(buf, C.POINTER(_lib.Params)), n + 32)                # a caller with fields the library lacks
    assert bytes(buf[:n]) == bytes(full) and set(buf[n:n + 32]) == {0} and set(buf[n + 32:]) == {0xAB}
    h = C.c_void_p()
    sz = [C.sizeof(_lib.Params), C.sizeof(_lib.Result), C.sizeof(_lib.Profile)]
    assert lib.icpgpu_create_abi(C.byref(h), 0, 4, *sz) == _lib.ERR_UNSUPPORTED and not h.value          # a 0.4 header
    assert b"major version" in lib.icpgpu_last_error(None)
    assert lib.icpgpu_create_abi(C.byref(h), 0, 2000, *sz) == _lib.ERR_UNSUPPORTED and not h.value
    assert lib.icpgpu_create_abi(C.byref(h), 0, _lib.HEADER_VERSION, 48, sz[1], sz[2]) == _lib.ERR_INVALID_ARG and not h.value
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    for gone in ("icpgpu_create", "icpgpu_default_params", "icpgpu_align_batch_multi"):
        assert not re.search(rf"\bT {gone}$", out, flags=re.M), gone


def test_version_and_default_params(built):
    lib = _lib.load()
    assert lib.icpgpu_version() >= 1
    p = _lib.Params()
    lib.icpgpu_default_params(C.byref(p))
    # the reference's odometer constants: /root/reference/include/icpslam/icp_odometer.h:63-65
    assert (p.max_iterations, p.transformation_epsilon, p.max_correspondence_distance) == (10, 1e-6, 1.0)
    assert p.min_correspondences == 3 and p.method == _lib.P2P_SVD and p.euclidean_fitness_epsilon < -1e300


def test_host_solve_matches_oracle(built):
    """a5 is host code (north_star keeps the 3x3 SVD on the host), so it is testable without a GPU."""
    lib = _lib.load()
    rng = np.random.default_rng(3)
    dp = C.POINTER(C.c_double)
    for trial in range(40):
        n = 500
        p = rng.normal(size=(n, 3)) * (30, 20, 1.5) + (5, -3, 0.2)
        T = synth.pose_matrix(*rng.uniform(-0.5, 0.5, 3), *rng.uniform(-0.1, 0.1, 3))
        q = p @ T[:3, :3].T + T[:3, 3] + rng.normal(size=(n, 3)) * 0.02
        if trial % 5 == 0:
            p[:, 2] = q[:, 2] = 0.0                       # rank-deficient covariance
        sums = np.zeros(17)
        sums[0], sums[1:4], sums[4:7], sums[7:16] = n, p.sum(0), q.sum(0), (q.T @ p).reshape(-1)
        Tk = np.zeros(16)
        assert lib.icpgpu_solve(sums.ctypes.data_as(dp), Tk.ctypes.data_as(dp)) == 0
        got = Tk.reshape(4, 4).T
        np.testing.assert_allclose(got, oracle.umeyama(sums), atol=1e-10)
        assert abs(np.linalg.det(got[:3, :3]) - 1) < 1e-10
    bad = np.zeros(17)
    Tk = np.zeros(16)
    assert lib.icpgpu_solve(bad.ctypes.data_as(dp), Tk.ctypes.data_as(dp)) != 0
    np.testing.assert_array_equal(Tk.reshape(4, 4), np.eye(4))


def test_no_gpu_fails_loudly(built):
    """No silent CPU fallback: without a device the context cannot be created and says why."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.icpgpu_create(C.byref(h), 0)
    assert rc == _lib.ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in lib.icpgpu_last_error(None)
    from icpslam_amd import Context, IcpGpuError
    with pytest.raises(IcpGpuError):
        Context(0)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "icpslam_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "icp_oracle" not in text.replace(
                    "oracle/icp_oracle.c", ""), f
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read()


def test_fingerprint_is_a_content_hash(built):
    """icpgpu_fingerprint (host side of icpgpu_set_target's recognition): equal for equal content wherever the buffer lives,
    different for one flipped bit, for two swapped points, for a shorter cloud."""
    import ctypes as C
    lib = _lib.load()
    rng = np.random.default_rng(3)
    a = rng.normal(size=(1000, 4)).astype(np.float32)
    fp = lambda x: lib.icpgpu_fingerprint(x.ctypes.data_as(C.POINTER(C.c_float)), x.shape[0])
    assert fp(a) == fp(a.copy()) != 0
    b = a.copy()
    b.view(np.uint32)[517, 2] ^= 1
    assert fp(b) != fp(a)
    c = a.copy()
    c[[3, 4]] = c[[4, 3]]
    assert fp(c) != fp(a)
    assert fp(a[:999]) != fp(a)
    assert fp(np.zeros((0, 4), np.float32)) == fp(np.zeros((0, 4), np.float32))


def test_fingerprint_equals_its_definition_at_every_length(built):
    """The host fingerprint has a vector path (AVX-512 DQ where the CPU has it: four points per step, two steps per trip) in front
    of the scalar loop: the number must be the DEFINITION's -- sum over the points of mix(w0 + K1 (2i + 1)) + mix(w1 ^ K2 (2i + 2)),
    plus mix(n ^ 0xa5..) -- at every length around the vector loop's trip sizes (the device kernel computes the same sum:
    tests/test_gpu_recognition.py)."""
    import ctypes as C
    lib = _lib.load()
    M = (1 << 64) - 1

    def mix(x):
        x ^= x >> 30
        x = (x * 0xbf58476d1ce4e5b9) & M
        x ^= x >> 27
        x = (x * 0x94d049bb133111eb) & M
        return x ^ (x >> 31)

    def definition(a):
        s = 0
        for i, (w0, w1) in enumerate(a.view(np.uint64).reshape(-1, 2).tolist()):
            s = (s + mix((w0 + 0x9e3779b97f4a7c15 * (2 * i + 1)) & M) + mix(w1 ^ ((0xd6e8feb86659fd93 * (2 * i + 2)) & M))) & M
        return (s + mix(a.shape[0] ^ 0xa5a5a5a5a5a5a5a5)) & M

    rng = np.random.default_rng(5)
    for n in (0, 1, 3, 8, 63, 64, 65, 71, 72, 73, 127, 128, 1001, 4099):
        a = rng.normal(size=(n, 4)).astype(np.float32)
        a.view(np.uint32)[:: 7] ^= 0x80000000  # sign bits, and ...
        if n > 3:
            a[3] = [np.nan, np.inf, -0.0, 1.0]  # ... bit patterns that are not numbers
        assert lib.icpgpu_fingerprint(a.ctypes.data_as(C.POINTER(C.c_float)), n) == definition(a), n
        b = np.empty((n + 1, 4), np.float32)[1:]  # (only 16-byte aligned)
        b[...] = a
        assert lib.icpgpu_fingerprint(b.ctypes.data_as(C.POINTER(C.c_float)), n) == definition(a), n


def test_multi_gpu_entry_reports_codes_without_a_gpu(built):
    """icpgpu_align_batch_multi (one process, one host thread per device): argument errors are status codes with a message, and
    without a usable device the call fails like icpgpu_create does -- no fallback, no abort."""
    import ctypes as C
    lib = _lib.load()
    res = (_lib.Result * 1)()
    assert lib.icpgpu_align_batch_multi(None, 1, None, 0, None, None, None, None, 0, res, None, 0) == _lib.ERR_INVALID_ARG
    assert b"device list" in lib.icpgpu_multi_last_error()
    dev = (C.c_int * 1)(0)
    assert lib.icpgpu_align_batch_multi(dev, 1, None, 0, None, None, None, None, 0, res, None, 7) == _lib.ERR_INVALID_ARG
    assert b"communicator" in lib.icpgpu_multi_last_error()
    import torch
    if not torch.cuda.is_available():
        rc = lib.icpgpu_align_batch_multi(dev, 1, None, 0, None, None, None, None, 0, res, None, 0)
        assert rc == _lib.ERR_NO_DEVICE and b"no CPU fallback" in lib.icpgpu_multi_last_error()


def test_release_library_has_no_development_switches(built):
    """icpslam_amd/csrc/icp_env.h: the release library reads only the PRODUCTION environment switches; every tuning / variant /
    test-mode switch -- among them the ones that change results on purpose (ICPGPU_SKIP_UNCERT, *_NO_EXACT,
    ICPGPU_VOXEL_TEST_STALL) -- and the experimental tile search exist only in libicpgpu_dev.so."""
    here = os.path.dirname(_lib.LIB_PATH)

    def names(path):
        blob = open(path, "rb").read()
        return set(m.decode() for m in re.findall(rb"ICPGPU_[A-Z][A-Z_0-9]+", blob))

    production = {"ICPGPU_WAIT_TIMEOUT_MS", "ICPGPU_BATCH_THREADS", "ICPGPU_BATCH_DEPTH", "ICPGPU_BATCH_GROUPS", "ICPGPU_RECOGNISE", "ICPGPU_GICP_SERVER",
                  "ICPGPU_GICP_DEVICE", "ICPGPU_GICP_INNER", "ICPGPU_MAILBOX", "ICPGPU_DEBUG"}
    rel = names(os.path.join(here, "libicpgpu.so"))
    dev = names(os.path.join(here, "libicpgpu_dev.so"))
    stray = {n for n in rel if n not in production and not n.startswith(("ICPGPU_COMM_", "ICPGPU_ERR_", "ICPGPU_MAP_"))}
    assert not stray, stray
    for n in ("ICPGPU_SKIP_UNCERT", "ICPGPU_MFMA_NO_EXACT", "ICPGPU_TILE_NO_EXACT", "ICPGPU_VOXEL_TEST_STALL", "ICPGPU_TILE_SEARCH",
              "ICPGPU_MFMA_CHECK_BOUND"):
        assert n in dev and n not in rel, n
    sym = subprocess.run(["nm", "-D", "--defined-only", os.path.join(here, "libicpgpu.so")], capture_output=True, text=True).stdout
    assert "tile_search" not in sym          # icp_tile.hip is not linked into the release library
