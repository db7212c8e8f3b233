"""The C-ABI's error convention on a real device (include/icpgpu.h): status codes instead of exceptions or aborts,
non-convergence is not an error (PCL: hasConverged() == false), a context stays usable after a failed call."""
import ctypes as C

import numpy as np
import pytest

from icpslam_amd import GICP, NN_AUTO, NN_GRID, IcpGpuError, _lib, synth

pytestmark = pytest.mark.gpu

ERR_INVALID_ARG, ERR_NO_INPUT = -1, -5
NO_CORRESPONDENCES = 5          # ICPGPU_CONV_NO_CORRESPONDENCES


def _code(exc):
    return exc.value.code


def test_calls_before_inputs_report_no_input_and_context_survives():
    from icpslam_amd import Context
    with Context(0) as ctx:
        for call in (ctx.align, ctx.fitness, lambda: ctx.nn(np.eye(4)), ctx.promote_source_to_target,
                     lambda: ctx.map_add_points(np.ones((4, 4), np.float32)), lambda: ctx.map_add_source(),
                     lambda: ctx.map_nn_target(np.eye(4), np.eye(4))):
            with pytest.raises(IcpGpuError) as e:
                call()
            assert _code(e) == ERR_NO_INPUT and str(e.value)
        src, tgt, _ = synth.make_pair(3000, 3000, seed=1)
        ctx.set_source(src)
        with pytest.raises(IcpGpuError) as e:       # target still missing
            ctx.align()
        assert _code(e) == ERR_NO_INPUT
        ctx.set_target(tgt)
        assert ctx.align()["converged"]              # the same context works after all those failures


def test_invalid_arguments(ctx):
    L = _lib.load()
    with pytest.raises(IcpGpuError) as e:
        ctx.set_params(ctx.default_params(), method=7)
    assert _code(e) == ERR_INVALID_ARG
    with pytest.raises(IcpGpuError) as e:
        ctx.set_params(ctx.default_params(), nn_mode=9)
    assert _code(e) == ERR_INVALID_ARG
    with pytest.raises(IcpGpuError) as e:
        ctx.map_reset(0.0)
    assert _code(e) == ERR_INVALID_ARG
    with pytest.raises(IcpGpuError) as e:
        ctx.voxel_grid(np.ones((10, 4), np.float32), -1.0)
    assert _code(e) == ERR_INVALID_ARG
    assert L.icpgpu_align(None, None, None, 0, None) == ERR_INVALID_ARG        # null context: code, no crash
    assert L.icpgpu_set_source(ctx._h, None, C.c_size_t(5)) == ERR_INVALID_ARG  # null cloud with n > 0
    assert b"null" in L.icpgpu_last_error(ctx._h)
    with pytest.raises(IcpGpuError) as e:
        ctx.profile_sampling(0)
    assert _code(e) == ERR_INVALID_ARG


def test_non_convergence_is_not_an_error(ctx):
    src, tgt, _ = synth.make_pair(3000, 3000, seed=2)
    ctx.set_params(ctx.default_params())
    ctx.set_source(src)
    # empty target: PCL's setInputTarget refuses it, align() returns with converged_ = false and the identity
    ctx.set_target(np.zeros((0, 4), np.float32))
    r = ctx.align()
    assert not r["converged"] and np.array_equal(r["T"], np.eye(4, dtype=np.float32))
    # empty source: no correspondences
    ctx.set_source(np.zeros((0, 4), np.float32))
    ctx.set_target(tgt)
    r = ctx.align()
    assert not r["converged"] and r["state"] == NO_CORRESPONDENCES and r["n_corr"] == 0
    # everything outside the gate: no correspondences either
    far = src.copy()
    far[:, :3] += 1000.0
    ctx.set_source(far)
    r = ctx.align(want_fitness=True)
    assert not r["converged"] and r["state"] == NO_CORRESPONDENCES and r["iterations"] == 0
    assert r["fitness"] > 1e5                       # getFitnessScore has no gate: ~1000 m away, squared
    # fewer correspondences than PCL's minimum (3)
    ctx.set_source(src[:2])
    r = ctx.align()
    assert not r["converged"] and r["state"] == NO_CORRESPONDENCES
    # all-NaN source
    ctx.set_source(np.full((100, 4), np.nan, np.float32))
    r = ctx.align()
    assert not r["converged"] and r["n_corr"] == 0


def test_fitness_and_reduce_on_a_fresh_context_with_an_empty_target(built):
    """an empty target never allocates its buffer; reduce_kernel's unconditional tgt[0] reads must not go
    to device address 0 (a memory fault that aborts the process).  getFitnessScore then reports PCL's DBL_MAX."""
    from icpslam_amd import Context
    src, _, _ = synth.make_pair(3000, 3000, seed=5)
    with Context(0) as c:                                   # fresh: the target buffer has never been allocated
        c.set_source(src)
        c.set_target(np.zeros((0, 4), np.float32))
        r = c.align()
        assert not r["converged"]
        assert c.fitness() == np.finfo(np.float64).max
        idx, d2 = c.nn(np.eye(4))
        assert (idx == -1).all()
        sums = c.reduce(np.eye(4), 1.0)
        assert np.array_equal(sums, np.zeros(17))
        _, tgt, _ = synth.make_pair(3000, 3000, seed=5)
        c.set_target(tgt)                                   # and the context is still alive
        assert c.align()["converged"]


def test_gicp_needs_twenty_points(ctx):
    src, tgt, _ = synth.make_pair(3000, 3000, seed=3)
    ctx.set_params(ctx.default_params(), method=GICP)
    ctx.set_source(src[:10])
    ctx.set_target(tgt)
    r = ctx.align()       # PCL: computeCovariances gives up below k_correspondences_ points, converged_ stays false
    assert not r["converged"] and np.array_equal(r["T"], np.eye(4, dtype=np.float32))
    with pytest.raises(IcpGpuError) as e:           # asked for explicitly, it is an argument error
        ctx.gicp_covariances(of_target=False)
    assert _code(e) == ERR_INVALID_ARG and "20" in str(e.value)
    ctx.set_source(src)
    assert ctx.align()["iterations"] >= 1


@pytest.mark.parametrize("mode", [NN_AUTO, NN_GRID])
def test_huge_and_degenerate_gates_still_give_the_brute_force_answer(ctx, mode):
    import oracle
    src, tgt, _ = synth.make_pair(5000, 5000, seed=4)
    for gate in (1e9, 1e-4):
        ctx.set_params(ctx.default_params(), nn_mode=mode, max_correspondence_distance=gate)
        ctx.set_source(src)
        ctx.set_target(tgt)
        r = ctx.align()
        o = oracle.icp_align(src, tgt, oracle.default_params(max_correspondence_distance=gate))
        assert r["iterations"] == o["iterations"] and r["n_corr"] == o["n_corr"] and r["converged"] == o["converged"]
        if o["n_corr"] >= 3:
            assert np.abs(r["T"] - o["T"]).max() <= 1e-4


def test_map_get_points_capacity_is_checked(ctx):
    L = _lib.load()
    ctx.map_reset(0.5)
    ctx.map_add_points(synth.make_pair(2000, 10, seed=5)[0])
    n = C.c_size_t()
    buf = np.zeros((4, 4), np.float32)
    rc = L.icpgpu_map_get_points(ctx._h, buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(4), C.byref(n))
    assert rc == ERR_INVALID_ARG and n.value == ctx.map_size() > 4


def test_callers_with_shorter_and_longer_structs(built):
    """include/icpgpu.h, "ABI rule", on a device: a context created by a caller whose structs are SHORTER than the library's (an older
    header: here icpgpu_result without gicp_solver and a profile without its last field -- the floor icpgpu_create_abi accepts)
    gets exactly its bytes written -- the guard bytes behind its structs stay untouched -- and the same numbers; a caller with LONGER
    structs gets the tail zeroed; batches honour the caller's stride."""
    L = _lib.load()
    src, tgt, _ = synth.make_pair(4000, 4000, seed=5)
    fp = C.POINTER(C.c_float)
    ptr = lambda a: a.ctypes.data_as(fp)
    with __import__("icpslam_amd").Context(0) as ref_ctx:
        ref_ctx.set_params(ref_ctx.default_params(), max_iterations=10)
        ref_ctx.set_source(src)
        ref_ctx.set_target(tgt)
        ref = ref_ctx.align(want_fitness=True)
    n_p, n_r, n_f = C.sizeof(_lib.Params), C.sizeof(_lib.Result), C.sizeof(_lib.Profile)
    for d_p, d_r, d_f in ((0, -8, -8), (16, 24, 40)):
        s_p, s_r, s_f = n_p + d_p, n_r + d_r, n_f + d_f
        h = C.c_void_p()
        assert L.icpgpu_create_abi(C.byref(h), 0, _lib.HEADER_VERSION, s_p, s_r, s_f) == 0
        try:
            pbuf = (C.c_ubyte * (s_p + 32))(*([0xCD] * (s_p + 32)))
            L.icpgpu_default_params_sz(C.cast(pbuf, C.POINTER(_lib.Params)), s_p)
            assert L.icpgpu_set_params(h, C.cast(pbuf, C.POINTER(_lib.Params))) == 0
            back = (C.c_ubyte * (s_p + 32))(*([0xCD] * (s_p + 32)))
            assert L.icpgpu_get_params(h, C.cast(back, C.POINTER(_lib.Params))) == 0
            assert bytes(back) == bytes(pbuf)
            assert L.icpgpu_set_source(h, ptr(src), src.shape[0]) == 0 and L.icpgpu_set_target(h, ptr(tgt), tgt.shape[0]) == 0
            rbuf = (C.c_ubyte * (s_r + 32))(*([0xCD] * (s_r + 32)))
            assert L.icpgpu_align(h, None, None, 1, C.cast(rbuf, C.POINTER(_lib.Result))) == 0
            assert set(rbuf[s_r:]) == {0xCD}                                       # nothing written behind the caller's struct
            if d_r > 0:
                assert set(rbuf[n_r:s_r]) == {0}
            res = _lib.Result.from_buffer_copy(bytes(rbuf[:n_r]) if d_r >= 0 else bytes(rbuf[:s_r]) + bytes(-d_r))
            assert np.array_equal(np.array(res.T, np.float32).reshape(4, 4).T, ref["T"])
            assert (res.iterations, res.n_correspondences, res.fitness) == (ref["iterations"], ref["n_corr"], ref["fitness"])
            fbuf = (C.c_ubyte * (s_f + 32))(*([0xCD] * (s_f + 32)))
            assert L.icpgpu_profile_get(h, C.cast(fbuf, C.POINTER(_lib.Profile))) == 0
            assert set(fbuf[s_f:]) == {0xCD}
            prof = _lib.Profile.from_buffer_copy(bytes(fbuf[:n_f]) if d_f >= 0 else bytes(fbuf[:s_f]) + bytes(-d_f))
            assert prof.aligns == 1 and prof.iterations == ref["iterations"]
            # a batch of three pairs: results at the CALLER's stride
            srcs = (fp * 3)(ptr(src), ptr(src), ptr(src))
            tgts = (fp * 3)(ptr(tgt), ptr(tgt), ptr(tgt))
            ns = (C.c_size_t * 3)(src.shape[0], src.shape[0], src.shape[0])
            nt = (C.c_size_t * 3)(tgt.shape[0], tgt.shape[0], tgt.shape[0])
            bbuf = (C.c_ubyte * (3 * s_r + 32))(*([0xCD] * (3 * s_r + 32)))
            assert L.icpgpu_align_batch(h, 3, srcs, ns, tgts, nt, 1, C.cast(bbuf, C.POINTER(_lib.Result))) == 0
            assert set(bbuf[3 * s_r:]) == {0xCD}
            for k in range(3):
                one = bytes(bbuf[k * s_r:(k + 1) * s_r])
                r = _lib.Result.from_buffer_copy(one[:n_r] if d_r >= 0 else one + bytes(-d_r))
                assert np.array_equal(np.array(r.T, np.float32).reshape(4, 4).T, ref["T"]) and r.fitness == ref["fitness"]
        finally:
            assert L.icpgpu_destroy(h) == 0


def test_view_entry_points_keep_the_error_convention():
    """icpgpu.h 1.1's icpgpu_align_view / icpgpu_voxel_grid_view: the same status codes as the calls they shadow, null outputs refused, the
    view pointer null and the count zero after a failure and for an empty result -- and the context works afterwards."""
    from icpslam_amd import Context
    L = _lib.load()
    FP = C.POINTER(C.c_float)
    with Context(0) as ctx:
        view, n = FP(), C.c_size_t(7)
        res = _lib.Result()
        assert L.icpgpu_align_view(ctx._h, None, 1, C.byref(res), C.byref(view), C.byref(n)) == ERR_NO_INPUT   # nothing set yet
        assert not view and n.value == 0
        src, tgt, _ = synth.make_pair(3000, 3000, seed=1)
        ctx.set_source(src)
        ctx.set_target(tgt)
        assert L.icpgpu_align_view(ctx._h, None, 1, None, C.byref(view), C.byref(n)) == ERR_INVALID_ARG
        assert L.icpgpu_align_view(ctx._h, None, 1, C.byref(res), None, C.byref(n)) == ERR_INVALID_ARG
        assert L.icpgpu_align_view(ctx._h, None, 1, C.byref(res), C.byref(view), None) == ERR_INVALID_ARG
        cloud = np.ascontiguousarray(src)
        p = cloud.ctypes.data_as(FP)
        assert L.icpgpu_voxel_grid_view(ctx._h, p, 3000, C.c_float(0.0), C.byref(view), C.byref(n)) == ERR_INVALID_ARG   # leaf
        assert not view and n.value == 0
        assert L.icpgpu_voxel_grid_view(ctx._h, None, 3000, C.c_float(0.2), C.byref(view), C.byref(n)) == ERR_INVALID_ARG
        assert L.icpgpu_voxel_grid_view(ctx._h, p, 3000, C.c_float(0.2), None, C.byref(n)) == ERR_INVALID_ARG
        assert L.icpgpu_voxel_grid_view(ctx._h, p, 0, C.c_float(0.2), C.byref(view), C.byref(n)) == 0 and not view and n.value == 0
        # an empty source: aligned "cloud" of zero points, no view
        ctx.set_source(np.empty((0, 4), np.float32))
        assert L.icpgpu_align_view(ctx._h, None, 0, C.byref(res), C.byref(view), C.byref(n)) == 0 and not view and n.value == 0
        # ... and the context still works: filtered view, then an alignment whose cloud comes back as a view
        got = ctx.voxel_grid_view(src, 0.5)
        assert 0 < len(got) < len(src)
        ctx.set_source(src)
        r = ctx.align_view(want_fitness=True)
        assert r["converged"] and r["cloud"].shape == src.shape and np.isfinite(r["fitness"])
