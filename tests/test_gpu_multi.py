"""icpgpu_align_batch_multi (include/icpgpu.h): one process, one host thread per device entry, contiguous
shards, one all-gather of the result records.  The GPU box has ONE GPU: the sharding / padding / gather logic runs with
several entries naming device 0 and the host-staged communicator; the RCCL path (ncclCommInitAll + ncclAllGather, loaded
with dlopen) runs over the one device."""
import numpy as np
import pytest

from icpslam_amd import Context, IcpGpuError, sharding, synth

pytestmark = pytest.mark.gpu


def _pairs(n_pairs, n_pts):
    out = [synth.make_pair(n_pts + 37 * k, n_pts, seed=400 + k)[:2] for k in range(n_pairs)]
    return [p[0] for p in out], [p[1] for p in out]


@pytest.mark.parametrize("n_dev,n_pairs", [(3, 7), (2, 2), (4, 3), (1, 5)])
def test_multi_matches_single_context_batch_and_gathers_every_record(built, n_dev, n_pairs):
    srcs, tgts = _pairs(n_pairs, 6000)
    with Context(0) as c:
        c.set_params(c.default_params(), max_iterations=8)
        want = c.align_batch(srcs, tgts, want_fitness=True)
        p = c.get_params() if hasattr(c, "get_params") else None
    from icpslam_amd import _lib
    import ctypes as C
    P = _lib.Params()
    _lib.load().icpgpu_default_params(C.byref(P))
    P.max_iterations = 8
    got, recs = sharding.align_batch_multi([0] * n_dev, srcs, tgts, params=P, want_fitness=True, communicator=sharding.COMM_HOST)
    assert len(got) == n_pairs and recs.shape == (n_pairs, sharding.RECORD_LEN)
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.array_equal(g["T"], w["T"]) and g["iterations"] == w["iterations"] and g["n_corr"] == w["n_corr"]
        assert g["fitness"] == w["fitness"]
        r = sharding.parse_record(recs[k])
        assert r["pair_id"] == k and r["iterations"] == w["iterations"] and r["n_corr"] == w["n_corr"]
        assert np.array_equal(r["T"].astype(np.float32), w["T"]) and r["fitness"] == w["fitness"]


def test_multi_rccl_communicator_on_the_one_device(built):
    srcs, tgts = _pairs(4, 5000)
    got, recs = sharding.align_batch_multi([0], srcs, tgts, want_fitness=True, communicator=sharding.COMM_RCCL)
    got2, recs2 = sharding.align_batch_multi([0], srcs, tgts, want_fitness=True, communicator=sharding.COMM_HOST)
    assert np.array_equal(recs, recs2) and np.array_equal(recs[:, 0], np.arange(4))
    for a, b in zip(got, got2):
        assert np.array_equal(a["T"], b["T"])


def test_multi_errors_are_codes(built):
    srcs, tgts = _pairs(2, 3000)
    with pytest.raises(IcpGpuError):
        sharding.align_batch_multi([0, 0], srcs, tgts, communicator=sharding.COMM_RCCL)   # a device named twice: RCCL refuses
    with pytest.raises(IcpGpuError):
        sharding.align_batch_multi([99], srcs, tgts, communicator=sharding.COMM_NONE)
    got, recs = sharding.align_batch_multi([0, 0], [], [], communicator=sharding.COMM_HOST)   # nothing to do
    assert got == [] and recs.shape == (0, sharding.RECORD_LEN)
    got, recs = sharding.align_batch_multi([0], srcs, tgts, communicator=sharding.COMM_NONE)
    assert recs is None and len(got) == 2
