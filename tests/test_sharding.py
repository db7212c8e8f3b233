"""CPU tests of the multi-GPU layer: partitioning, record packing, and the world_size-2 gather over gloo.
The per-pair compute is stood in for by the CPU oracle (test infrastructure) -- the sharding logic is what is tested."""
import os
import socket
import sys

import numpy as np
import pytest

from icpslam_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n,world", [(0, 1), (1, 4), (7, 2), (8, 8), (512, 8), (1999, 8), (5, 3)])
def test_shard_ranges_partition(n, world):
    seen = []
    for r in range(world):
        rg = sharding.shard_range(n, r, world)
        seen += list(rg)
        for k in rg:
            assert sharding.owner_of(k, n, world) == r
    assert seen == list(range(n))
    sizes = [len(sharding.shard_range(n, r, world)) for r in range(world)]
    assert max(sizes) - min(sizes) <= 1


def test_record_roundtrip():
    res = dict(iterations=7, converged=True, state=2, n_corr=4321, mse=0.0123, fitness=0.5,
               T=np.arange(16, dtype=np.float32).reshape(4, 4))
    back = sharding.parse_record(sharding.make_record(42, res))
    assert back["pair_id"] == 42 and back["iterations"] == 7 and back["converged"] and back["n_corr"] == 4321
    np.testing.assert_array_equal(back["T"], res["T"])


def test_single_rank_gather_is_identity():
    recs = np.stack([sharding.make_record(k, dict(iterations=k, converged=False, state=0, n_corr=0, mse=0, fitness=0,
                                                  T=np.eye(4))) for k in range(5)])
    out = sharding.gather_records(recs, 5, 0, 1)
    np.testing.assert_array_equal(out, recs)


def _worker(rank, world, port, n_items, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle
    from icpslam_amd import sharding as sh
    from icpslam_amd import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        def load(k):
            s, t, _ = synth.make_pair(600, 600, seed=1000 + k)
            return s, t

        def align(s, t):
            return oracle.icp_align(s, t, oracle.default_params(max_iterations=5))
        out = sh.run_sharded(n_items, rank, world, load, align)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_world2_gloo_gather_matches_serial():
    import torch.multiprocessing as mp

    import oracle
    from icpslam_amd import synth
    n_items, world = 5, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(outs[0], outs[1])          # every rank holds every result
    for k in range(n_items):
        s, t, _ = synth.make_pair(600, 600, seed=1000 + k)
        ref = oracle.icp_align(s, t, oracle.default_params(max_iterations=5))
        rec = sharding.parse_record(outs[0][k])
        assert rec["pair_id"] == k and rec["iterations"] == ref["iterations"] and rec["n_corr"] == ref["n_corr"]
        np.testing.assert_array_equal(rec["T"].astype(np.float32), ref["T"])
