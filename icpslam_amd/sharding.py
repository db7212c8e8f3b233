"""Multi-GPU layer: independent scan pairs shard across ranks; the only collective is the result gather.

SURVEY.md section 8(e): each (source_k, target_k) is an independent ICP problem -- the reference processes them one
after another in IcpOdometer::laserCloudCallback (/root/reference/src/icpslam/icp_odometer.cpp:147-210) -- so pair k
goes to rank `owner(k)`, no data is exchanged during the solve, and one all_gather of fixed-size records (RCCL over
xGMI on GPUs, gloo in the CPU tests) returns every result to every rank.  A record is 23 float64 = 184 B; 512 pairs
are 94 KB: latency-bound, link bandwidth is irrelevant.
"""
from __future__ import annotations

import numpy as np

RECORD_LEN = 23   # pair_id, iterations, converged, state, n_corr, mse, fitness, T[16] (row-major 4x4)


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block partition (keeps consecutive scans of a sequence on one GPU; sizes differ by at most 1)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def owner_of(k: int, n_items: int, world: int) -> int:
    base, extra = divmod(n_items, world)
    split = extra * (base + 1)
    if k < split:
        return k // (base + 1)
    return extra + (k - split) // base if base else world - 1


def make_record(pair_id: int, res: dict) -> np.ndarray:
    r = np.empty(RECORD_LEN, np.float64)
    r[0] = pair_id
    r[1] = res["iterations"]
    r[2] = 1.0 if res["converged"] else 0.0
    r[3] = res["state"]
    r[4] = res["n_corr"]
    r[5] = res["mse"]
    r[6] = res["fitness"]
    r[7:23] = np.asarray(res["T"], np.float64).reshape(16)
    return r


def parse_record(r: np.ndarray) -> dict:
    return dict(pair_id=int(r[0]), iterations=int(r[1]), converged=bool(r[2] != 0.0), state=int(r[3]),
                n_corr=int(r[4]), mse=float(r[5]), fitness=float(r[6]), T=np.asarray(r[7:23]).reshape(4, 4).copy())


def gather_records(local: np.ndarray, n_items: int, rank: int, world: int, device=None) -> np.ndarray:
    """all_gather the per-rank record blocks into one (n_items, RECORD_LEN) array ordered by pair id.

    `local` holds this rank's records in shard order. Uses torch.distributed when world > 1 (backend chosen by the
    caller: nccl == RCCL on the GPU box, gloo in CPU tests). Blocks are padded to the largest shard so that a single
    fixed-size collective suffices.
    """
    local = np.ascontiguousarray(local, np.float64).reshape(-1, RECORD_LEN)
    if world == 1:
        out = local
    else:
        import torch
        import torch.distributed as dist
        cap = -(-n_items // world)
        pad = np.full((cap, RECORD_LEN), -1.0, np.float64)
        pad[: local.shape[0]] = local
        t = torch.from_numpy(pad)
        if device is not None:
            t = t.to(device)
        bufs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(bufs, t)
        parts = []
        for r in range(world):
            n_r = len(shard_range(n_items, r, world))
            parts.append(bufs[r].cpu().numpy()[:n_r])
        out = np.concatenate(parts, axis=0) if parts else np.zeros((0, RECORD_LEN))
    if out.shape[0] != n_items:
        raise RuntimeError(f"gather size mismatch: {out.shape[0]} != {n_items}")
    ids = out[:, 0].astype(np.int64)
    if not np.array_equal(ids, np.arange(n_items)):
        raise RuntimeError("gathered records are not a permutation-free cover of the pair ids")
    return out


def run_sharded(n_items: int, rank: int, world: int, load_pair, align_pair, device=None) -> np.ndarray:
    """Process this rank's shard with `align_pair(src, tgt) -> result dict`, then gather everything."""
    recs = []
    for k in shard_range(n_items, rank, world):
        src, tgt = load_pair(k)
        recs.append(make_record(k, align_pair(src, tgt)))
    local = np.stack(recs) if recs else np.zeros((0, RECORD_LEN))
    return gather_records(local, n_items, rank, world, device)


COMM_NONE, COMM_RCCL, COMM_HOST = 0, 1, 2


def align_batch_multi(devices, sources, targets, params=None, want_fitness: bool = False, communicator: int = COMM_RCCL):
    """icpgpu_align_batch_multi (include/icpgpu.h): the single-process, one-host-thread-per-GPU form of the layer above --
    contiguous shards over `devices`, one all-gather of the 184-byte records (RCCL, or host-staged for tests).
    Returns (list of result dicts, records (n, RECORD_LEN) or None)."""
    import ctypes as C

    from . import _lib
    from .registration import _as_cloud, _fp, result_dict
    L = _lib.load()
    n = len(sources)
    srcs = [_as_cloud(s) for s in sources]
    tgts = [_as_cloud(t) for t in targets]
    FP = C.POINTER(C.c_float)
    sp = (FP * n)(*[_fp(s) for s in srcs])
    tp = (FP * n)(*[_fp(t) for t in tgts])
    ns = (C.c_size_t * n)(*[s.shape[0] for s in srcs])
    nt = (C.c_size_t * n)(*[t.shape[0] for t in tgts])
    res = (_lib.Result * n)()
    recs = np.zeros((n, RECORD_LEN), np.float64) if communicator != COMM_NONE else None
    dev = (C.c_int * len(devices))(*devices)
    rc = L.icpgpu_align_batch_multi(dev, len(devices), C.byref(params) if params is not None else None, n, sp, ns, tp, nt,
                                    int(want_fitness), res, recs.ctypes.data_as(C.POINTER(C.c_double)) if recs is not None else None,
                                    communicator)
    if rc != 0:
        from .registration import IcpGpuError
        raise IcpGpuError(rc, (L.icpgpu_multi_last_error() or b"").decode())
    return [result_dict(r, None) for r in res], recs
