"""Threading contract of the boundary (include/icpgpu.h): the reference runs the odometer's ICP on
an AsyncSpinner worker thread while the mapper's ICP runs on the main thread (/root/reference/src/icpslam_node.cpp:9,
src/icpslam/icpslam.cpp:135) -- distinct contexts must be usable concurrently and give the answers they give alone."""
import threading

import numpy as np
import pytest

from icpslam_amd import Context, GICP, synth

pytestmark = pytest.mark.gpu


def _job(kind, seed):
    if kind == "odometer":          # 10 iterations + fitness on 40k-point scans
        src, tgt, _ = synth.make_pair(40000, 40000, seed=seed)
        return dict(src=src, tgt=tgt, params=dict(max_iterations=10), fitness=True)
    if kind == "mapper":            # 30 iterations against a bigger target
        src, tgt, _ = synth.make_pair(30000, 120000, seed=seed)
        return dict(src=src, tgt=tgt, params=dict(max_iterations=30), fitness=False)
    src, tgt, _ = synth.make_pair(8000, 8000, seed=seed)
    return dict(src=src, tgt=tgt, params=dict(max_iterations=10, method=GICP), fitness=False)


def _run(ctx, job, reps):
    out = []
    for _ in range(reps):
        ctx.set_params(ctx.default_params(), **job["params"])
        ctx.set_source(job["src"])
        ctx.set_target(job["tgt"])
        r = ctx.align(want_fitness=job["fitness"])
        out.append((r["T"].copy(), r["iterations"], r["n_corr"], r["fitness"]))
    return out


def test_distinct_contexts_run_concurrently_and_reproducibly():
    jobs = [_job("odometer", 1), _job("mapper", 2), _job("gicp", 3), _job("odometer", 4)]
    ctxs = [Context(0) for _ in jobs]
    try:
        alone = [_run(c, j, 1)[0] for c, j in zip(ctxs, jobs)]
        results, errors = [None] * len(jobs), []

        def work(i):
            try:
                results[i] = _run(ctxs[i], jobs[i], 6)
            except Exception as e:  # pragma: no cover - reported below
                errors.append((i, repr(e)))

        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not errors and all(r is not None for r in results)
        for i, reps in enumerate(results):
            for T, it, nc, fit in reps:      # bitwise the single-threaded answer, every time
                assert np.array_equal(T, alone[i][0]) and it == alone[i][1] and nc == alone[i][2]
                assert (np.isnan(fit) and np.isnan(alone[i][3])) or fit == alone[i][3]
    finally:
        for c in ctxs:
            c.close()
