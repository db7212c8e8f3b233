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


def test_device_solver_contexts_that_share_an_xcd_run_concurrently(tmp_path):
    """The one-XCD variant of the GICP device solver (ICPGPU_GICP_DEVICE=1) confines a run's workgroups to the context's XCD
    (context serial mod 8): contexts 0 and 8 share XCD 0.  Two threads drive them at once, several registrations each: every
    result must be the single-threaded one bit for bit -- if the two runs ever starve each other of CUs, the 50 ms gather timeout
    and the fallback (any-placement variant, then the host solver) must turn that into a slower run, never into a wrong one or
    a hang.  Sub-process: the switch is read once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import threading, numpy as np\n"
        "from icpslam_amd import Context, GICP, synth\n"
        "ctxs = [Context(0) for _ in range(9)]\n"
        "jobs = [synth.make_pair(20000, 20000, seed=81)[:2], synth.make_pair(14000, 18000, seed=82)[:2]]\n"
        "def run(c, j, reps):\n"
        "    out = []\n"
        "    for _ in range(reps):\n"
        "        c.set_params(c.default_params(), method=GICP, max_iterations=10)\n"
        "        c.set_source(j[0]); c.set_target(j[1])\n"
        "        r = c.align(want_fitness=True)\n"
        "        out.append((r['T'].tobytes(), r['iterations'], r['n_corr'], r['fitness']))\n"
        "    return out\n"
        "pair = [ctxs[0], ctxs[8]]\n"
        "alone = [run(c, j, 1)[0] for c, j in zip(pair, jobs)]\n"
        "res = [None, None]\n"
        "def work(i): res[i] = run(pair[i], jobs[i], 8)\n"
        "ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]\n"
        "[t.start() for t in ts]; [t.join(timeout=200) for t in ts]\n"
        "assert all(r is not None for r in res)\n"
        "for i in range(2):\n"
        "    assert all(x == alone[i] for x in res[i]), i\n"
        "print('device solves', [int(c.profile().gicp_device_solves) for c in pair])\n"
        "print('ok')\n")
    env = dict(os.environ, ICPGPU_GICP_DEVICE="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-500:], out.stderr[-2000:])
    print(out.stdout.strip().splitlines()[-2])
