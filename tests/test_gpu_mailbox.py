"""The result mailbox (icpslam_amd/csrc/icp_kernels.h: mailbox_tag): every value the host waits for -- the 17 sums of an ICP
sweep (a4), a GICP evaluation's partial sums, the device solver's result -- is a 16-byte pair {bits, tag(number, checksum of the
bits)} that the reader accepts only when number and checksum fit, so that it is valid or recognisably not.  Two device-side
forms: one 16-byte write-through store (default) and value / system-scope release / tag (ICPGPU_MAILBOX=release, or chosen by
the start-up self-test if it ever sees a torn pair)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = (
    "import sys, numpy as np\n"
    "from icpslam_amd import Context, GICP, synth\n"
    "out = {}\n"
    "with Context(0) as ctx:\n"
    "    for n in (3000, 40000):\n"
    "        src, tgt, _ = synth.make_pair(n, n, seed=40 + n)\n"
    "        ctx.set_params(ctx.default_params(), max_iterations=10)\n"
    "        ctx.set_source(src); ctx.set_target(tgt)\n"
    "        r = ctx.align(want_fitness=True)\n"
    "        out['p%d' % n] = np.concatenate([r['T'].ravel(), [r['iterations'], r['n_corr'], r['fitness'], r['mse']]])\n"
    "        res = ctx.align_batch([src, tgt], [tgt, src], want_fitness=True)\n"
    "        out['b%d' % n] = np.concatenate([res[0]['T'].ravel(), res[1]['T'].ravel(), [res[0]['fitness'], res[1]['fitness']]])\n"
    "    src, tgt, _ = synth.make_pair(9000, 9000, seed=7)\n"
    "    ctx.set_params(ctx.default_params(), method=GICP, max_iterations=10)\n"
    "    ctx.set_source(src); ctx.set_target(tgt)\n"
    "    r = ctx.align(want_fitness=True)\n"
    "    out['g'] = np.concatenate([r['T'].ravel(), [r['iterations'], r['n_corr'], r['fitness']]])\n"
    "np.savez(sys.argv[1], **out)\n")


def _run(tmp_path, name, **env):
    e = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), ICPGPU_DEBUG="1", **env)
    path = str(tmp_path / (name + ".npz"))
    res = subprocess.run([sys.executable, "-c", CODE, path], env=e, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    return dict(np.load(path)), res.stderr


def test_release_form_and_pairs_give_the_same_bits_and_the_self_test_sees_no_torn_pair(tmp_path):
    auto, log = _run(tmp_path, "auto")
    line = [l for l in log.splitlines() if "mailbox self-test" in l]
    assert line and " 0 torn" in line[0] and "16-byte pairs" in line[0], log[-1500:]     # this platform delivers pairs whole
    seen = int(line[0].split(":")[1].split("of")[0])
    assert seen >= 10, line[0]                                                           # (the host really watched the slot change)
    rel, log_r = _run(tmp_path, "release", ICPGPU_MAILBOX="release")
    assert "mailbox self-test" not in log_r                                              # forced: no test
    dev, _ = _run(tmp_path, "device", ICPGPU_MAILBOX="release", ICPGPU_GICP_DEVICE="1")  # the device solver's result granules too
    for k in auto:
        assert np.array_equal(auto[k], rel[k], equal_nan=True), k
        assert np.array_equal(auto[k], dev[k], equal_nan=True), k
