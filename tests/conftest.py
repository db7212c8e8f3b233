import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class _DevFlavour:
    """Tests of the DEVELOPMENT modes (environment switches that exist only in libicpgpu_dev.so, icpslam_amd/csrc/icp_env.h:
    the every-pair bound check of the bf16 brute force, the experimental tile search, forced variants) cannot run in this
    process, which has the release library loaded.  `dev_flavour.delegated` re-runs the requesting test in a sub-process with
    ICPGPU_FLAVOUR=dev and fails with its output if it fails there; inside that sub-process it is False and the body runs.

        def test_x(built, dev_flavour):
            if dev_flavour.delegated:
                return
            ...body, free to set the development switches in os.environ...
    """

    def __init__(self, request):
        self._request = request

    @property
    def delegated(self) -> bool:
        import subprocess
        from icpslam_amd import _lib
        if _lib.FLAVOUR == "dev":
            return False
        env = dict(os.environ, ICPGPU_FLAVOUR="dev", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        node = self._request.node.nodeid
        out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", node], env=env,
                             cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0 and " passed" in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])
        return True


@pytest.fixture
def dev_flavour(request, built):
    return _DevFlavour(request)


@pytest.fixture(scope="session")
def built():
    """Native pieces are built once per session (no-op when up to date)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def ctx(built):
    from icpslam_amd import Context
    c = Context(0)
    yield c
    c.close()
