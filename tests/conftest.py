import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Soaks only (MDVT_SEGV_TRACE=1): a native backtrace if the process dies of a signal (tools/probe/segv_trace.c) -- installed
    after pytest's faulthandler, which it hands the signal on to."""
    if os.environ.get("MDVT_SEGV_TRACE") != "1":
        return
    import ctypes
    import subprocess
    so = f"/tmp/libsegv_trace_{os.getpid()}.so"
    subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, os.path.join(REPO, "tools", "probe", "segv_trace.c")])
    out = os.path.join(os.environ.get("MDVT_SEGV_TRACE_DIR", "/tmp"), f"segv_trace_{os.getpid()}.log")      # (pytest captures fd 2)
    ctypes.CDLL(so).segv_trace_install(out.encode())


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def orc():
    """The plain-C oracle (built on demand with gcc)."""
    from oracle import c_oracle
    c_oracle.build()
    return c_oracle


@pytest.fixture(autouse=True, scope="session")
def _whole_suite_on_another_grid():
    """MDVT_TEST_SUBPIXEL_BITS=4 python -m pytest tests -m gpu: every StereoRerenderer the GPU tests make renders on the 4-bit
    sub-pixel grid (mdvt_config.subpixel_bits) unless the test chose one itself, and -- because the tests hand the renderer's
    grid to the oracle -- is compared with the oracle on that grid.  That is how the second copy of the rasterising kernels
    (csrc/Makefile: *_g4.o) is held to everything the default grid is held to."""
    bits = os.environ.get("MDVT_TEST_SUBPIXEL_BITS")
    if not bits:
        yield
        return
    from metric_depth_video_toolbox_amd import stereo_rerender
    orig = stereo_rerender.StereoRerenderer.__init__

    def init(self, *a, **kw):
        kw.setdefault("subpixel_bits", int(bits))
        orig(self, *a, **kw)
    stereo_rerender.StereoRerenderer.__init__ = init
    yield
    stereo_rerender.StereoRerenderer.__init__ = orig
