import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def cpu_ops_backend():
    """Install the numpy stand-in ops (tests/cpu_ops.py) for host-logic tests; restore afterwards."""
    import cpu_ops
    from hpfrec_amd import cython_loops_float as backend
    old = backend.HipOps
    backend.HipOps = lambda device=None: cpu_ops.CpuOps()
    yield backend
    backend.HipOps = old


@pytest.fixture
def hip_backend():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hpfrec_amd import cython_loops_float as backend
    from hpfrec_amd.ops_hip import HipOps
    assert backend.HipOps is HipOps
    return backend


@pytest.fixture(params=["standin", pytest.param("hip", marks=pytest.mark.gpu)])
def any_backend(request):
    """Run a host-level test twice: on the numpy stand-in ops (no GPU) and, under -m gpu, on the HIP path."""
    import cpu_ops
    from hpfrec_amd import cython_loops_float as backend
    from hpfrec_amd.ops_hip import HipOps
    old = backend.HipOps
    if request.param == "standin":
        backend.HipOps = lambda device=None: cpu_ops.CpuOps()
    else:
        import torch
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
        backend.HipOps = HipOps
    yield backend
    backend.HipOps = old
