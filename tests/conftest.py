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


def spawn_ranks(fn, args_for_port, world, out_dir):
    """torch.multiprocessing.spawn of `world` ranks with a fresh rendezvous port; the workers leave their tracebacks in
    out_dir/rank*.err (tests/dist_worker.py).  A failure of the RENDEZVOUS itself (port taken between the probe and the
    bind, connection refused/timeouts of the gloo store) is retried once on another port; anything else -- an
    exception of the code under test, a wrong result -- is not."""
    import glob
    import socket
    import torch.multiprocessing as mp
    infra = ("Address already in use", "Connection refused", "Connection reset", "timed out", "Timed out",
             "connect() failed", "Broken pipe", "store", "rendezvous")
    for attempt in (0, 1):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        for f in glob.glob(os.path.join(out_dir, "rank*.err")):
            os.remove(f)
        try:
            mp.spawn(fn, args=args_for_port(port), nprocs=world, join=True)
            return
        except Exception as exc:   # noqa: BLE001
            errs = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(out_dir, "rank*.err"))))
            print(errs)
            text = errs + str(exc)
            if attempt == 0 and any(t in text for t in infra) and "AssertionError" not in text and "HpfHipError" not in text:
                print("rendezvous failure, retrying once on another port")
                continue
            raise
