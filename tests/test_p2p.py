"""The primitives of the direct exchange (include/hpf_hip.h "Multi-GPU, direct exchange"; hpfrec_amd/p2p.py,
csrc/hpf_p2p.hip) on their own: flags, pulls of a peer's buffer, the k-float all-reduce by granules, the time-out.  The
sharded fits of test_hip_parity.py / test_full_size.py exercise them inside an iteration; here a failure names the primitive.
Nothing in the reference corresponds (single-node OpenMP, cython_loops.pxi:4)."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_control_block_layout_is_what_the_header_says():
    """No device needed: the size of a control block follows from the header's constants (flags[kind][src] words after a
    16-word header, then 8-byte granules vec[which][parity][src][ld])."""
    from hpfrec_amd import _lib
    L = _lib.lib()
    max_ranks, nkinds, nvec = 16, 32, 2
    for ld in (32, 64, 128, 256, 1024):
        want = (16 + nkinds * max_ranks) * 4 + nvec * 2 * max_ranks * ld * 8
        assert L.hpf_hip_p2p_ctrl_bytes(ld) == want
    assert L.hpf_hip_p2p_ctrl_bytes(0) < 0
    # argument checks come before any device call
    h = ctypes.c_void_p()
    for world, rank, ld, nbytes in ((0, 0, 64, 1024), (17, 0, 64, 1024), (2, 2, 64, 1024), (2, 0, 0, 1024), (2, 0, 64, 0)):
        assert L.hpf_hip_p2p_region_create(world, rank, ld, nbytes, ctypes.byref(h)) == -1
    assert L.hpf_hip_p2p_region_create(2, 0, 64, 1024, None) == -1
    assert L.hpf_hip_p2p_region_destroy(None) in (0, -1)


@pytest.mark.gpu
def test_region_connected_to_itself():
    """A region standing alone (probes, the bench's compute-only twin): every peer is the local memory, nothing is
    waited for, the all-reduce over "the ranks" is the local value."""
    import torch
    from hpfrec_amd import p2p
    dev = torch.device("cuda", 0)
    ld, n = 64, 1 << 16
    reg = p2p.PeerRegion(dev, 2 * n * 4, ld, rank=1, world=4, local=True, timeout_ms=2000)
    assert reg.local
    a, b = reg.tensor(0, (n,)), reg.tensor(n * 4, (n,))
    a.copy_(torch.arange(n, device=dev, dtype=torch.float32))
    b.zero_()
    for peer in range(4):       # every "peer" buffer is this one
        assert reg.data_ptr(peer) == reg.data_ptr()
    e = reg.next_epoch()
    reg.signal(p2p.FLAG_USER, e)
    reg.wait(p2p.FLAG_USER, e)
    got = torch.empty(n, device=dev)
    reg.pull(got, 2, 0, kind=p2p.FLAG_USER, epoch=e)
    assert torch.equal(got, a)
    vec = torch.arange(ld, device=dev, dtype=torch.float32) * 0.5
    want = vec.clone()
    for which in (0, 1):
        e = reg.next_epoch()
        reg.allreduce_vec(which, e, vec)
        assert torch.equal(vec, want)
    reg.status()
    with pytest.raises(p2p.P2PError):
        reg.tensor(8, (4,))                     # not 16-byte aligned
    with pytest.raises(p2p.P2PError):
        reg.tensor(0, (2 * n + 4,))             # past the end
    del a, b
    reg.close()
    reg.close()                                 # idempotent


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_primitives_between_processes_sharing_the_gpu(world):
    """tools/p2p_probe.py: `world` processes map one another's regions (hipIpc of the coarse-grained data buffer and the
    fine-grained control block); each pulls every peer's buffer after its flag and finds the peer's values, the granule
    all-reduce gives the rank-order sum (the same floats on every rank), and a wait nobody satisfies comes back as
    HPF_ETIMEOUT within its budget instead of hanging."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "p2p_probe.py"), str(world)], env=env, cwd=ROOT,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:]
    assert "world %d: exit 0" % world in out.stdout, out.stdout[-3000:]
    assert "time-out path OK" in out.stdout
