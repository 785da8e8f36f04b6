"""Side streams, ONE per (device, purpose) for the life of the process.

ROCm maps a new stream onto a hardware queue by creation order (the streams of one priority are multiplexed over four
queues), and a queue shared with the compute stream serialises what was meant to overlap: the exchange streams of a SECOND
sharded model of a process ran its iteration at 1.1 ms instead of 0.64 (tools/shard_probe.py with emulated link time,
DESIGN.md section 6.2), and bench.py builds ~25 models in a row.  Only one fit runs at a time in a process, so fits share
their side streams instead of creating new ones.
"""
import torch

_SIDE_STREAMS = {}


def side_stream(device, kind, priority=0):
    """The process-wide stream of `kind` on `device` (created on first use; priority -1: its own hardware-queue pool)."""
    key = (str(torch.device(device)), kind)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=priority)
    return _SIDE_STREAMS[key]
