"""Run by tests/test_hip_parity.py::test_sharded_path_single_rank_nccl under torch.distributed.run with
one rank and HPF_FORCE_SHARDED=1: the RCCL process group, the exchange schedules and the packed-stride kernels run on
real hardware; results must equal the ordinary single-GPU path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import datagen  # noqa: E402
from hpfrec_amd import cython_loops_float as be  # noqa: E402


ITS = 5


def fit():
    df, nU, nI = datagen.mid_counts()
    Y, iu, ii = datagen.triplets(df)
    k = 50
    Theta = np.empty((nU, k), np.float32)
    Beta = np.empty((nI, k), np.float32)
    i, temp, llk = be.fit_hpf(0.3, 0.3, 1.0, 0.3, 0.3, 1.0, Y, iu, ii, Theta, Beta, ITS, "maxiter", ITS, 1e-3, 0, 0, None, 0,
                              np.zeros(1, np.uint64), "", 123, 1, 1, 0, 0, np.empty(0, np.float32),
                              np.empty(0, np.uint64), np.empty(0, np.uint64), 0, 1, 0)
    return (Theta, Beta) + tuple(temp), float(llk)


if __name__ == "__main__":
    torch.cuda.set_device(0)
    os.environ["HPF_FORCE_SHARDED"] = "0"
    plain, llk0 = fit()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    os.environ["HPF_FORCE_SHARDED"] = "1"
    from hpfrec_amd import cavi
    seen = []
    orig = cavi.FullBatchCavi._scatter_views

    def noted(self):
        v = orig(self)
        seen.append(self.schedule)
        return v
    cavi.FullBatchCavi._scatter_views = noted
    sharded, llk1 = fit()
    print("SCHEDULE %s" % (seen[-1] if seen else None))
    if cavi._DIRECT_COMMS:
        print("DIRECT_RCCL_USED")
    if cavi.NATIVE_PLANS_CREATED[0] > 0:
        print("NATIVE_PLAN_USED")
    from hpfrec_amd import shard
    print("CHECKED_ITERATIONS %d" % max(shard._PASSED.values(), default=0))
    dist.destroy_process_group()
    worst = max(float(np.max(np.abs(a - b) / np.abs(b))) for a, b in zip(sharded, plain))
    print("SHARDED_VS_PLAIN max-rel %.3e llk-rel %.3e" % (worst, abs(llk1 / llk0 - 1)))
    assert worst < 1e-5 and abs(llk1 / llk0 - 1) < 1e-6
    print("SHARDED_OK")
