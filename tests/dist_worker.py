"""Worker for tests/test_distributed.py: one rank of a world_size-N gloo job on CPU.

Every rank builds the SAME global batch (seeded), takes its channel range, runs it through the engine
(CPU wave emulation here; on a GPU node the identical code path runs libdigiham_amd.so per device),
and rank 0 gathers the per-rank outputs to compare with an unsharded run.  The data path has no
collective: only the barrier and the timing/units reduction used by bench.py.
"""
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch.distributed as dist   # noqa: E402

from digiham_amd import shard      # noqa: E402
import hostemu                     # noqa: E402
from common import make_channels, run_engine   # noqa: E402


def main(out_path):
    rank, world, _ = shard.init_process_group("gloo")
    x = make_channels("dmr", list(range(1, 8)), 16)          # 7 channels: uneven split on purpose
    lo, hi = shard.channel_range(x.shape[0], rank, world)
    shard.barrier()
    t0 = time.perf_counter()
    res = run_engine(hostemu.context(), np.ascontiguousarray(x[lo:hi]), "dmr", [5000])
    dt = time.perf_counter() - t0
    shard.barrier()
    dt_max, units = shard.reduce_report(dt, (hi - lo) * x.shape[1])
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, res["syms"], res["frames"], [e.tobytes() for e in res["events"]]))
    if rank == 0:
        pickle.dump({"parts": gathered, "dt_max": dt_max, "units": units, "total": x.shape[0] * x.shape[1]}, open(out_path, "wb"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
