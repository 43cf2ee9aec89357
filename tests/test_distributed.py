"""N > 1 path on CPU: two gloo ranks shard the channels; results must equal the unsharded run."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np

from common import make_channels, run_engine
from digiham_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_channel_range_partitions_exactly():
    for total in (1, 7, 16384, 65536, 65537):
        for world in (1, 2, 3, 4, 8):
            spans = [shard.channel_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_job_matches_single_process(tmp_path, emu_ctx):
    port = socket.socket()
    port.bind(("127.0.0.1", 0))
    p = port.getsockname()[1]
    port.close()
    out = tmp_path / "dist.pkl"
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(p))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(out)], env=env))
    for pr in procs:
        assert pr.wait(timeout=300) == 0
    got = pickle.load(open(out, "rb"))
    assert got["units"] == got["total"]            # every unit counted exactly once across ranks
    assert got["dt_max"] > 0
    x = make_channels("dmr", list(range(1, 8)), 16)
    ref = run_engine(emu_ctx, x, "dmr", [5000])
    seen = 0
    for lo, hi, syms, frames, events in sorted(got["parts"]):
        assert lo == seen
        seen = hi
        for i, ch in enumerate(range(lo, hi)):
            assert (syms[i] == ref["syms"][ch]).all()
            assert (frames[i] == ref["frames"][ch]).all()
            assert events[i] == ref["events"][ch].tobytes()
    assert seen == x.shape[0]
