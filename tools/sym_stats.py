#!/usr/bin/env python3
"""tools/sym_stats.py [proto] [channels]: push the bench workload twice and print, per channel, how many of its runs were symbol-major
(state header word 20), how many symbols / runs / blocks went through exact arithmetic, and the push time."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from digiham_amd import api, synth_torch

proto = sys.argv[1] if len(sys.argv) > 1 else "dmr"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ctx = api.Context(device=0)
dev = torch.device("cuda", 0)
x, info = synth_torch.make_batch(torch, dev, proto, B, {"dmr": 132, "ysf": 40}[proto], seed=1007, sps=10)
T = info["samples_per_channel"]
eng = api.Engine(B, T, ctx=ctx, rrc="wide", demod="gfsk", sps=10, proto=proto)
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.push(x); eng.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    runs = eng.debug_header(20); unc = eng.debug_header(16); ex = eng.debug_header(17); exb = eng.debug_header(18)
    print("push %d: %.3f ms  sym runs/channel min %d max %d (of %d)  uncertain %.2f  exact runs %.2f  exact blocks %.3f"
          % (k, dt * 1e3, runs.min(), runs.max(), (k + 1) * T // 1000, unc.mean(), ex.mean(), exb.mean()))
