#!/usr/bin/env python3
"""Rate of the engine when the boundary hands over HOST buffers (dh_engine_push_host): H2D copy + kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch
B = 4096
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), "dmr", B, 132)
T = info["samples_per_channel"]
xh = x.cpu().numpy()
eng = api.Engine(B, T, proto="dmr")
for pinned in (False, True):
    src = xh
    if pinned:
        t = torch.from_numpy(xh).pin_memory(); src = t.numpy()
    eng.push_host(src); eng.sync()
    t0 = time.perf_counter()
    for _ in range(4):
        eng.push_host(src)
    eng.sync()
    dt = (time.perf_counter() - t0) / 4
    print("pinned" if pinned else "pageable", "ms/push %.1f" % (dt * 1e3), "Gsamples/s %.2f" % (B * T / dt / 1e9), "GB/s %.1f" % (B * T * 4 / dt / 1e9), flush=True)
