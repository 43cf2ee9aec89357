#!/usr/bin/env python3
"""A/B: one-wavefront chain kernel vs. separate slicer + decoder launches (same library, same box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch
proto = sys.argv[1] if len(sys.argv) > 1 else "dmr"
B = 16384
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), proto, B, 132 if proto == "dmr" else 40, seed=1000)
T = info["samples_per_channel"]
for split in (True, False, True, False):
    eng = api.Engine(B, T, proto=proto, split_stages=split)
    eng.timing_enable(8)
    for _ in range(2): eng.push(x)
    eng.sync(); eng.timing_read()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.push(x)
    eng.sync(); dt = (time.perf_counter() - t0) / 5
    a, b, c = eng.timing_read()
    f, fc = eng.frames()
    print("split" if split else "chain", "ms/step %.2f" % (dt * 1e3), "slicer %.2f decoder %.2f" % (b.mean(), c.mean()), "frame bytes", int(fc.sum()), flush=True)
    eng.close()
