#!/usr/bin/env python3
"""Shader-clock breakdown of the DMR decoder kernel (diagnostic build, see tools/phase_clocks.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import _capi, api, synth_torch
lib = _capi.load(sys.argv[1]); ctx = api.Context(lib=lib)
B = 16384
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), "dmr", B, 132, seed=1000)
eng = api.Engine(B, info["samples_per_channel"], ctx=ctx, proto="dmr", split_stages=True)
eng.timing_enable(4); eng.push(x); eng.sync()
_, _, ms = eng.timing_read()
clk = np.stack([eng.debug_header(128 + i).astype(np.float64) * 64 for i in range(4)])
tot = clk.sum(0).mean()
print("decoder %.2f ms; wave cycles per channel %.3g; per burst %.0f" % (float(ms[0]), tot, tot / 132))
for i, n in enumerate(["prologue", "view_ensure", "frame head", "payload/bptc"]):
    print("  %-14s %8.0f cycles/burst %5.1f %%" % (n, clk[i].mean() / 132, 100 * clk[i].mean() / tot))
