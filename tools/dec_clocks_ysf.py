#!/usr/bin/env python3
"""Shader-clock breakdown of the YSF decoder kernel (diagnostic build: tools/build_variant.sh phaseclk -DDH_PHASE_CLOCKS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import _capi, api, synth_torch
lib = _capi.load(sys.argv[1]); ctx = api.Context(lib=lib)
B = 4096
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), "ysf", B, 8, seed=1000)
eng = api.Engine(B, info["samples_per_channel"], ctx=ctx, proto="ysf", split_stages=True)
eng.timing_enable(4); eng.push(x); eng.sync()
_, _, ms = eng.timing_read()
w = np.stack([eng.debug_header(128 + i) for i in range(4)])
clk = np.concatenate([[(w[i] & 0xFFFF).astype(np.float64) * 64, (w[i] >> 16).astype(np.float64) * 64] for i in range(4)])
tot = clk.sum(0).mean()
print("decoder %.2f ms; wave cycles per channel %.3g" % (float(ms[0]), tot))
for i, n in enumerate(["loop top", "view+planes+sync", "gather", "viterbi", "fich golay/crc", "payload/dch/header", "-", "-"]):
    print("  %-20s %9.0f cycles  %5.1f %%" % (n, clk[i].mean(), 100 * clk[i].mean() / tot))
