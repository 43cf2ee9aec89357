#!/usr/bin/env python3
"""Shader-clock breakdown of the D-Star decoder kernel (diagnostic build: tools/build_variant.sh clk -DDH_PHASE_CLOCKS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import _capi, api, synth_torch
lib = _capi.load(sys.argv[1]); ctx = api.Context(lib=lib)
B = 16384
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), "dstar", B, 198, seed=1000)
eng = api.Engine(B, info["samples_per_channel"], ctx=ctx, rrc="none", demod="fsk", sps=10, proto="dstar")
eng.timing_enable(4); eng.push(x); eng.sync()
_, sl, ms = eng.timing_read()
w = np.stack([eng.debug_header(128 + i) for i in range(4)])
clk = np.stack([(w[i // 2] >> (16 * (i % 2))) & 0xFFFF for i in range(8)]).astype(np.float64) * 64
tot = clk.sum(0).mean()
print("slicer %.2f ms decoder %.2f ms; wave cycles per channel %.3g" % (float(sl[0]), float(ms[0]), tot))
for i, n in enumerate(["prologue", "sync search", "header parse", "take128", "voice out + end check", "slow data / sync", "epilogue", "-"]):
    print("  %-22s %9.0f cycles %5.1f %%" % (n, clk[i].mean(), 100 * clk[i].mean() / tot))
