#!/usr/bin/env python3
"""Step time when every channel is noise (decoders stay in their sync search): tools/idle_channels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api
B, T = 16384, 190080
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
x = torch.randn((B, T), device=dev, generator=g, dtype=torch.float32) * 0.2
ctx = api.Context(device=0)
for proto, kw in (("dmr", {}), ("ysf", {}), ("dstar", dict(rrc="none", demod="fsk")), ("nxdn", dict(rrc="narrow", sps=20)), ("pocsag", dict(rrc="none", demod="fsk", sps=40, invert=True))):
    eng = api.Engine(B, T, proto=proto, ctx=ctx, **kw)
    eng.timing_enable(8)
    for _ in range(2): eng.push(x)
    eng.sync(); eng.timing_read()
    for _ in range(4): eng.push(x)
    eng.sync()
    a, b, c = eng.timing_read()
    print("%-7s noise only: rrc %.2f slicer/chain %.2f decoder %.2f ms; output bytes %d" % (proto, a.mean(), b.mean(), c.mean(), int(eng.frames()[1].sum())), flush=True)
    eng.close()

# how often the timing recovery takes its ordered (exact-order) fallback on noise vs. on a clean signal
from digiham_amd import synth_torch
for name, sig in (("noise", x[:4096]), ("dmr", synth_torch.make_batch(torch, dev, "dmr", 4096, 132, seed=1000)[0])):
    eng = api.Engine(4096, T, proto="none", ctx=ctx)
    eng.push(sig); eng.sync()
    blocks, ordered = eng.timing_stats()
    print("%s: %d timing blocks, %d ordered fallbacks (%.2f %%)" % (name, int(blocks.sum()), int(ordered.sum()), 100.0 * ordered.sum() / max(blocks.sum(), 1)))
    eng.close()
