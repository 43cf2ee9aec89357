#!/usr/bin/env python3
"""A/B of build variants on the materialised-RRC path: tools/tile_ab.py lib_a.so lib_b.so ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch, _capi
B = 16384
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), "dmr", B, 132, seed=1000)
T = info["samples_per_channel"]
for path in sys.argv[1:] * 2:
    ctx = api.Context(lib=_capi.load(path))
    row = []
    for fast in (False, True):
        eng = api.Engine(B, T, proto="none", keep_filtered=True, fast_fir=fast, ctx=ctx)
        eng.timing_enable(8)
        for _ in range(2): eng.push(x)
        eng.sync(); eng.timing_read()
        for _ in range(5): eng.push(x)
        eng.sync()
        a, b, c = eng.timing_read()
        row.append("%s rrc %.2f slicer %.2f" % ("fma" if fast else "exact", a.mean(), b.mean()))
        eng.close()
    print(os.path.basename(path), " | ".join(row), flush=True)
