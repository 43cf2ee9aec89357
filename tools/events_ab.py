#!/usr/bin/env python3
"""Cost of the decoder event stream in the chain kernel: tools/events_ab.py [proto]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch
proto = sys.argv[1] if len(sys.argv) > 1 else "dmr"
B = 16384
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), proto, B, 132 if proto == "dmr" else 40, seed=1000)
ctx = api.Context(device=0)
for rep in range(2):
    for events in (True, False):
        eng = api.Engine(B, info["samples_per_channel"], proto=proto, events=events, ctx=ctx)
        eng.timing_enable(8)
        for _ in range(2): eng.push(x)
        eng.sync(); eng.timing_read()
        for _ in range(5): eng.push(x)
        eng.sync()
        _, b, _ = eng.timing_read()
        print(proto, "events" if events else "no events", "%.2f ms" % b.mean(), flush=True)
        eng.close()
