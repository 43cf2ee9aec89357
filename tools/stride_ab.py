#!/usr/bin/env python3
"""Does the row stride of the input matter?  tools/stride_ab.py [proto]: the bench workload pushed from rows padded by 0, 16, 64, ... floats
(16 384 channels stream from addresses one row stride apart: a stride that maps them onto few HBM channels would show here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from digiham_amd import api, synth_torch
proto = sys.argv[1] if len(sys.argv) > 1 else "dmr"
B = 16384
units = {"dmr": 132, "ysf": 40}[proto]
dev = torch.device("cuda", 0)
x, info = synth_torch.make_batch(torch, dev, proto, B, units, seed=1000, sps=10)
T = info["samples_per_channel"]
for pad in (0, 16, 64, 128, 256, 448, 1024, 2048 + 64, 0):
    buf = torch.zeros((B, T + pad), dtype=torch.float32, device=dev)
    buf[:, :T] = x
    eng = api.Engine(B, T, proto=proto, split_stages=False)
    eng.timing_enable(8)
    for _ in range(2): eng.push(buf, n=T)
    eng.sync(); eng.timing_read()
    for _ in range(5): eng.push(buf, n=T)
    eng.sync()
    a, b, c = eng.timing_read()
    print("row stride %d floats (%d B, pad %d): chain %.3f ms" % (T + pad, 4 * (T + pad), pad, b.mean()), flush=True)
    eng.close(); del buf
