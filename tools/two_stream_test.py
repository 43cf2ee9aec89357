#!/usr/bin/env python3
"""Experiment: one engine of 16384 channels vs two engines of 8192 on two streams (kernel overlap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch

dev = torch.device("cuda", 0)
B = 16384
x, info = synth_torch.make_batch(torch, dev, "dmr", B, 132)
T = info["samples_per_channel"]
ctx = api.Context()

def run(nsplit, steps=6):
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    engs = []
    per = B // nsplit
    for i, s in enumerate(streams):
        with torch.cuda.stream(s):
            engs.append(api.Engine(per, T, ctx=ctx, proto="dmr"))
    parts = [x[i * per:(i + 1) * per] for i in range(nsplit)]
    for w in range(2):
        for e, s, p in zip(engs, streams, parts):
            with torch.cuda.stream(s):
                e.push(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        for e, s, p in zip(engs, streams, parts):
            with torch.cuda.stream(s):
                e.push(p)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for e in engs: e.close()
    return dt * 1e3

for n in (1, 2, 4):
    print("engines/streams", n, "ms/step %.2f" % run(n), flush=True)
