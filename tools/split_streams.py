#!/usr/bin/env python3
"""Experiment: one batch of channels as two engines on two HIP streams with unequal shares (the drain of one kernel's
last wavefronts is filled by the other stream's kernel):  tools/split_streams.py [proto] [share_a ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch
proto = sys.argv[1] if len(sys.argv) > 1 else "dmr"
shares = [tuple(float(v) for v in a.split(",")) for a in sys.argv[2:]] or [(1.0,), (0.75,), (0.8,), (0.85,), (0.9,), (0.6, 0.3), (0.5, 0.3), (0.55, 0.3, 0.1)]
B = int(os.environ.get("SS_B", "16384"))
dev = torch.device("cuda", 0)
x, info = synth_torch.make_batch(torch, dev, proto, B, 132 if proto == "dmr" else 40, seed=1000)
T = info["samples_per_channel"]
ctx = api.Context(device=0)
K = int(os.environ.get('SS_K', '10'))
for share in shares:
    cuts = [0]
    for v in share:
        cuts.append(min(B, cuts[-1] + int(B * v) // 64 * 64))
    if cuts[-1] < B:
        cuts.append(B)
    parts = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
    engs = []
    prio = os.environ.get('SS_PRIO')                  # "1": the first (largest) part on a high-priority stream, the rest low
    for i, (lo, hi) in enumerate(parts):
        st = torch.cuda.Stream(dev) if not prio else torch.cuda.Stream(dev, priority=(-1 if i == 0 else 0))
        with torch.cuda.stream(st):
            engs.append((api.Engine(hi - lo, T, proto=proto, ctx=ctx), x[lo:hi]))
    torch.cuda.synchronize()
    for _ in range(2):
        for e, xs in engs: e.push(xs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        for e, xs in engs: e.push(xs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    for e, _ in engs: e.sync(); e.close()
    print("%s %s: %.3f ms/step = %.0f channels" % (proto, "+".join(str(b - a) for a, b in parts), dt * 1e3, B * T / dt / 48000), flush=True)
