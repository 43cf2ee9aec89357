#!/usr/bin/env python3
"""Cost of small pushes: the DMR chain over the same 190 080 samples per channel, handed over in 1 / 10 / 40 / 120 pushes.
    python tools/push_size.py [lib.so ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from digiham_amd import api, synth_torch, _capi
B = 16384
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), "dmr", B, 132, seed=1000, sps=10)
T = info["samples_per_channel"]
for path in (sys.argv[1:] or [None]):
    ctx = api.Context(lib=_capi.load(path)) if path else api.Context()
    for parts in (1, 10, 40, 120):
        n = T // parts
        eng = api.Engine(B, n, proto="dmr", ctx=ctx)
        views = [x[:, i * n:(i + 1) * n] for i in range(parts)]
        for v in views[:max(1, parts // 4)]: eng.push(v)
        eng.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record(torch.cuda.current_stream())
        for _ in range(reps):
            for v in views: eng.push(v)
        eng.sync()
        e1.record(torch.cuda.current_stream()); e1.synchronize()
        print("%s pushes of %6d samples: %.2f ms per %d samples" % (os.path.basename(path or "product"), n, e0.elapsed_time(e1) / reps, n * parts), flush=True)
        eng.close()
