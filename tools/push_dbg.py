import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from digiham_amd import api, synth_torch
B = 16384
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), "dmr", B, 132, seed=1000, sps=10)
T = info["samples_per_channel"]
ctx = api.Context()
for parts in (1, 40):
    n = T // parts
    for split in (False, True):
        eng = api.Engine(B, n, proto="dmr", ctx=ctx, split_stages=split)
        views = [x[:, i * n:(i + 1) * n] for i in range(parts)]
        for v in views[:max(1, parts // 4)]: eng.push(v)
        eng.sync()
        eng.timing_enable(parts)
        t0 = time.time()
        for v in views: eng.push(v)
        t1 = time.time()
        eng.sync()
        t2 = time.time()
        a, b, c = eng.timing_read()
        print("parts %d split %d: host enqueue %.2f ms, total %.2f ms; GPU per push: a %.3f b %.3f c %.3f (sum over pushes a %.2f b %.2f c %.2f)" % (parts, split, 1e3*(t1-t0), 1e3*(t2-t0), a.mean(), b.mean(), c.mean(), a.sum(), b.sum(), c.sum()), flush=True)
        eng.close()
