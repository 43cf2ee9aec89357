import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from digiham_amd import api, synth_torch
B=4096
x, info = synth_torch.make_batch(torch, torch.device("cuda",0), "dmr", B, 132, seed=1000)
eng = api.Engine(B, info["samples_per_channel"], proto="dmr")
eng.push(x); eng.sync()
bl, od = eng.timing_stats()
print("blocks", bl.sum(), "ordered", od.sum(), "frac", od.sum()/bl.sum(), "channels with ordered", (od>0).sum())
print(np.bincount(od)[:10])
