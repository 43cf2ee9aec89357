#!/usr/bin/env python3
"""Float error of DH_FLAG_ONE_LAUNCH against the oracle for builds of the library: tools/one_launch_err.py lib.so ...
(max over 32 channels x 2 pushes of |y - ref| / max(|ref|, rms(ref)); configs[1] asks for 1e-6)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch, _capi
from oracle import oracle
U = 32
dev = torch.device("cuda", 0)
for seed in (555, 556):
    base, info = synth_torch.make_batch(torch, dev, "dmr", U, 33, U=U, seed=seed)
    T = info["samples_per_channel"]
    ref = oracle.chain(np.tile(base.cpu().numpy(), (1, 2)), proto=0, keep_filtered=True, threads=8)
    r = ref["filtered"]
    rms = np.sqrt(np.mean(r.astype(np.float64) ** 2)) + 1e-30
    for path in sys.argv[1:]:
        ctx = api.Context(lib=_capi.load(path))
        eng = api.Engine(U, T, proto="none", keep_filtered=True, one_launch=True, ctx=ctx)
        ys = []
        for _ in range(2):
            eng.push(base); ys.append(eng.filtered()[:, :T].copy())
        eng.close()
        yy = np.concatenate(ys, axis=1)
        err = np.abs(yy.astype(np.float64) - r) / np.maximum(np.abs(r), rms)
        print("seed %d %s: max %.3e  99.99%% %.3e  mean %.3e" % (seed, os.path.basename(path), err.max(), np.quantile(err, 0.9999), err.mean()), flush=True)
