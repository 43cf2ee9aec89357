#!/usr/bin/env python3
"""How often does the error-bounded FIR need the reference's arithmetic on the bench workload?
    python tools/bound_stats.py [dmr|ysf] [channels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch
proto = sys.argv[1] if len(sys.argv) > 1 else "dmr"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ctx = api.Context()
ekw = dict(rrc="narrow", sps=20) if proto == "nxdn" else {}
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), proto, B, {"dmr": 132, "nxdn": 50}.get(proto, 40), seed=1000, sps=ekw.get("sps", 10))
eng = api.Engine(B, info["samples_per_channel"], ctx=ctx, proto=proto, **ekw)
for _ in range(3):
    eng.push(x)
eng.sync()
unc, ex, nsym = eng.debug_header(16).astype(np.int64), eng.debug_header(17).astype(np.int64), eng.debug_header(3).astype(np.int64)
blocks, ordered = eng.timing_stats()
runs = 3 * info["samples_per_channel"] / 1000.0
exb = eng.debug_header(18).astype(np.int64)
print("%s: symbols %d, decided exactly %d (%.4f %%), worst channel %.3f %%" % (proto, nsym.sum(), unc.sum(), 100.0 * unc.sum() / nsym.sum(), 100.0 * (unc / np.maximum(nsym, 1)).max()))
print("runs through the exact FIR: %.3f %% (mean %.1f of ~%.0f per channel)" % (100.0 * ex.mean() / runs, ex.mean(), runs))
print("timing blocks %d, ordered chain %d (%.4f %%), ring recomputed exactly %d (%.4f %%)" % (blocks.sum(), ordered.sum(), 100.0 * ordered.sum() / max(blocks.sum(), 1), exb.sum(), 100.0 * exb.sum() / max(blocks.sum(), 1)))
by_class = [(unc[c::3].sum() / max(nsym[c::3].sum(), 1)) for c in range(3)]
print("share decided exactly by noise class (clean, 20 dB, 12 dB):", ["%.5f" % v for v in by_class])
