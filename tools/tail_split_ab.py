#!/usr/bin/env python3
"""Tail split of the chain launches (HipBackend::go_chain): the same library, DH_TAIL_SPLIT = share of a push (percent) the
first workgroup of a channel takes, 0 = one workgroup per channel.  tools/tail_split_ab.py [proto] [pct ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch
proto = sys.argv[1] if len(sys.argv) > 1 else "dmr"
pcts = sys.argv[2:] or ["0", "80", "70,92", "0", "80"]
B = 16384
units = {"dmr": 132, "ysf": 40, "nxdn": 50, "dstar": 198, "pocsag": 148}[proto]
ekw = {"nxdn": dict(rrc="narrow", sps=20), "dstar": dict(rrc="none", demod="fsk", sps=10),
       "pocsag": dict(rrc="none", demod="fsk", sps=40, invert=True)}.get(proto, {})
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), proto, B, units, seed=1000, sps=ekw.get("sps", 10))
T = info["samples_per_channel"]
ref = None
for pct in pcts:
    os.environ["DH_TAIL_SPLIT"] = str(pct)
    eng = api.Engine(B, T, proto=proto, **ekw)
    eng.timing_enable(8)
    for _ in range(2): eng.push(x)
    eng.sync(); eng.timing_read()
    for _ in range(5): eng.push(x)
    eng.sync()
    a, b, c = eng.timing_read()
    import hashlib
    h = hashlib.sha256()
    for rows, counts in (eng.symbols(), eng.frames(), eng.events()):
        h.update(counts.tobytes())
        m = np.arange(rows.shape[1])[None, :] < counts[:, None]
        h.update(np.where(m if rows.dtype.fields is None else m, rows, np.zeros((), rows.dtype)).tobytes())
    dig = h.hexdigest()[:16]
    if ref is None: ref = dig
    print("%s DH_TAIL_SPLIT=%-6s chain %.3f ms  outputs %s %s" % (proto, pct, float(np.mean(b) + np.mean(c)), dig, "same" if dig == ref else "DIFFERENT"), flush=True)
    eng.close()
