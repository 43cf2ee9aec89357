#!/usr/bin/env python3
"""A/B of build variants on the full DMR (or YSF) chain: tools/lib_ab.py dmr variants/lib_a.so variants/lib_b.so ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import api, synth_torch, _capi
proto = sys.argv[1]
B = int(os.environ.get("DH_AB_CHANNELS", "16384"))
units = {"dmr": 132, "ysf": 40, "nxdn": 50, "dstar": 198, "pocsag": 148}[proto]
ekw = {"nxdn": dict(rrc="narrow", sps=20), "dstar": dict(rrc="none", demod="fsk", sps=10),
       "pocsag": dict(rrc="none", demod="fsk", sps=40, invert=True)}.get(proto, {})
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), proto, B, units, seed=1000, sps=ekw.get("sps", 10))
T = info["samples_per_channel"]
for path in sys.argv[2:] * 2:
    ctx = api.Context(lib=_capi.load(path))
    row = []
    for split in (True, False):
        try:
            eng = api.Engine(B, T, proto=proto, split_stages=split, ctx=ctx, **ekw)
        except TypeError:
            continue
        eng.timing_enable(8)
        for _ in range(2): eng.push(x)
        eng.sync(); eng.timing_read()
        for _ in range(5): eng.push(x)
        eng.sync()
        a, b, c = eng.timing_read()
        tag = ""
        if hasattr(eng, "read_rows") and not split:          # what came out, for 128 channels spread over the batch: variants must agree
            import hashlib
            pick = list(range(0, B, B // 128))
            h = hashlib.sha256()
            for what in ("symbols", "frames"):
                rows, cnt = eng.read_rows(what, pick)
                h.update(cnt.tobytes())
                for j in range(len(pick)):
                    h.update(rows[j, :cnt[j]].tobytes())
            tag = " out " + h.hexdigest()[:12]
        row.append("%s slicer %.3f decoder %.3f%s" % ("split" if split else "chain", b.mean(), c.mean(), tag))
        eng.close()
    print(os.path.basename(path), " | ".join(row), flush=True)
