#!/usr/bin/env python3
"""Push the bench workload through a given build of the library (for counter passes of diagnostic variants):
    tools/run_lib.py <dmr|ysf|...> <lib.so> [pushes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from digiham_amd import api, synth_torch, _capi
proto, path = sys.argv[1], sys.argv[2]
pushes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B = 16384
units = {"dmr": 132, "ysf": 40, "nxdn": 50, "dstar": 198, "pocsag": 148}[proto]
ekw = {"nxdn": dict(rrc="narrow", sps=20), "dstar": dict(rrc="none", demod="fsk", sps=10),
       "pocsag": dict(rrc="none", demod="fsk", sps=40, invert=True)}.get(proto, {})
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), proto, B, units, seed=1000, sps=ekw.get("sps", 10))
eng = api.Engine(B, info["samples_per_channel"], proto=proto, ctx=api.Context(lib=_capi.load(path)), **ekw)
for _ in range(pushes):
    eng.push(x)
eng.sync()
