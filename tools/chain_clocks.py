#!/usr/bin/env python3
"""Wave-cycle budget of the DMR CHAIN kernel (slicer + decoder halves in one wavefront), -DDH_PHASE_CLOCKS build:
    tools/build_variant.sh phaseclk -DDH_PHASE_CLOCKS;  python tools/chain_clocks.py variants/lib_phaseclk.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from digiham_amd import _capi, api, synth_torch
lib = _capi.load(sys.argv[1]); ctx = api.Context(lib=lib)
B = 16384
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), "dmr", B, 132, seed=1000)
T = info["samples_per_channel"]
eng = api.Engine(B, T, ctx=ctx, proto="dmr")
eng.timing_enable(4); eng.push(x); eng.sync()
_, ms, _ = eng.timing_read()
sl = np.stack([eng.debug_header(20 + i).astype(np.float64) * 64 for i in range(8)])
de = np.stack([eng.debug_header(128 + i).astype(np.float64) * 64 for i in range(4)])
tot = sl.sum(0).mean() + de.sum(0).mean()
print("chain kernel %.2f ms; wave cycles per channel: slicer %.3g + decoder %.3g = %.3g (x 4 rounds / 2.0 GHz = %.2f ms)" % (float(ms[0]), sl.sum(0).mean(), de.sum(0).mean(), tot, 4 * tot / 2.0e6))
for i, n in enumerate(["P1 stage", "P2 FIR+wb", "P3 windows", "pf+P4 scan", "P5 slice", "P6 timing", "P7 commit", "pro/epilogue"]):
    print("  slicer %-14s %9.0f cycles/run   %5.1f %%" % (n, sl[i].mean() / (T / 1000.0), 100 * sl[i].mean() / tot))
for i, n in enumerate(["prologue", "view_ensure", "frame head", "payload/bptc"]):
    print("  decoder %-13s %9.0f cycles/burst %5.1f %%" % (n, de[i].mean() / 132, 100 * de[i].mean() / tot))
