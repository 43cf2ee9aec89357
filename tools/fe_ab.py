#!/usr/bin/env python3
"""A/B of front-end builds: tools/fe_ab.py lib...  (16 384 channels x 190 080 int16 I/Q pairs, ms per dh_frontend_s16 call)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from digiham_amd import api, _capi
B, T = 16384, 190080
iq = (torch.randn((B, 2 * T), device="cuda") * 8000).to(torch.int16)
out = torch.empty((B, T), dtype=torch.float32, device="cuda")
for path in sys.argv[1:] * 2:
    ctx = api.Context(lib=_capi.load(path)); mem = ctx.mem
    st = torch.zeros((B, 4), dtype=torch.float32, device="cuda")
    for mode in (2, 1):
        for _ in range(2):
            ctx.lib.dh_frontend_s16(mem.ptr(iq), 2 * T, mem.ptr(out), T, mem.ptr(st), B, T, mode, 1, mem.stream())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ctx.lib.dh_frontend_s16(mem.ptr(iq), 2 * T, mem.ptr(out), T, mem.ptr(st), B, T, mode, 1, mem.stream())
        e1.record(); e1.synchronize()
        print(os.path.basename(path), "iq" if mode == 2 else "audio", "%.2f ms" % (e0.elapsed_time(e1) / 4), flush=True)
