#!/usr/bin/env python3
"""Micro-benchmark of the FIR / slicer kernels for A/B work on build variants.

    python tools/fir_bench.py [lib.so ...]      # each lib: RRC-only tile kernel, slicer-only, fused; ms per launch
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from digiham_amd import _capi, api


def main(libs):
    B, T = int(os.environ.get("FB_B", 16384)), int(os.environ.get("FB_T", 190080))
    torch.manual_seed(0)
    x = (torch.randn((B, T), device="cuda") * 0.3).contiguous()
    for path in libs or [None]:
        lib = _capi.load(path) if path else _capi.load()
        ctx = api.Context(lib=lib)
        row = []
        for name, kw in (("rrc_tile", dict(rrc="wide", demod="none", proto="none", keep_filtered=True)),
                         ("rrc_tile_fast", dict(rrc="wide", demod="none", proto="none", keep_filtered=True, fast_fir=True)),
                         ("slicer_only", dict(rrc="none", demod="gfsk", proto="none")),
                         ("fused", dict(rrc="wide", demod="gfsk", proto="none")),
                         ("fused_fast", dict(rrc="wide", demod="gfsk", proto="none", fast_fir=True))):
            eng = api.Engine(B, T, ctx=ctx, **kw)
            eng.timing_enable(8)
            for _ in range(2):
                eng.push(x)
            eng.sync(); eng.timing_read()
            for _ in range(4):
                eng.push(x)
            a, b, c = eng.timing_read()
            row.append("%s %.2f" % (name, float(np.mean(a) + np.mean(b))))
            eng.close()
        print(os.path.basename(path or "default"), "B=%d T=%d" % (B, T), " | ".join(row), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
