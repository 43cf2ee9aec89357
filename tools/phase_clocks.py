#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the slicer kernel (diagnostic build):

    tools/build_variant.sh phaseclk -DDH_PHASE_CLOCKS
    python tools/phase_clocks.py variants/lib_phaseclk.so [rrc] [sps] [demod]

Prints, per phase, the mean wavefront cycles per 100-symbol run and the share of the total."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from digiham_amd import _capi, api, synth_torch

NAMES = ["P1 stage", "P2 FIR+wb", "P3 windows", "pf+P4 scan", "P5 slice", "P6 timing", "P7 commit", "pro/epilogue"]


def main():
    lib = _capi.load(sys.argv[1])
    rrc = sys.argv[2] if len(sys.argv) > 2 else "wide"
    sps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    demod = sys.argv[4] if len(sys.argv) > 4 else "gfsk"
    ctx = api.Context(lib=lib)
    B = 16384
    proto, units = {10: ("dmr", 132), 20: ("nxdn", 50), 40: ("pocsag", 148)}[sps]
    x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), proto, B, units, seed=1000, sps=sps)
    T = info["samples_per_channel"]
    eng = api.Engine(B, T, ctx=ctx, rrc=rrc, demod=demod, sps=sps, invert=sps == 40, proto="none")
    eng.timing_enable(4)
    eng.push(x); eng.sync()
    _, ms, _ = eng.timing_read()
    clk = np.stack([eng.debug_header(20 + i).astype(np.float64) * 64 for i in range(8)])     # [phase][channel]
    runs = T / (100.0 * sps)
    tot = clk.sum(0).mean()
    print("kernel %.2f ms; wave cycles per channel %.3g; per run %.0f" % (float(ms[0]), tot, tot / runs))
    for i, n in enumerate(NAMES):
        print("  %-14s %8.0f cycles/run  %5.1f %%" % (n, clk[i].mean() / runs, 100 * clk[i].mean() / tot))


if __name__ == "__main__":
    main()
