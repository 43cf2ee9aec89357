#!/usr/bin/env python3
"""When did each channel's wavefront run?  (diagnostic build: tools/build_variant.sh timeline -DDH_WAVE_TIMELINE)

    python tools/wave_timeline.py variants/lib_timeline.so [channels]

Prints the number of resident wavefronts over the launch (start / end stamps of every channel, 100 MHz wall clock) --
how long the machine runs below its 4 096 wavefront slots at the start and at the end of a launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from digiham_amd import _capi, api, synth_torch

lib = _capi.load(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
ctx = api.Context(lib=lib)
x, info = synth_torch.make_batch(torch, torch.device("cuda", 0), "dmr", B, 132, seed=1000)
eng = api.Engine(B, info["samples_per_channel"], ctx=ctx, proto="dmr")
eng.timing_enable(4)
for _ in range(3):
    eng.push(x)
eng.sync()
_, ms, _ = eng.timing_read()
t0 = eng.debug_header(30).astype(np.int64)
t1 = eng.debug_header(31).astype(np.int64)
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0          # microseconds
dur = t1 - t0
print("kernel %.2f ms; first start 0, last end %.0f us; wave duration mean %.0f us (min %.0f, max %.0f)" % (ms[-1], t1.max(), dur.mean(), dur.min(), dur.max()))
edges = np.linspace(0, t1.max(), 41)
for a, b in zip(edges[:-1], edges[1:]):
    mid = (a + b) / 2
    n = int(((t0 <= mid) & (t1 > mid)).sum())
    print("  %6.0f us  %5d resident  %s" % (mid, n, "#" * (n // 128)))
order = np.argsort(t0)
rounds = [dur[order[i:i + 4096]].mean() for i in range(0, B, 4096)]
print("mean duration by start order (groups of 4096):", ["%.0f" % r for r in rounds])

hw = eng.debug_header(28); xcc = eng.debug_header(29) & 15
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; slot = hw & 15
key = (((xcc.astype(np.int64) * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
print("distinct SIMDs seen: %d, wave slots used: %s" % (len(np.unique(key)), sorted(set(slot.tolist()))))
# per SIMD: how many waves ran there, and the gaps between one wave's end and the next wave's start in the same slot
counts = np.bincount(np.unique(key, return_inverse=True)[1])
print("waves per SIMD: min %d mean %.1f max %d" % (counts.min(), counts.mean(), counts.max()))
gaps = []
k2 = key * 16 + slot
for k in np.unique(k2)[:4096]:
    sel = np.nonzero(k2 == k)[0]
    o = sel[np.argsort(t0[sel])]
    gaps += list(t0[o[1:]] - t1[o[:-1]])
gaps = np.array(gaps)
print("gap between consecutive waves in one wave slot (us): n %d median %.1f mean %.1f p90 %.1f max %.1f" % (len(gaps), np.median(gaps), gaps.mean(), np.percentile(gaps, 90), gaps.max()))
# resident waves per SIMD over time
for tq in (500, 2000, 4000, 6000, 8000, 10000):
    res = np.bincount(np.unique(key, return_inverse=True)[1][(t0 <= tq) & (t1 > tq)], minlength=len(counts))
    print("  t=%5d us: waves per SIMD histogram %s" % (tq, np.bincount(res, minlength=6)[:6].tolist()))
