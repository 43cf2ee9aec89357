#!/usr/bin/env python3
"""Soak of the tail split (engine.hip: k_chain): batches of > 8192 DISTINCT noisy channels, every channel's symbols,
frames and events of two consecutive pushes compared between launches split at random points (two and three parts) and
whole launches; 64 channels of every batch also against the oracle.   tools/soak_split.py [seeds] [channels]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from digiham_amd import api, synth_torch
from oracle import oracle as O

ctx = api.Context()
dev = torch.device("cuda", 0)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
base_B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192            # channels per batch (plus a random 1..199)
PROTOS = {"dmr": (50, {}, dict(proto=1)), "ysf": (15, {}, dict(proto=2)), "nxdn": (19, dict(rrc="narrow", sps=20), dict(proto=3, rrc=2, sps=20)),
          "dstar": (76, dict(rrc="none", demod="fsk"), dict(proto=5, rrc=0, levels=2))}
cases = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(9100 + seed)
    for proto, (units, kw, okw) in PROTOS.items():
        B = base_B + int(rng.integers(1, 200))
        x, info = synth_torch.make_batch(torch, dev, proto, B, units, seed=5000 + 17 * seed, sps=kw.get("sps", 10))
        T = info["samples_per_channel"]
        assert T >= 65536, (proto, T)
        a = int(rng.integers(5, 96)); b = int(rng.integers(a + 1, 100))
        digests = {}
        for pct in ("0", "80", str(a), "%d,%d" % (a, b)):
            os.environ["DH_TAIL_SPLIT"] = pct
            eng = api.Engine(B, T, proto=proto, ctx=ctx, **kw)
            h = [hashlib.sha256() for _ in range(B)]
            keep = []
            for push in range(2):
                eng.push(x)
                outs = (eng.symbols(), eng.frames(), eng.events())
                for rows, counts in outs:
                    raw = rows.view(np.uint8).reshape(B, -1); item = raw.shape[1] // rows.shape[1]
                    for ch in range(B):
                        h[ch].update(counts[ch].tobytes()); h[ch].update(raw[ch, :int(counts[ch]) * item].tobytes())
                if pct == "0":
                    keep.append([(rows[:64].copy(), counts[:64].copy()) for rows, counts in outs])
            eng.close()
            digests[pct] = [v.hexdigest() for v in h]
            if pct == "0":
                ref = O.chain(np.tile(x[:64].cpu().numpy(), (1, 2)), threads=8, **okw)
                for ch in range(64):
                    gs = np.concatenate([k[0][0][ch, :k[0][1][ch]] for k in keep]); gf = np.concatenate([k[1][0][ch, :k[1][1][ch]] for k in keep])
                    ge = np.concatenate([k[2][0][ch, :k[2][1][ch]] for k in keep])
                    assert len(gs) == ref["sym_count"][ch] and (gs == ref["syms"][ch, :len(gs)]).all(), (proto, seed, ch)
                    assert len(gf) == ref["out_count"][ch] and (gf == ref["out"][ch, :len(gf)]).all(), (proto, seed, ch)
                    assert ge.tobytes() == ref["events"][ch, :ref["event_count"][ch]].tobytes(), (proto, seed, ch)
            else:
                bad = [ch for ch in range(B) if digests[pct][ch] != digests["0"][ch]]
                assert not bad, (proto, seed, pct, bad[:8])
                cases += B
        frames = sum(1 for _ in digests["0"])
        print("seed %d %-5s %5d channels x %6d samples x 2 pushes: split at 80 / %d / %d,%d %% identical to whole launches; 64 channels = oracle" % (seed, proto, B, T, a, a, b), flush=True)
print("soak_split: %d channel-cases (channel x split setting, two pushes each) identical" % cases)
