"""Randomised parity soak: many small cases with drawn impairments and push patterns through the chain kernels, compared
bit for bit with the oracle (dibits, frame bytes, events).  Aimed at the error-bounded FIR (dsp_core.hpp): amplitudes from
1e-6 to 1e4 (the error radius scales with max |x|), DC offsets larger than the signal, signal-to-noise ratios from -3 dB
up, fades to silence and back, pushes from one sample to the whole stream -- a wrong bound or a lost history sample would
show up as a flipped dibit somewhere in here.  The case list is fixed by the seeds below (failures are reproducible)."""
import numpy as np
import pytest

from common import assert_matches_oracle, run_engine
from digiham_amd import _taps, synth


def _case(rng, proto):
    seed = int(rng.integers(1, 10 ** 6))
    kw, okw, sps, taps = {}, {}, 10, None
    if proto == "dmr":
        s = synth.dmr_stream(seed, int(rng.integers(6, 14)), two_slots=bool(rng.integers(0, 2)))
        okw = dict(proto=1)
    elif proto == "ysf":
        s = synth.ysf_stream(seed, int(rng.integers(3, 6)), mode=["vd2", "vd1", "fr", "datafr"][int(rng.integers(0, 4))])
        okw = dict(proto=2)
    else:
        s = synth.nxdn_stream(seed, int(rng.integers(6, 12)))
        kw, okw, sps, taps = dict(rrc="narrow", sps=20), dict(rrc=2, sps=20, proto=3), 20, _taps.narrow()
    x = synth.shape(s, sps=sps, taps=taps) if taps is not None else synth.shape(s)
    gain = float(10.0 ** rng.uniform(-6, 4)) if rng.random() < 0.5 else float(rng.uniform(0.2, 3.0))
    snr = None if rng.random() < 0.2 else float(rng.uniform(-3, 40))
    x = synth.impair(x, seed, snr_db=snr, dc=float(rng.uniform(-2, 2)) * (1.0 if rng.random() < 0.3 else 0.05),
                     gain=1.0, delay=int(rng.integers(0, 40)))
    if rng.random() < 0.3:                                   # a fade to (near) silence and back
        a, b = sorted(rng.integers(0, len(x), 2))
        x[a:b] *= np.float32(rng.choice([0.0, 1e-4, 0.02]))
    x = (x * np.float32(gain)).astype(np.float32)
    n = len(x)
    kind = int(rng.integers(0, 4))
    chunks = ([n], [int(rng.integers(1, 50)), int(rng.integers(2000, 9000)), int(rng.integers(1, 400))],
              [int(rng.integers(900, 1100))], [int(rng.integers(100, 30000)) for _ in range(6)])[kind]
    return x, kw, okw, chunks, dict(seed=seed, gain=gain, snr=snr, chunks=chunks[:3])


@pytest.mark.parametrize("proto,batch", [("dmr", 0), ("dmr", 1), ("ysf", 0), ("ysf", 1), ("nxdn", 0)])
def test_randomised_cases_match_the_oracle(ctx, oracle, proto, batch):
    rng = np.random.default_rng(20260929 + 17 * batch + {"dmr": 0, "ysf": 1000, "nxdn": 2000}[proto])
    flips = 0
    for c in range(5 if proto != "nxdn" else 3):
        x, kw, okw, chunks, what = _case(rng, proto)
        ref = oracle.chain(x[None, :], **okw)
        res = run_engine(ctx, x[None, :], proto, chunks, **kw)
        assert_matches_oracle(res, ref, 1, "%s case %d %r" % (proto, c, what))
        flips += int(ref["sym_count"][0])
    assert flips > 0
