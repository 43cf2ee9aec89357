"""the receiver front-end in front of rrc_filter (examples/dmr-decoder.sh:13-17:
`rtl_fm -M fm -s 48000 | csdr convert -i s16 -o float | csdr dcblock`).  rtl_fm and csdr are third-party tools without
source in the reference tree, so this stage is the project's OWN specification -- parity unpinned;
the product (dh_frontend_s16) is held bit-for-bit to the independent restatement of that specification in
oracle/frontend.c, alone and in front of the whole DMR chain."""
import numpy as np
import pytest

from common import make_channels
from digiham_amd import api, synth


def _fm_iq(x, seed, deviation=0.35, amplitude=12000.0, snr_db=30.0, cfo=0.01):
    """complex baseband of an FM transmitter driven by the 4FSK audio x: int16 I / Q interleaved [n][2]"""
    rng = np.random.default_rng(seed)
    phase = np.cumsum(np.pi * (deviation * x.astype(np.float64) + cfo))       # a carrier offset becomes DC in the audio
    z = amplitude * np.exp(1j * phase)
    z += (rng.normal(0, 1, len(z)) + 1j * rng.normal(0, 1, len(z))) * amplitude / np.sqrt(2) * 10 ** (-snr_db / 20)
    iq = np.stack([z.real, z.imag], axis=1)
    return np.clip(np.rint(iq), -32768, 32767).astype(np.int16)


def test_atan2_polynomial_matches_its_specification(oracle):
    """the discriminator's arctangent against libm: |error| <= 3e-8 turns (the A&S 4.4.49 bound / pi plus rounding)"""
    import ctypes as C
    L = oracle.lib()
    L.orc_fe_atan2_over_pi.restype = C.c_float
    L.orc_fe_atan2_over_pi.argtypes = [C.c_int32, C.c_int32]
    rng = np.random.default_rng(1)
    pts = np.concatenate([rng.integers(-2 ** 30, 2 ** 30, (4000, 2)), rng.integers(-40, 40, (2000, 2)),
                          [[0, 0], [0, 5], [0, -5], [7, 0], [-7, 0], [3, 3], [-3, 3], [3, -3], [-3, -3]]])
    got = np.array([L.orc_fe_atan2_over_pi(int(im), int(re)) for im, re in pts])
    want = np.arctan2(pts[:, 0].astype(np.float64), pts[:, 1].astype(np.float64)) / np.pi
    assert np.abs(got - want).max() < 1e-7


@pytest.mark.parametrize("mode", ["audio", "iq"])
def test_frontend_bit_exact_and_streamable(ctx, oracle, mode):
    x = make_channels("dmr", [5, 6, 7], 6)
    if mode == "audio":
        raw = np.clip(np.rint(x * 9000.0 + 700.0), -32768, 32767).astype(np.int16)       # int16 audio with a DC offset
    else:
        raw = np.stack([_fm_iq(row, 11 + i).reshape(-1) for i, row in enumerate(x)])
    per = 1 if mode == "audio" else 2
    n = raw.shape[1] // per
    for dc in (True, False):
        ref = np.stack([oracle.frontend(r, mode, dc)[0] for r in raw])
        got, st, pos = [], None, 0
        for c in (1000, 1, 4097, n):                                 # ragged chunks: the state carries x[n-1], y[n-1], z[n-1]
            c = min(c, n - pos)
            out, st = ctx.frontend(np.ascontiguousarray(raw[:, per * pos:per * (pos + c)]), mode, dc, st)
            got.append(ctx.mem.to_numpy(out).copy())
            pos += c
        assert np.concatenate(got, axis=1).tobytes() == ref.tobytes(), (mode, dc)
    # the DC blocker does its job: the offset of the input is gone from the output
    assert abs(float(np.stack([oracle.frontend(r, mode, True)[0] for r in raw])[:, 3000:].mean())) < 0.02


def test_frontend_iq_edge_values(ctx, oracle):
    """The discriminator on I / Q that hits every branch of the arctangent: zero vectors (either sample 0), products on the
    axes and the diagonals, both signs, full-scale values; odd and even tile boundaries (the GPU evaluates pairs of samples)."""
    rng = np.random.default_rng(77)
    n = 3 * 4096 + 37
    rows = []
    for kind in range(4):
        z = rng.integers(-32768, 32768, (n, 2))
        if kind == 1:
            z = rng.integers(-3, 4, (n, 2))                              # tiny values: zeros, axes and diagonals are common
        if kind == 2:
            z[rng.random(n) < 0.3] = 0
            rep = rng.random(n) < 0.3
            z[1:][rep[1:]] = z[:-1][rep[1:]]                             # repeated samples: Im d = 0, Re d > 0
            neg = rng.random(n) < 0.2
            z[1:][neg[1:]] = -z[:-1][neg[1:]]                            # opposite samples: Im d = 0, Re d < 0
        if kind == 3:
            z = rng.choice([-32768, 32767, 0, 1, -1], (n, 2))
        rows.append(np.clip(z, -32768, 32767).astype(np.int16).reshape(-1))
    raw = np.stack(rows)
    ref = np.stack([oracle.frontend(r, "iq", True)[0] for r in raw])
    got, st, pos = [], None, 0
    for c in (127, 1, 4096, 130, n):
        c = min(c, n - pos)
        out, st = ctx.frontend(np.ascontiguousarray(raw[:, 2 * pos:2 * (pos + c)]), "iq", True, st)
        got.append(ctx.mem.to_numpy(out).copy())
        pos += c
    assert np.concatenate(got, axis=1).tobytes() == ref.tobytes()


@pytest.mark.parametrize("mode", ["audio", "iq"])
def test_frontend_in_front_of_the_dmr_chain(ctx, oracle, mode):
    """IQ (or int16 audio) in, DMR frames out: front-end -> rrc(wide) -> gfsk(10) -> dmr_decoder on the engine equals the
    same pipe through the oracle, and the decoder finds the calls the generator put in."""
    x = make_channels("dmr", [21, 22], 14, impair=False)
    if mode == "audio":
        raw = np.clip(np.rint(x * 9000.0 - 1500.0), -32768, 32767).astype(np.int16)
    else:
        raw = np.stack([_fm_iq(row, 31 + i).reshape(-1) for i, row in enumerate(x)])
    audio, _ = ctx.frontend(raw, mode, True)
    eng = api.Engine(raw.shape[0], audio.shape[1], proto="dmr", ctx=ctx)
    eng.push(audio)
    s, sc = eng.symbols(); f, fc = eng.frames(); e, ec = eng.events()
    eng.close()
    ref_audio = np.stack([oracle.frontend(r, mode, True)[0] for r in raw])
    ref = oracle.chain(ref_audio, proto=1)
    for b in range(raw.shape[0]):
        assert sc[b] == ref["sym_count"][b] and (s[b, :sc[b]] == ref["syms"][b, :sc[b]]).all()
        assert fc[b] == ref["out_count"][b] and (f[b, :fc[b]] == ref["out"][b, :fc[b]]).all()
        assert e[b, :ec[b]].tobytes() == ref["events"][b, :ref["event_count"][b]].tobytes()
        assert fc[b] >= 27 * 6 and ec[b] > 0                     # voice bursts came out: the pipe decodes, not just agrees
