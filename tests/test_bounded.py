"""The error-bounded FIR of the wide-filter sps-10 kernels (dsp_core.hpp: DH_BOUNDED_FIR): FMA filtering + a proven
error radius decide what they can, the reference's arithmetic decides the rest.  Whatever the split, dibits, frames and
events must be the oracle's, bit for bit:

* normal mode, DH_FLAG_EXACT_SYMBOLS (every symbol through the exact evaluation: its bookkeeping of ring-slot positions,
  history across pushes, candidate search) and DH_FLAG_EXACT_FIR (every run through the rounded-product FIR);
* ragged and tiny pushes (the raw-sample history lives in the carried tail);
* inputs outside the bound's assumptions: silence, constants, denormal-sized and huge samples, NaN and infinity;
* signals built to sit ON the decision thresholds, where the bound cannot decide.
"""
import numpy as np
import pytest

from common import make_channels, run_engine, assert_matches_oracle
from digiham_amd import api, synth


def _stats(eng):
    return eng.debug_header(16), eng.debug_header(17)


def _run(ctx, x, proto, chunks, **kw):
    """like common.run_engine, plus the (uncertain symbols, exact runs) counters"""
    B, n = x.shape
    eng = api.Engine(B, max(chunks), proto=proto, ctx=ctx, **kw)
    syms, frames, evs = [[] for _ in range(B)], [[] for _ in range(B)], [[] for _ in range(B)]
    pos = i = 0
    while pos < n:
        c = min(chunks[i % len(chunks)], n - pos); i += 1
        eng.push(np.ascontiguousarray(x[:, pos:pos + c])); pos += c
        s, sc = eng.symbols()
        for b in range(B):
            syms[b].append(s[b, :sc[b]].copy())
        if eng.has_proto:
            f, fc = eng.frames(); e, ec = eng.events()
            for b in range(B):
                frames[b].append(f[b, :fc[b]].copy()); evs[b].append(e[b, :ec[b]].copy())
    st = _stats(eng)
    eng.close()
    cat = lambda parts, dt: [np.concatenate(p) if p else np.zeros(0, dt) for p in parts]
    return {"syms": cat(syms, np.uint8), "frames": cat(frames, np.uint8), "events": cat(evs, api.EVENT_DTYPE), "filtered": None}, st


@pytest.mark.parametrize("proto", ["dmr", "ysf"])
def test_all_three_modes_match_the_oracle(ctx, oracle, proto):
    x = make_channels(proto, [31, 32, 33, 34, 35], 14 if proto == "dmr" else 5)
    x[4] = np.random.default_rng(3).normal(0, 0.3, x.shape[1]).astype(np.float32)        # a noise-only channel
    ref = oracle.chain(x, proto=1 if proto == "dmr" else 2)
    total = int(ref["sym_count"].sum())
    for chunks in ([x.shape[1]], [4000, 37, 1, 2500, 999]):
        res, (unc, ex) = _run(ctx, x, proto, chunks)
        assert_matches_oracle(res, ref, x.shape[0], "normal %s" % chunks[:1])
        assert unc.sum() < total // 20                      # the bound decides nearly everything by itself
        res, (unc, ex) = _run(ctx, x, proto, chunks, exact_symbols=True)
        assert_matches_oracle(res, ref, x.shape[0], "exact symbols %s" % chunks[:1])
        assert int(unc.sum()) == total                      # every symbol went through the exact evaluation
        res, (unc, ex) = _run(ctx, x, proto, chunks, exact_fir=True)
        assert_matches_oracle(res, ref, x.shape[0], "exact fir %s" % chunks[:1])
        assert ex.min() > 0


def test_tiny_pushes_keep_the_history(ctx, oracle):
    """pushes far shorter than the 1 152-sample history, every symbol evaluated exactly: ring slots of earlier pushes are
    recomputed from the carried tail"""
    x = make_channels("dmr", [41, 42], 6)
    ref = oracle.chain(x, proto=1)
    for chunks in ([97], [1, 250, 13]):
        res, _ = _run(ctx, x, "dmr", chunks, exact_symbols=True)
        assert_matches_oracle(res, ref, x.shape[0], "tiny %s" % chunks)
        res, _ = _run(ctx, x, "dmr", chunks)
        assert_matches_oracle(res, ref, x.shape[0], "tiny normal %s" % chunks)


def test_samples_outside_the_bound(ctx, oracle):
    base = make_channels("dmr", [51], 8)[0]
    n = len(base)
    rng = np.random.default_rng(9)
    rows = [np.zeros(n, np.float32),                                        # digital silence
            np.full(n, 0.37, np.float32),                                   # a constant
            base * np.float32(1e-30), base * np.float32(1e-38),             # below the bound's range / denormals
            base * np.float32(1e20), base * np.float32(3e37),               # above it / overflowing accumulators
            base.copy(), base.copy(), base.copy(),
            np.where(rng.random(n) < 0.5, base, 0.0).astype(np.float32)]    # stretches of zeros inside a signal
    rows[6][5000] = np.nan; rows[6][20000:20003] = np.nan
    rows[7][7000] = np.inf; rows[7][9000] = -np.inf
    rows[8][11000:13000] *= np.float32(1e22)                                # a burst of huge samples, then normal again
    x = np.stack(rows)
    with np.errstate(all="ignore"):
        ref = oracle.chain(x, proto=1)
    for chunks in ([n], [3000, 711]):
        for kw in ({}, {"exact_symbols": True}):
            res, _ = _run(ctx, x, "dmr", chunks, **kw)
            assert_matches_oracle(res, ref, x.shape[0], "odd samples %s %s" % (chunks[:1], kw))


def test_signals_on_the_thresholds(ctx, oracle):
    """Symbols whose mid-symbol average sits within a few float ulps of an AGC threshold: for single symbols of a clean
    4-level signal the level at which the ORACLE's dibit flips is found by bisection (to 1e-10), and the signal is then
    sent with that symbol a hair to either side of the flip.  The bound cannot decide these; the exact evaluation must."""
    rng = np.random.default_rng(17)
    g = synth.wide_rrc_taps().astype(np.float64)
    g = g / g.sum() * 10

    def wave(lv):
        imp = np.zeros(len(lv) * 10); imp[::10] = lv
        return (np.convolve(imp, g)[:len(imp)] * 0.5).astype(np.float32)

    def dibits(lv):
        r = oracle.chain(wave(lv)[None, :], proto=1)
        return r["syms"][0, :r["sym_count"][0]].copy()

    rows = []
    for ch in range(4):
        s = rng.integers(0, 4, 420).astype(np.uint8)
        lv = synth.LEVELS[s].astype(np.float64)
        for k in range(200 + 7 * ch, 400, 23):                           # a dozen symbols per channel, one threshold each
            lo_lv, hi_lv = [(-1.0, -1 / 3), (-1 / 3, 1 / 3), (1 / 3, 1.0)][(k // 23) % 3]
            a, b = lv.copy(), lv.copy(); a[k], b[k] = lo_lv, hi_lv
            da, db = dibits(a), dibits(b)
            if len(da) != len(db) or (da != db).sum() != 1:
                continue                                                 # the change leaked into a neighbour: skip this one
            idx = int(np.nonzero(da != db)[0][0])
            x0, x1 = lo_lv, hi_lv
            for _ in range(40):
                mid = 0.5 * (x0 + x1)
                t = lv.copy(); t[k] = mid
                d = dibits(t)
                if len(d) == len(da) and d[idx] == da[idx]: x0 = mid
                else: x1 = mid
            lv[k] = x0 if (k // 23) % 2 else x1                          # just below / just above the flip
        rows.append(wave(lv))
    x = np.stack(rows)
    ref = oracle.chain(x, proto=1)
    res, (unc, ex) = _run(ctx, x, "dmr", [x.shape[1]])
    assert_matches_oracle(res, ref, x.shape[0], "thresholds")
    assert unc.sum() >= 8, "too few symbols needed the exact evaluation (%s): the bisection did not land on the thresholds" % unc
    res, _ = _run(ctx, x, "dmr", [1700, 301])
    assert_matches_oracle(res, ref, x.shape[0], "thresholds, ragged")


@pytest.mark.parametrize("sps", [20, 10, 40, 7])
def test_narrow_filter_generic_sps_all_three_modes(ctx, oracle, sps):
    """The narrow (161-tap) filter at a run-time sps is error-bounded too (the NXDN pipe at sps 20): the generic exact
    symbol evaluation and exact ring recomputation (raw samples staged through LDS), the ordered chain on the approximate
    ring with its interval test, tap-count dependent radius.  Normal, every-symbol-exact and every-run-exact must all
    be the oracle's dibits; ragged and tiny pushes exercise the longer history (100 symbols x sps samples)."""
    from digiham_amd import _taps
    rng = np.random.default_rng(100 + sps)
    chans = []
    for i in range(4):
        dibits = rng.integers(0, 4, 1200 if sps <= 20 else 500)
        x = synth.shape(dibits, sps=sps, taps=_taps.narrow())
        chans.append(synth.impair(x, 50 + i, snr_db=[None, 24, 15, 9][i], dc=[0, 0.1, -0.2, 0][i], delay=3 * i, gain=[1, 0.4, 1.9, 1][i]))
    n = min(len(c) for c in chans)
    x = np.stack([c[:n] for c in chans]).astype(np.float32)
    ref = oracle.chain(x, rrc=2, sps=sps, proto=0)
    total = int(ref["sym_count"].sum())
    kw = dict(rrc="narrow", sps=sps)
    for chunks in ([n], [3 * sps + 1, 5000, 17, 2600]):
        res, (unc, ex) = _run(ctx, x, "none", chunks, **kw)
        assert_matches_oracle(res, ref, x.shape[0], "normal sps %d %s" % (sps, chunks[:1]))
        assert unc.sum() < total // 10
        res, (unc, ex) = _run(ctx, x, "none", chunks, exact_fir=True, **kw)
        assert_matches_oracle(res, ref, x.shape[0], "exact fir sps %d %s" % (sps, chunks[:1]))
    res, (unc, ex) = _run(ctx, x[:, : n // 3], "none", [n // 3], exact_symbols=True, **kw)
    ref3 = oracle.chain(x[:, : n // 3], rrc=2, sps=sps, proto=0)
    assert_matches_oracle(res, ref3, x.shape[0], "exact symbols sps %d" % sps)
    assert int(unc.sum()) == int(ref3["sym_count"].sum())
    res, _ = _run(ctx, x, "none", [n], ordered_timing=True, **kw)
    assert_matches_oracle(res, ref, x.shape[0], "ordered timing sps %d" % sps)
