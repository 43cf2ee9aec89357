"""The arithmetic contract pieces that random signals rarely exercise."""
import numpy as np
import pytest

from digiham_amd import _taps


def _div_ref(x, gain):
    return (x.astype(np.float64) / gain).astype(np.float32)


def near_tie_operands(gain, start=0x3E000000, count=1 << 26, chunk=1 << 24, ulps=2000):
    """Floats x whose quotient x/gain lies within `ulps` ulp(double) of a float rounding tie.  Within 4 ulp the
    kernels take the exact-division fallback; from 5 ulp on they trust the reciprocal product, so this band is
    where a wrong error bound would show (found by scanning: the density is ~ulps * 3.7e-9 per float)."""
    r = 1.0 / gain
    hits = []
    for lo in range(start, start + count, chunk):
        x = np.arange(lo, lo + chunk, dtype=np.uint32).view(np.float32)
        m = (x.astype(np.float64) * r).view(np.uint64) & np.uint64(0x1FFFFFFF)
        sel = np.abs(m.astype(np.int64) - 0x10000000) <= ulps
        if sel.any():
            hits.append(x[sel].copy())
    return np.concatenate(hits) if hits else np.zeros(0, np.float32)


@pytest.mark.parametrize("narrow", [False, True])
def test_gain_division_is_exactly_the_double_division(ctx, narrow):
    """(float)((double)sum / gain) (rrc_filter.cpp:33) via reciprocal-multiply + tie fallback: bit-exact for
    random floats of every exponent, for subnormal results, and for operands found next to a float rounding
    tie of the quotient (which force the fallback branch)."""
    gain = _taps.NARROW_GAIN if narrow else _taps.WIDE_GAIN
    rng = np.random.default_rng(17)
    bits = rng.integers(0, 1 << 32, 2_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x)]
    signal = (rng.normal(0, 1, 1_000_000) * 10.0 ** rng.uniform(-6, 3, 1_000_000)).astype(np.float32)
    tiny = (rng.normal(0, 1, 100_000) * 1e-37).astype(np.float32)
    ties = near_tie_operands(gain)
    assert ties.size > 0, "scan found no near-tie operand: enlarge the scan"
    allx = np.concatenate([x, signal, tiny, np.array([0.0, -0.0, 1.0, -1.0], np.float32), ties, -ties])
    got = ctx.debug_div_gain(allx, narrow)
    ref = _div_ref(allx, gain)
    assert got.tobytes() == ref.tobytes()
    assert ties.size > 100


@pytest.mark.parametrize("divisor", [5, 6, 10, 20, 40])
def test_division_by_samples_per_symbol_is_the_ieee_division(ctx, divisor):
    """`volume_sum / samplesPerSymbol` (gfsk_demodulator.cpp:83; 6 = the `sum / (highestEval - lowestEval)` of sps 20, :91) as three instructions (reciprocal product, exact FMA
    residual, correction; dsp_core.hpp: dh_div_const): equal to the IEEE float division for random floats of every
    exponent, signal-sized values, the guard band around 2^-100 / 2^100, zeros, infinities and NaN."""
    rng = np.random.default_rng(23 + divisor)
    bits = rng.integers(0, 1 << 32, 2_000_000, dtype=np.uint64).astype(np.uint32)
    signal = (rng.normal(0, 1, 1_000_000) * 10.0 ** rng.uniform(-6, 3, 1_000_000)).astype(np.float32)
    edge = np.concatenate([np.arange(b - 4096, b + 4096, dtype=np.uint32) for b in (0x0D800000, 0x71800000, 0x00800000, 0x7F800000, 1 << 31)]
                          + [np.arange(0, 8192, dtype=np.uint32)])
    x = np.concatenate([bits.view(np.float32), signal, edge.view(np.float32), (edge | np.uint32(1 << 31)).view(np.float32)])
    got = ctx.debug_div_const(x, divisor)
    with np.errstate(all="ignore"):
        ref = x / np.float32(divisor)
    nan = np.isnan(ref)
    assert (np.isnan(got) == nan).all()
    assert got[~nan].tobytes() == ref[~nan].tobytes()


@pytest.mark.gpu
def test_division_by_ten_all_floats(gpu_ctx):
    """Every one of the 2^32 float bit patterns through dh_div_const(x, 10): the quotient is the IEEE one (computed on the
    device by torch's own division, itself spot-checked against numpy per chunk)."""
    import torch
    dev = gpu_ctx.mem.device
    ten = torch.tensor(10.0, dtype=torch.float32, device=dev)
    chunk = 1 << 27
    for lo in range(0, 1 << 32, chunk):
        bits = torch.arange(lo, lo + chunk, dtype=torch.int64, device=dev).to(torch.int32)     # wraps into the negative half
        x = bits.view(torch.float32)
        got = gpu_ctx.debug_div_const(x, 10)
        ref = x / ten
        nan = torch.isnan(ref)
        assert bool((torch.isnan(got) == nan).all())
        assert bool((got.view(torch.int32)[~nan] == ref.view(torch.int32)[~nan]).all()), "chunk at %#x" % lo
        probe = x[::65537].cpu().numpy()
        with np.errstate(all="ignore"):
            host = probe / np.float32(10)
        ok = ~np.isnan(host)
        assert ref[::65537].cpu().numpy()[ok].tobytes() == host[ok].tobytes()


# ---------------------------------------------------------------------------------------------------------------
# The split-f16 FIR on the matrix cores (dsp_core.hpp, "DH_FIR_F16").  Its error radius is computed from (a) the split of
# samples and taps into two halves each and (b) assumption (H1) about v_mfma_f32_16x16x32_f16:
#     |D - (C + sum a b)| <= 41 u mu,  u = 2^-24,  mu >= every addend and partial sum  (here: mu = |C| + sum |a b|)
# These tests pin both on the device (and run against the harness's stand-in on the CPU tier).
def _f16_exact(a, b, c):
    """exact C + A @ B per tile in float64 (products of halves are exact in 22 bits; 33 addends: 2^-48 relative at worst),
    and the magnitude sum |C| + sum |a b|"""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    return c.astype(np.float64) + a64 @ b64, np.abs(c.astype(np.float64)) + np.abs(a64) @ np.abs(b64)


def test_f16_split_is_round_to_nearest_with_subnormals(ctx):
    """h1 = f16(x s), h2 = f16((x s - h1) 2^11): identical to numpy's IEEE conversions (round to nearest even, subnormal halves
    kept), and x s = h1 + 2^-11 h2 to within 2^-24 for |x s| <= 1."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-1, 1, 200_000), rng.uniform(-1, 1, 100_000) * 10.0 ** rng.uniform(-9, 0, 100_000),
                        [0.0, -0.0, 1.0, -1.0, 2.0 ** -14, 2.0 ** -15, 2.0 ** -24, 2.0 ** -25, 3 * 2.0 ** -26, 0.99999994, 6.1e-5, 5.96e-8]]).astype(np.float32)
    for scale in (1.0, 0.5, 2.0 ** -7, 2.0 ** 20):
        xin = (x / np.float32(scale)).astype(np.float32)                 # so that x s covers [-1, 1] again
        h1, h2 = ctx.debug_f16_split(xin, scale)
        xs = xin * np.float32(scale)
        r1 = xs.astype(np.float16)
        r2 = ((xs - r1.astype(np.float32)) * np.float32(2048)).astype(np.float16)
        assert np.array_equal(h1, r1) and np.array_equal(h2, r2)      # (as values: the device's fused scale-and-convert turns -0 into +0)
        rec = h1.astype(np.float64) + h2.astype(np.float64) / 2048
        assert np.abs(rec - xs.astype(np.float64)).max() <= 2.0 ** -24


def _mfma_cases(rng, tiles):
    a = rng.uniform(-1, 1, (tiles, 16, 32)); b = rng.uniform(-1, 1, (tiles, 32, 16)); c = rng.uniform(-1, 1, (tiles, 16, 16))
    kind = np.arange(tiles) % 6
    spread_a = 2.0 ** -rng.integers(0, 12, (tiles, 16, 32)); spread_b = 2.0 ** -rng.integers(0, 12, (tiles, 32, 16))
    a = np.where((kind == 1)[:, None, None] | (kind == 4)[:, None, None], a * spread_a, a)
    b = np.where((kind == 1)[:, None, None] | (kind == 4)[:, None, None], b * spread_b, b)
    a = np.where((kind == 2)[:, None, None] & (rng.integers(0, 16, a.shape) == 0), a * 64, a)       # a few dominant terms
    c = np.where((kind == 0)[:, None, None], 0.0, c)
    c = np.where((kind == 3)[:, None, None], c * 8, c)                                              # accumulator larger than the products
    c = np.where((kind == 4)[:, None, None], c * 1e-3, c)
    # kind 5: adversarial for a truncating adder -- one large product per group of eight, the other seven just below 2^-23 of it, all positive
    big = np.zeros((tiles, 16, 32)); big[:, :, ::8] = 1.0
    a5 = np.where(big > 0, 1.0, (2047.0 / 1024.0) * 2.0 ** -12); b5 = np.where(np.swapaxes(big, 1, 2)[:, :, :16] > 0, 1.0, 2.0 ** -12)
    a = np.where((kind == 5)[:, None, None], a5, a); b = np.where((kind == 5)[:, None, None], b5, b)
    return a.astype(np.float16), b.astype(np.float16), c.astype(np.float32)


def test_matrix_core_f16_sum_stays_inside_the_assumed_bound(ctx):
    """(H1) on 6 000 tiles x 256 dot products: uniform terms, wide exponent spreads, dominant terms, large and small accumulators,
    and a construction that maximises what a truncating aligner loses.  Also what the bound needs besides: subnormal halves
    take part, NaN and infinity come through."""
    rng = np.random.default_rng(11)
    a, b, c = _mfma_cases(rng, 6000)
    d = ctx.debug_mfma_f16(a, b, c)
    exact, mag = _f16_exact(a, b, c)
    u = 2.0 ** -24
    ratio = np.abs(d.astype(np.float64) - exact) / (u * mag + 1e-300)
    assert ratio.max() <= 41.0, ratio.max()
    assert ratio.max() <= 12.0, "far more than observed when the bound was set (worst 3.5 on random sums, 7.1 on the adversarial tiles): %g" % ratio.max()
    # subnormal halves are operands like any other
    a0 = np.zeros((1, 16, 32), np.float16); b0 = np.zeros((1, 32, 16), np.float16); c0 = np.zeros((1, 16, 16), np.float32)
    a0[0, 0, 0] = 2.0 ** -20; b0[0, 0, 0] = 1.0                       # subnormal A
    a0[0, 1, 0] = 1.0; b0[0, 0, 1] = 2.0 ** -20                       # subnormal B
    a0[0, 2, 3] = 2.0 ** -20; b0[0, 3, 2] = 2.0 ** -20                # both
    d0 = ctx.debug_mfma_f16(a0, b0, c0)
    assert d0[0, 0, 0] == 2.0 ** -20 and d0[0, 1, 1] == 2.0 ** -20 and d0[0, 2, 2] == 2.0 ** -40
    # non-finite operands reach the result (the kernels detect a NaN sample behind the FIR)
    a1 = np.ones((1, 16, 32), np.float16); b1 = np.ones((1, 32, 16), np.float16); c1 = np.zeros((1, 16, 16), np.float32)
    a1[0, 3, 17] = np.nan; a1[0, 5, 2] = np.inf
    d1 = ctx.debug_mfma_f16(a1, b1, c1)
    assert np.isnan(d1[0, 3]).all() and np.isinf(d1[0, 5]).all() and np.isfinite(d1[0, 4]).all()


def test_f16_fir_error_radius_covers_a_brute_force_comparison(ctx):
    """End to end for the wide filter: samples split as the kernels do, the three chained MFMAs of the main sum and the six of the
    second sum through dh_debug_mfma_f16, combined -- against the exact (float64) filter output and against the reference's
    rounded-product chain.  The distance must stay inside the radius the engine uses (dh_f16_error_coefficient, restated
    here), with room to spare."""
    rng = np.random.default_rng(3)
    taps = _taps.wide().astype(np.float32); gain = _taps.WIDE_GAIN
    g1 = taps.astype(np.float16); g2 = ((taps.astype(np.float64) - g1.astype(np.float64)) * 2048).astype(np.float16)
    delta = taps.astype(np.float64) - g1.astype(np.float64) - g2.astype(np.float64) / 2048
    l1, l1g1, l1g2 = np.abs(taps.astype(np.float64)).sum(), np.abs(g1.astype(np.float64)).sum(), np.abs(g2.astype(np.float64)).sum()
    u = 2.0 ** -24
    kl1 = (82.01 * l1 + 1.0001 * l1 + 123.0 * l1g1 * (1 + 2.0 ** -11) + 246.0 / 2048 * (l1g2 + 0.5 * l1g1)
           + 2.0 * (0.5 * l1g2 / 2 ** 22 + np.abs(delta).sum() * (1 + 2.0 ** -12) + u * l1) / u + 3.1 * l1)
    radius_per_xmax = kl1 * u / gain                                   # the engine's coefficient is 1.44 x this (slicer head room)
    T = 96
    worst = 0.0
    for amp in (1.0, 3e-4, 700.0):
        x = (rng.normal(0, 0.4, (T, 16, 16 + 95)) * amp).astype(np.float32)          # tile t, block m: samples x[t, m, 0 .. 110]
        xmax = np.abs(x).max(axis=(1, 2), keepdims=True)
        ex = np.floor(np.log2(xmax)).astype(np.int64)
        scale = (2.0 ** (-1 - ex)).astype(np.float32)                                # max |x s| in [0.5, 1)
        xs = x * scale
        h1 = xs.astype(np.float16); h2 = ((xs - h1.astype(np.float32)) * np.float32(2048)).astype(np.float16)
        # Toeplitz operands: A[m][p] = h[m, p] (p = 0..95), B[p][n] = g[p - n]
        def toeplitz(g):
            B = np.zeros((96, 16), np.float16)
            for n in range(16):
                B[n:n + 81, n] = g
            return B
        B1, B2 = toeplitz(g1), toeplitz(g2)
        def chain(parts):
            acc = np.zeros((T, 16, 16), np.float32)
            for h, B in parts:
                for s in range(3):
                    acc = ctx.debug_mfma_f16(h[:, :, 32 * s:32 * s + 32], np.broadcast_to(B[32 * s:32 * s + 32], (T, 32, 16)), acc)
            return acc
        main = chain([(h1, B1)]); second = chain([(h1, B2), (h2, B1)])
        k1 = (np.float32(1.0 / gain) / scale).astype(np.float32); k2 = k1 * np.float32(2.0 ** -11)
        y = (second.astype(np.float32) * k2 + (main * k1).astype(np.float32)).astype(np.float32)     # (numpy rounds the product: one more rounding than the kernel's fma)
        # reference: rounded products, rounded sums in tap order, double division (rrc_filter.cpp:22-34)
        ref = np.zeros((T, 16, 16), np.float32)
        for k in range(81):
            win = np.stack([x[:, :, n + k] for n in range(16)], axis=2)
            ref = (ref + (np.float32(taps[k]) * win).astype(np.float32)).astype(np.float32)
        ref = (ref.astype(np.float64) / gain).astype(np.float32)
        err = np.abs(y.astype(np.float64) - ref.astype(np.float64)) / (radius_per_xmax * xmax)
        worst = max(worst, err.max())
    assert worst <= 1.0, worst
    assert worst <= 0.25, "the filtered samples sit much closer to the reference's than the radius: %g of it" % worst


@pytest.mark.parametrize("narrow", [False, True])
def test_rrc_tables_equal_the_references(oracle, narrow):
    """The two mkshape tables exist ONCE in this repository (digiham_amd/csrc/rrc_taps.h; the oracle includes that header),
    so no parity test can notice a typo in them.  tests/golden/rrc_taps_ref_hashes.json holds the SHA-256 of the numbers in the
    reference's own src/rrc_filter/rrc_filter.cpp:36-115 (made by tests/golden/make_golden_taps.py in the development
    container); the expanded tables -- dh_rrc_expand_taps() as compiled into the oracle, and what the Python side parses for
    the tests -- and the two gains must hash to the same."""
    import hashlib
    import json
    import os
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rrc_taps_ref_hashes.json")))["narrow" if narrow else "wide"]
    taps, gain = oracle.rrc_taps(narrow)
    assert taps.dtype == np.float32 and taps.size == want["taps"] == want["nzeros"] + 1
    assert hashlib.sha256(taps.astype("<f4").tobytes()).hexdigest() == want["taps_float32_sha256"]
    assert np.float64(gain).tobytes().hex() == want["gain_float64_hex"]
    parsed = _taps.narrow() if narrow else _taps.wide()
    assert hashlib.sha256(parsed.astype("<f4").tobytes()).hexdigest() == want["taps_float32_sha256"]
    assert np.float64(_taps.NARROW_GAIN if narrow else _taps.WIDE_GAIN).tobytes().hex() == want["gain_float64_hex"]
