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


@pytest.mark.parametrize("divisor", [5, 10, 20, 40])
def test_division_by_samples_per_symbol_is_the_ieee_division(ctx, divisor):
    """`volume_sum / samplesPerSymbol` (gfsk_demodulator.cpp:83) as three instructions (reciprocal product, exact FMA
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
