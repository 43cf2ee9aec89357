"""FEC kernels (C ABI) against the reference-generated golden vectors and the oracle.

Each test runs twice: on the CPU wave emulation of the kernel bodies (CPU tier) and, with -m gpu,
through libdigiham_amd.so on the MI355X.  Bit-exact is the bar.
"""
import os

import numpy as np
import pytest

from common import CODES, sha


@pytest.mark.parametrize("code,bits", [c for c in CODES if c[1] <= 20])
def test_block_codes_exhaustive(ctx, golden, code, bits):
    h = golden["hashes"]["codes"][code]
    cw, ok = ctx.block_decode(code, np.arange(1 << bits, dtype=np.uint32))
    assert int(ok.sum()) == h["n_ok"]
    assert sha(cw.astype(np.uint32), ok) == h["sha256_all"]


def test_golay_24_12_strided(ctx, golden):
    h = golden["hashes"]["codes"]["golay_24_12"]
    cw, ok = ctx.block_decode("golay_24_12", np.arange(0, 1 << 24, 16, dtype=np.uint32))
    assert sha(cw.astype(np.uint32), ok) == h["stride16_sha256"]


@pytest.mark.gpu
def test_golay_24_12_exhaustive_gpu(gpu_ctx, golden):
    """All 2^24 words: maximum size of the domain, hash of (corrected words, ok flags) from the reference."""
    h = golden["hashes"]["codes"]["golay_24_12"]
    cw, ok = gpu_ctx.block_decode("golay_24_12", np.arange(1 << 24, dtype=np.uint32))
    assert int(ok.sum()) == h["n_ok"]
    assert sha(cw.astype(np.uint32), ok) == h["sha256_all"]


def test_bptc_golden(ctx, golden):
    g = golden["fec"]
    out, ok = ctx.bptc_196_96(g["bptc_in"])
    assert (ok == g["bptc_ok"]).all()
    assert (out == g["bptc_out"]).all()


@pytest.mark.parametrize("nd", [100, 180])
def test_trellis_golden(ctx, golden, nd):
    g = golden["fec"]
    out, metric = ctx.trellis(g["trellis%d_in" % nd], nd)
    assert (metric == g["trellis%d_metric" % nd]).all()
    assert (out == g["trellis%d_out" % nd]).all()


def test_trellis_ragged_sizes_vs_oracle(ctx, oracle):
    """sizes that are not multiples of 4/8 dibits and batch sizes that do not fill a wavefront."""
    rng = np.random.default_rng(11)
    for nd, n in ((1, 3), (7, 5), (37, 9), (99, 1), (192, 6)):
        x = rng.integers(0, 256, (n, (nd + 3) // 4), dtype=np.uint8)
        if nd % 4:
            x[:, -1] &= (0xFF << (2 * (4 - nd % 4))) & 0xFF
        o1, m1 = ctx.trellis(x, nd)
        o2, m2 = oracle.trellis(x, nd)
        assert (o1 == o2).all() and (m1 == m2).all(), nd


def test_trellis_encode_noise_decode(ctx, oracle):
    """Property: a valid codeword with <= 2 well-separated channel errors decodes to the message, metric = #errors."""
    rng = np.random.default_rng(5)
    nd = 100
    bits = rng.integers(0, 256, (64, 13), dtype=np.uint8)
    bits[:, -1] &= 0xF0
    bits[:, 12] = 0          # 4 tail zeros + padding
    enc = np.stack([oracle.trellis_encode(b, nd) for b in bits])
    noisy = enc.copy()
    for r in range(len(noisy)):
        for bp in (17, 120)[: r % 3]:
            noisy[r, bp // 8] ^= 0x80 >> (bp % 8)
    out, metric = ctx.trellis(noisy, nd)
    assert (out[:, :12] == bits[:, :12]).all()
    assert (metric == np.array([r % 3 for r in range(len(noisy))])).all()


def _encode_from_state(msg_bits, start):
    """Dibits of the K = 5 code (G1 = 1 + D^3 + D^4 -> bit 1, G2 = 1 + D + D^2 + D^4 -> bit 0; src/ysf_decoder/trellis.c:8-25) for a
    message and a 4-bit start state (u(-1) in bit 3 .. u(-4) in bit 0), packed four per byte MSB first like the reference's input."""
    hist = [(start >> 3) & 1, (start >> 2) & 1, (start >> 1) & 1, start & 1]        # u(t-1), u(t-2), u(t-3), u(t-4)
    dib = []
    for u in msg_bits:
        hi = u ^ hist[2] ^ hist[3]
        lo = u ^ hist[0] ^ hist[1] ^ hist[3]
        dib.append((hi << 1) | lo)
        hist = [int(u)] + hist[:3]
    out = np.zeros((len(dib) + 3) // 4, np.uint8)
    for t, d in enumerate(dib):
        out[t // 4] |= d << (2 * (3 - t % 4))
    return out


def test_trellis_clean_codewords_from_every_start_state_vs_reference(ctx, oracle):
    """The clean-codeword shortcut (decoder_core.hpp, dh_viterbi_clean): noiseless words from all 16 start states, every length class,
    and the same words with one bit flipped (every position of one word per length; those must take the full decoder) -- outputs and
    metrics equal the reference's decoder compiled in place (oracle `ref`), or the restatement where `_ref` is not built."""
    which = "ref" if oracle.ref() is not None else "oracle"
    rng = np.random.default_rng(77)
    for nd in (8, 9, 12, 13, 37, 64, 65, 100, 128, 129, 180, 192):
        words = []
        for start in range(16):
            for _ in range(3):
                words.append(_encode_from_state(rng.integers(0, 2, nd), start))
        base = words[5].copy()
        for bit in range(2 * nd):                                # one flipped bit at every position
            w = base.copy(); w[bit // 8] ^= 0x80 >> (bit % 8); words.append(w)
        for _ in range(16):                                       # two flipped bits
            w = words[int(rng.integers(0, 48))].copy()
            for bit in rng.integers(0, 2 * nd, 2):
                w[bit // 8] ^= 0x80 >> (bit % 8)
            words.append(w)
        x = np.stack(words)
        o1, m1 = ctx.trellis(x, nd)
        o2, m2 = oracle.trellis(x, nd, which)
        assert (m1 == m2).all(), nd
        assert (o1 == o2).all(), nd
        assert (m1[:48] == 0).all() and (m1[48:48 + 2 * nd] > 0).all(), nd


def test_trellis_single_dibit_repair_vs_reference(ctx, oracle):
    """The single-dibit repair of the 100-dibit codewords (decoder_core.hpp, dh_ysf_clean100): one wrong dibit -- bit 1, bit 0 or both -- at
    EVERY position of words from every start state, and pairs of wrong dibits at every distance up to 12 across the window the repair is
    allowed in: outputs and metrics equal the reference's decoder compiled in place (oracle `ref`), or the restatement where `_ref` is not built."""
    which = "ref" if oracle.ref() is not None else "oracle"
    rng = np.random.default_rng(99)
    nd = 100
    words = []
    for start in range(16):
        base = _encode_from_state(rng.integers(0, 2, nd), start)
        for pos in range(nd):
            for pat in (1, 2, 3):                                # wrong bit 0, bit 1, both
                w = base.copy(); w[pos // 4] ^= pat << (6 - 2 * (pos % 4)); words.append(w)
    for _ in range(4):
        base = _encode_from_state(rng.integers(0, 2, nd), int(rng.integers(0, 16)))
        for pos in range(0, nd):
            for gap in range(1, 13):
                if pos + gap < nd:
                    w = base.copy()
                    w[pos // 4] ^= int(rng.integers(1, 4)) << (6 - 2 * (pos % 4))
                    w[(pos + gap) // 4] ^= int(rng.integers(1, 4)) << (6 - 2 * ((pos + gap) % 4))
                    words.append(w)
    x = np.stack(words)
    o1, m1 = ctx.trellis(x, nd)
    o2, m2 = oracle.trellis(x, nd, which)
    assert (m1 == m2).all()
    assert (o1 == o2).all()


def test_trellis_repair_window_is_inside_the_proven_margin():
    """DH_YSF_REPAIR_LO / _HI (decoder_core.hpp) against the margin tools/trellis_margin.py derives from the code itself (free distance, the
    weights a difference path must have to stay active from the block's start / to its end): an edit of the bounds that leaves the proven
    range fails here, not in some rare frame."""
    import importlib.util
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("trellis_margin", os.path.join(root, "tools", "trellis_margin.py"))
    tm = importlib.util.module_from_spec(spec); spec.loader.exec_module(tm)
    assert tm.d_free() == 7                                       # a single wrong dibit (weight <= 2) is closer to its codeword than to any detour: 2 * 2 + 1 = 5 <= 7
    # a path from another start state must weigh >= 5 while it is still unmerged at step p: f_start(p + 1); a divergence at or before p that
    # is still open at the end of the 100 steps: f_end(100 - p).  Both grow with the distance from the block's end they guard.
    lo_min = next(p for p in range(100) if all(tm.f_start(q + 1) >= 5 for q in range(p, min(p + 24, 100))))
    hi_max = next(p for p in range(99, -1, -1) if all(tm.f_end(100 - q) >= 5 for q in range(max(p - 24, 0), p + 1)))
    src = open(os.path.join(root, "digiham_amd", "csrc", "decoder_core.hpp")).read()
    lo = int(re.search(r"#define DH_YSF_REPAIR_LO (\d+)", src).group(1)); hi = int(re.search(r"#define DH_YSF_REPAIR_HI (\d+)", src).group(1))
    assert (lo_min, hi_max) == (14, 91)                           # (what the comment above dh_ysf_clean100 quotes)
    assert lo_min <= lo <= hi <= hi_max, (lo_min, lo, hi, hi_max)


def test_trellis_error_clusters_at_the_edges_of_the_repair_window_vs_reference(ctx, oracle):
    """Two and three wrong dibits close together around the ends of the repair window (positions 12..20 and 84..96): whatever the lane-local
    shortcut makes of them -- clean, one repairable dibit, or hand over to the Viterbi decoder -- outputs and metrics are the reference's
    (its tie rules included: equal metrics keep predecessor 0 and the lowest end state, trellis.c:68-99)."""
    which = "ref" if oracle.ref() is not None else "oracle"
    rng = np.random.default_rng(2024)
    nd = 100
    words = []
    for _ in range(40):
        base = _encode_from_state(rng.integers(0, 2, nd), int(rng.integers(0, 16)))
        for lo, hi in ((12, 20), (84, 96)):
            for _ in range(60):
                w = base.copy()
                k = int(rng.integers(2, 4))
                first = int(rng.integers(lo, hi + 1))
                pos = {first}
                while len(pos) < k:
                    pos.add(int(np.clip(first + rng.integers(-6, 7), 0, nd - 1)))
                for q in pos:
                    w[q // 4] ^= int(rng.integers(1, 4)) << (6 - 2 * (q % 4))
                words.append(w)
    x = np.stack(words)
    o1, m1 = ctx.trellis(x, nd)
    o2, m2 = oracle.trellis(x, nd, which)
    assert (m1 == m2).all()
    assert (o1 == o2).all()


def test_crc_whitening_golden(ctx, golden):
    g = golden["fec"]
    for cnt in (4, 10, 20):
        assert (ctx.crc16(g["crc_in"], cnt) == g["crc%d" % cnt]).all()
    for nb in (100, 104, 160):
        k = (nb + 7) // 8
        assert (ctx.whitening(g["crc_in"], nb)[:, :k] == g["whiten%d" % nb]).all()


def test_empty_batches(ctx):
    cw, ok = ctx.block_decode("golay_20_8", np.zeros(0, np.uint32))
    assert cw.size == 0 and ok.size == 0
    out, ok = ctx.bptc_196_96(np.zeros((0, 25), np.uint8))
    assert out.shape == (0, 12)


def test_dvfilter_vs_oracle(ctx, oracle):
    rng = np.random.default_rng(9)
    x = np.stack([rng.normal(0, 9000, 4000).clip(-32768, 32767),
                  20000 * np.sin(np.arange(4000) * 0.3),
                  np.full(4000, 32767.0), rng.integers(-32768, 32768, 4000)]).astype(np.int16)
    ref = np.stack([oracle.DvFilter().process(r) for r in x])
    y, st = ctx.dvfilter(x)
    assert (y == ref).all()
    # state carry: two halves == one run
    y1, st = ctx.dvfilter(x[:, :1500])
    y2, st = ctx.dvfilter(x[:, 1500:], st)
    assert (np.concatenate([y1, y2], axis=1) == ref).all()
