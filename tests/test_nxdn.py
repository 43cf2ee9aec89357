"""NXDN48: narrow RRC -> gfsk_demodulator -s 20 -> nxdn_decoder (examples/nxdn48-decoder.sh:19-21).

* the oracle's frame elements (scrambler, LICH, SACCH, FACCH1, trellis) against tests/golden/nxdn_ref.npz, whose
  expected values come from the reference's own classes compiled in place (PINNED), incl. the four SACCH
  patterns of the NXDN "Common Air Interface Test" document quoted at nxdn_phase.cpp:73-98;
* the engine (CPU wave emulation / MI355X) against the oracle: decoder on dibits, and the whole chain on audio.
"""
import os

import numpy as np
import pytest

from digiham_amd import api, synth, _taps
from common import assert_matches_oracle, run_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nx():
    return np.load(os.path.join(ROOT, "tests", "golden", "nxdn_ref.npz"))


def test_oracle_frame_elements_match_the_reference_vectors(oracle, nx):
    for i in range(len(nx["scr_in"])):
        assert (oracle.nxdn_scramble(nx["scr_in"][i]) == nx["scr_out"][i]).all()
    assert [oracle.nxdn_lich(r) for r in nx["lich_in"]] == list(nx["lich_out"])
    for nb in (72, 192):
        for p, o, m in zip(nx["trellis%d_in" % nb], nx["trellis%d_out" % nb], nx["trellis%d_metric" % nb]):
            got, metric = oracle.nxdn_trellis(p, nb)
            assert metric == m and (got == o).all()
    for name, fn in (("sacch", oracle.nxdn_sacch), ("facch1", oracle.nxdn_facch1)):
        n_ok = 0
        for d, ok, o in zip(nx[name + "_in"], nx[name + "_ok"], nx[name + "_out"]):
            got_ok, got = fn(d)
            assert got_ok == bool(ok) and (not got_ok or (got == o).all())
            n_ok += got_ok
        assert n_ok > 20


def test_oracle_vs_compiled_reference_when_present(oracle):
    if oracle.ref_nxdn() is None:
        pytest.skip("oracle/_ref/libdigiham_ref_nxdn.so not built (no /root/reference)")
    rng = np.random.default_rng(7)
    for _ in range(300):
        d = rng.integers(0, 4, 182).astype(np.uint8)
        assert (oracle.nxdn_scramble(d) == oracle.nxdn_scramble(d, "ref")).all()
        assert oracle.nxdn_lich(d[:8]) == oracle.nxdn_lich(d[:8], "ref")
        a, b = oracle.nxdn_sacch(d[:30]), oracle.nxdn_sacch(d[:30], "ref")
        assert a[0] == b[0] and (not a[0] or (a[1] == b[1]).all())
        a, b = oracle.nxdn_facch1(d[:72]), oracle.nxdn_facch1(d[:72], "ref")
        assert a[0] == b[0] and (not a[0] or (a[1] == b[1]).all())


def test_cai_test_patterns_form_a_vcall_superframe(oracle, nx):
    """The four transmitted SACCH patterns of the CAI test document, put into frames, give one VCALL superframe."""
    rng = np.random.default_rng(1)
    frames = []
    for row in nx["cai_sacch_tx"]:
        body = synth.nxdn_scramble(synth.nxdn_lich_dibits(0x56) + [0] * 30 + list(rng.integers(0, 4, 144)))
        body[8:38] = list(row)                       # the patterns are given as transmitted (already scrambled)
        frames += synth.NXDN_SYNC + body
    s = np.array(list(rng.integers(0, 4, 17)) + frames + [0] * 200, np.uint8)
    out, ev = oracle.Decoder("nxdn").process(s)
    sf = ev[ev["type"] == 34]
    assert len(sf) == 1 and sf[0]["payload"][0] & 0x3F == 0x01          # NXDN_MESSAGE_TYPE_VCALL
    assert [int(e["a"]) for e in ev[ev["type"] == 33]] == [0, 1, 2, 3]


def _decode_symbols(ctx, s, chunk):
    eng = api.Engine(1, max(chunk, 16), rrc="none", demod="none", proto="nxdn", ctx=ctx)
    o, e = [], []
    for lo in range(0, len(s), chunk):
        part = np.ascontiguousarray(s[None, lo:lo + chunk])
        eng.push_symbols(part, np.full(1, part.shape[1], np.uint32))
        f, fc = eng.frames(); ev, ec = eng.events()
        o.append(f[0, :fc[0]].copy()); e.append(ev[0, :ec[0]].copy())
    eng.close()
    return np.concatenate(o), np.concatenate(e)


@pytest.mark.parametrize("seed", [3, 4])
def test_decoder_on_dibits_matches_oracle(ctx, oracle, seed):
    s = synth.nxdn_stream(seed, 40)
    rng = np.random.default_rng(seed)
    noisy = s.copy()
    hit = rng.random(len(s)) < 0.01                  # dibit errors: FEC, CRC failures, lost sync words
    noisy[hit] ^= rng.integers(1, 4, int(hit.sum())).astype(np.uint8)
    for stream in (s, noisy):
        out, ev = oracle.Decoder("nxdn").process(stream)
        assert len(out) > 0 and (ev["type"] == 34).sum() > 0
        for chunk in (len(stream), 1000, 193):
            go, ge = _decode_symbols(ctx, stream, chunk)
            assert len(go) == len(out) and (go == out).all()
            assert ge.tobytes() == ev.tobytes()


def test_full_chain_narrow_rrc_sps20(ctx, oracle):
    chans = []
    for i, seed in enumerate((11, 12, 13)):
        s = synth.nxdn_stream(seed, 14)
        x = synth.shape(s, sps=20, taps=_taps.narrow())
        chans.append(synth.impair(x, seed, snr_db=[None, 22, 16][i], dc=[0, 0.1, -0.2][i], delay=7 * i, gain=[1, 0.5, 1.7][i]))
    n = min(len(c) for c in chans)
    x = np.stack([c[:n] for c in chans])
    ref = oracle.chain(x, rrc=2, sps=20, proto=3)
    assert ref["out_count"].sum() > 0
    for chunks in ([n], [4800, 12345]):
        for split in (False, True):                    # one-wavefront chain kernel / slicer and decoder as two launches
            res = run_engine(ctx, x, "nxdn", chunks, rrc="narrow", sps=20, split_stages=split)
            assert_matches_oracle(res, ref, len(x), "nxdn %s %s" % (chunks[:1], "split" if split else "chain"))
