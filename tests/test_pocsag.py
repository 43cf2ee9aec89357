"""POCSAG (examples/pocsag-decoder.sh: fsk_demodulator -i -s 40 | pocsag_decoder).

* BCH(31,21): oracle and product against tests/golden/pocsag_ref.npz, whose expected values come from the reference's
  own bch_31_21.c compiled in place (PINNED);
* the decoder on bits and the whole chain on 2-level FSK audio, engine (CPU wave emulation / MI355X) against the oracle;
  the decoded pages are the module's output bytes (`address:<n>;message:<text>` lines).
"""
import os

import numpy as np
import pytest

from digiham_amd import api, synth
from common import assert_matches_oracle, run_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "pocsag_ref.npz"))


def test_oracle_bch_matches_the_reference_vectors(oracle, gold):
    out, ok = oracle.block_decode("bch_31_21", gold["bch_in"])
    assert (ok == gold["bch_ok"]).all() and (np.where(ok == 1, out, 0) == gold["bch_out"]).all()
    assert 5000 < int(ok.sum()) < len(ok)


def test_product_bch_matches_the_reference_vectors(ctx, gold):
    out, ok = ctx.block_decode("bch_31_21", gold["bch_in"])
    assert (ok == gold["bch_ok"]).all() and (np.where(ok == 1, out, 0) == gold["bch_out"]).all()


def _decode_bits(ctx, bits, chunk):
    eng = api.Engine(1, max(chunk, 64), rrc="none", demod="none", proto="pocsag", ctx=ctx)
    o, e = [], []
    for lo in range(0, len(bits), chunk):
        part = np.ascontiguousarray(bits[None, lo:lo + chunk])
        eng.push_symbols(part, np.full(1, part.shape[1], np.uint32))
        f, fc = eng.frames(); ev, ec = eng.events()
        o.append(f[0, :fc[0]].copy()); e.append(ev[0, :ec[0]].copy())
    eng.close()
    return np.concatenate(o), np.concatenate(e)


@pytest.mark.parametrize("seed", [3, 4])
def test_decoder_on_bits_matches_oracle(ctx, oracle, seed):
    bits, sent = synth.pocsag_stream(seed, 8)
    rng = np.random.default_rng(seed)
    noisy = bits.copy()
    hit = rng.random(len(bits)) < 0.004                # bit errors: BCH corrections, dropped messages, lost sync words
    noisy[hit] ^= 1
    for stream in (bits, noisy):
        out, ev = oracle.Decoder("pocsag").process(stream)
        lines = bytes(out).decode("latin1").split("\n")
        if stream is bits:                             # most pages come out verbatim (a transmission right behind another is missed)
            assert sum(("address:%d;message:%s" % (a, t)) in lines for a, f, t in sent) >= len(sent) // 2
        for chunk in (len(stream), 1000, 97):
            go, ge = _decode_bits(ctx, stream, chunk)
            assert len(go) == len(out) and (go == out).all()
            assert ge.tobytes() == ev.tobytes()


def test_full_chain_fsk_inverted_sps40(ctx, oracle):
    chans = []
    for i, seed in enumerate((21, 22)):
        bits, _ = synth.pocsag_stream(seed, 3)
        x = synth.fsk_shape(bits, sps=40, invert=True)
        chans.append(synth.impair(x, seed, snr_db=[None, 20][i], dc=[0.0, 0.1][i], delay=11 * i, gain=[1, 0.6][i]))
    n = min(len(c) for c in chans)
    x = np.stack([c[:n] for c in chans])
    ref = oracle.chain(x, rrc=0, levels=2, invert=True, sps=40, proto=4)
    assert ref["out_count"].sum() > 0
    for chunks in ([n], [48000, 12345]):
        res = run_engine(ctx, x, "pocsag", chunks, rrc="none", demod="fsk", sps=40, invert=True)
        assert_matches_oracle(res, ref, len(x), "pocsag %s" % chunks[:1])
