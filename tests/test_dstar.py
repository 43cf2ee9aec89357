"""D-Star (examples/dstar-decoder.sh: fsk_demodulator -s 10 | dstar_decoder).

* scrambler and CRC: the oracle against tests/golden/dstar_ref.npz, whose expected values come from the reference's own
  src/dstar_decoder/{scrambler,crc}.cpp compiled in place (PINNED);
* the radio header (de-interleave + K=3 Viterbi + CRC; header.cpp needs ICU, so UNPINNED): known-answer round trips
  through an independent encoder, error tolerance as header.cpp:37 states it (path metric <= 10);
* the decoder on bits and the whole chain on 2-level FSK audio, engine (CPU wave emulation / MI355X) against the oracle.
"""
import os

import numpy as np
import pytest

from digiham_amd import api, synth
from common import assert_matches_oracle, run_engine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EV_HEADER, EV_VOICE_START, EV_SYNC_VOICE, EV_MESSAGE, EV_SIMPLE, EV_FRAME_SYNC, EV_META_RESET = range(64, 71)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "dstar_ref.npz"))


def test_oracle_scrambler_and_crc_match_the_reference_vectors(oracle, gold):
    for i, o in zip(gold["scr_in"], gold["scr_out"]):
        assert (oracle.dstar_scramble(i) == o).all()
    assert (synth.dstar_pn(660) == gold["scr_out"][0]).all()           # the generator's own sequence, and the product's below
    for d, n, c, v in zip(gold["crc_data"], gold["crc_len"], gold["crc_cand"], gold["crc_valid"]):
        assert oracle.dstar_crc_valid(d[:n], int(c)) == bool(v)
        assert (synth.dstar_crc(d[:n]) == int(c)) == bool(v)
    assert 0 < gold["crc_valid"].sum() < len(gold["crc_valid"])


def test_oracle_header_known_answers(oracle):
    rng = np.random.default_rng(5)
    for t in range(40):
        call = "".join(chr(int(c)) for c in rng.integers(48, 91, 8))
        h = synth.dstar_header_bytes("DB0ABC G", "DB0ABC B", "CQCQCQ", call, "X%d" % t, flags=(int(rng.integers(0, 128)), 0, 0))
        bits = synth.dstar_header_bits(h)
        ok, out = oracle.dstar_header_parse(bits)
        assert ok and bytes(out) == h
        few = bits.copy(); few[rng.choice(660, 4, replace=False)] ^= 1  # isolated errors are corrected
        ok, out = oracle.dstar_header_parse(few)
        assert ok and bytes(out) == h
        many = bits.copy(); many[rng.choice(660, 60, replace=False)] ^= 1
        assert not oracle.dstar_header_parse(many)[0]                   # metric > 10 or CRC failure
        bad = bytearray(h); bad[int(rng.integers(0, 39))] ^= 0x10       # a consistent code word with a wrong FCS
        assert not oracle.dstar_header_parse(synth.dstar_header_bits(bytes(bad)))[0]


def _decode_bits(ctx, bits, chunk):
    eng = api.Engine(1, max(chunk, 64), rrc="none", demod="none", proto="dstar", ctx=ctx)
    o, e = [], []
    for lo in range(0, len(bits), chunk):
        part = np.ascontiguousarray(bits[None, lo:lo + chunk])
        eng.push_symbols(part, np.full(1, part.shape[1], np.uint32))
        f, fc = eng.frames(); ev, ec = eng.events()
        o.append(f[0, :fc[0]].copy()); e.append(ev[0, :ec[0]].copy())
    eng.close()
    return np.concatenate(o), np.concatenate(e)


def _headers(ev):
    h = ev[ev["type"] == EV_HEADER]
    return [(int(a["b"]), bytes(a["payload"][:24]) + bytes(b["payload"][:17])) for a, b in zip(h[0::2], h[1::2])]


@pytest.mark.parametrize("seed", [3, 4])
def test_decoder_on_bits_matches_oracle(ctx, oracle, seed):
    clean, infos = synth.dstar_stream(seed, 8)
    noisy, _ = synth.dstar_stream(seed, 8, ber=0.004)                  # corrected headers, missed syncs, broken slow data
    for stream in (clean, noisy):
        out, ev = oracle.Decoder("dstar").process(stream)
        if stream is clean:                                            # the stream says what it should: headers, messages, voice
            heads = _headers(ev)
            for i in infos:
                if i["kind"] not in (3, 6, 7):
                    assert (0, i["header"]) in heads
                if i["kind"] != 4 and i["frames"] > 44:
                    assert (1, i["header"]) in heads
                assert i["kind"] == 6 or any(bytes(m["payload"][:20]).decode("latin1").rstrip() == i["message"].rstrip()
                                             for m in ev[ev["type"] == EV_MESSAGE])
            simple = b"".join(bytes(e["payload"][:e["len"]]) for e in ev[ev["type"] == EV_SIMPLE])
            assert all(i["simple"] in simple for i in infos if i["kind"] != 6)
            assert len(out) % 9 == 0 and len(out) // 9 > sum(i["frames"] for i in infos) // 2
            assert (ev["type"] == EV_META_RESET).sum() >= sum(i["kind"] != 6 for i in infos)
        for chunk in (len(stream), 1000, 97):
            go, ge = _decode_bits(ctx, stream, chunk)
            assert len(go) == len(out) and (go == out).all()
            assert ge.tobytes() == ev.tobytes()


def test_decoder_on_noise_and_sync_storms(ctx, oracle):
    """Random bits (false syncs, false header starts), runs of sync words, and a header cut by the end of the input."""
    rng = np.random.default_rng(11)
    h = synth.dstar_header_bits(synth.dstar_header_bytes("A", "B", "C", "D"))
    parts = [rng.integers(0, 2, 30000).astype(np.uint8)]
    for _ in range(40):
        parts += [np.array([1, 0] * 16 + synth.DSTAR_FRAME_SYNC, np.uint8), rng.integers(0, 2, int(rng.integers(0, 700))).astype(np.uint8),
                  np.array(synth.DSTAR_VOICE_SYNC, np.uint8), rng.integers(0, 2, int(rng.integers(0, 300))).astype(np.uint8),
                  np.array(synth.DSTAR_TERMINATOR, np.uint8)]
    parts += [np.array([1, 0] * 16 + synth.DSTAR_FRAME_SYNC, np.uint8), h[:500]]
    stream = np.concatenate(parts)
    out, ev = oracle.Decoder("dstar").process(stream)
    assert (ev["type"] == EV_VOICE_START).sum() >= 5 and (ev["type"] == EV_META_RESET).sum() >= 5
    for chunk in (len(stream), 4096, 661):
        go, ge = _decode_bits(ctx, stream, chunk)
        assert len(go) == len(out) and (go == out).all()
        assert ge.tobytes() == ev.tobytes()


def test_full_chain_fsk_sps10(ctx, oracle):
    chans = []
    for i, seed in enumerate((21, 22, 23)):
        bits, _ = synth.dstar_stream(seed, 3)
        x = synth.fsk_shape(bits, sps=10)
        chans.append(synth.impair(x, seed, snr_db=[None, 20, 14][i], dc=[0.0, 0.1, -0.05][i], delay=3 * i, gain=[1, 0.6, 1.5][i]))
    n = min(len(c) for c in chans)
    x = np.stack([c[:n] for c in chans])
    ref = oracle.chain(x, rrc=0, levels=2, sps=10, proto=5)
    assert ref["out_count"].min() > 0
    for chunks in ([n], [48000, 12345]):
        for split in (False, True):                    # one-wavefront chain kernel / slicer and decoder as two launches
            res = run_engine(ctx, x, "dstar", chunks, rrc="none", demod="fsk", sps=10, split_stages=split)
            assert_matches_oracle(res, ref, len(x), "dstar %s %s" % (chunks[:1], "split" if split else "chain"))


def test_tiny_empty_and_ragged_pushes(ctx, oracle):
    """Pushes of 0..13 samples, one just under / over a bit, and a header that arrives split over many pushes."""
    bits, _ = synth.dstar_stream(31, 2)
    x = synth.impair(synth.fsk_shape(bits, sps=10), 3, snr_db=25, dc=0.02)[None, :]
    ref = oracle.chain(x, rrc=0, levels=2, sps=10, proto=5)
    assert ref["out_count"][0] > 0
    res = run_engine(ctx, x, "dstar", [1, 0, 2, 3, 9, 10, 11, 13, 0, 659, 661, 6600, 37, 5000], rrc="none", demod="fsk", sps=10)
    assert_matches_oracle(res, ref, 1)
    # decoder-only engine fed 1 .. 7 bits at a time through a header and the first superframe
    rng = np.random.default_rng(8)
    stream = np.concatenate([rng.integers(0, 2, 50).astype(np.uint8), synth.dstar_transmission(rng, n_superframes=1)[0]])
    out, ev = oracle.Decoder("dstar").process(stream)
    eng = api.Engine(1, 64, rrc="none", demod="none", proto="dstar", ctx=ctx)
    o, e, lo, k = [], [], 0, 0
    while lo < len(stream):
        n = [1, 7, 0, 3, 5, 2][k % 6]; k += 1
        part = np.zeros((1, 64), np.uint8); part[0, :min(n, len(stream) - lo)] = stream[lo:lo + n]
        eng.push_symbols(part, np.full(1, min(n, len(stream) - lo), np.uint32))
        lo += n
        f, fc = eng.frames(); evs, ec = eng.events()
        o.append(f[0, :fc[0]].copy()); e.append(evs[0, :ec[0]].copy())
    eng.close()
    go, ge = np.concatenate(o), np.concatenate(e)
    assert len(go) == len(out) and (go == out).all() and ge.tobytes() == ev.tobytes()
    assert (ev["type"] == EV_HEADER).sum() >= 2


def test_decoder_only_rows_at_odd_addresses(ctx, oracle):
    """Three channels in a symbol buffer with an odd row pitch: the packed fast path funnel-shifts unaligned rows."""
    streams = [synth.dstar_stream(s, 4)[0] for s in (41, 42, 43)]
    n = min(len(s) for s in streams)
    chunk, pitch = 1501, 1503
    eng = api.Engine(3, pitch, rrc="none", demod="none", proto="dstar", ctx=ctx)
    got_o, got_e = [[] for _ in range(3)], [[] for _ in range(3)]
    for lo in range(0, n, chunk):
        part = np.zeros((3, pitch), np.uint8)
        cnt = np.zeros(3, np.uint32)
        for b in range(3):
            seg = streams[b][lo:min(lo + chunk, n)]
            part[b, :len(seg)] = seg; cnt[b] = len(seg)
        eng.push_symbols(part, cnt)
        f, fc = eng.frames(); ev, ec = eng.events()
        for b in range(3):
            got_o[b].append(f[b, :fc[b]].copy()); got_e[b].append(ev[b, :ec[b]].copy())
    eng.close()
    for b in range(3):
        out, ev = oracle.Decoder("dstar").process(streams[b][:n])
        go, ge = np.concatenate(got_o[b]), np.concatenate(got_e[b])
        assert len(out) > 0 and len(go) == len(out) and (go == out).all() and ge.tobytes() == ev.tobytes()
