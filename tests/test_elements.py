"""Burst / frame element parsers against the REFERENCE's own classes.

tests/golden/elements_ref.{npz,json} were produced by the reference's csdr-free sources compiled where they lie
(src/dmr_decoder/{cach,tact,emb,embedded,slottype,lc}.cpp, src/ysf_decoder/fich.cpp, src/pocsag_decoder/codeword.cpp,
src/dstar_decoder/header.cpp; oracle/Makefile `ref`, tests/golden/make_golden_elements.py).  Three layers are held to them:

* the oracle's restatement (oracle/elements.c, pocsag.c, dstar.c) -- CPU tier, exhaustive where the reference's domain
  is enumerable (all 2^24 CACHs, 2^16 EMB words, 2^20 slot-type words);
* `_ref` itself, when present, reproduces the committed vectors (guards the fixtures);
* the product: decoder-only engines (CPU wave emulation and, with -m gpu, libdigiham_amd.so) are fed dibit streams in
  which every burst / frame carries one golden input, and the SLOTTYPE / EMB / LC / FICH / CODEWORD / HEADER events they
  emit are compared with the reference's results -- position by position.
"""
import json
import os

import numpy as np
import pytest

from common import sha
from digiham_amd import api, synth

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "elements_ref.npz")), json.load(open(os.path.join(HERE, "golden", "elements_ref_hashes.json")))


def _exhaustive_hashes(E, O):
    import hashlib
    s = hashlib.sha256()
    for start in range(0, 1 << 24, 1 << 20):
        s.update(E.dmr_cach(O.all_cach_dibits(start, 1 << 20)).tobytes())
    o1, c1 = E.dmr_emb(np.arange(1 << 16))
    o2, c2 = E.dmr_slottype(np.arange(1 << 20))
    return {"dmr_cach_all_2^24": s.hexdigest(), "dmr_emb_all_2^16": sha(o1, c1), "dmr_slottype_all_2^20": sha(o2, c2)}, (o1, c1, o2, c2)


# ------------------------------------------------------------------ oracle (and _ref) against the committed vectors
@pytest.mark.parametrize("which", ["oracle", "ref"])
def test_elements_vs_reference_vectors(oracle, gold, which):
    v, h = gold
    if which == "ref" and any(oracle.ref_lib(n) is None for n in ("dmr", "ysf", "pocsag", "dstar")):
        pytest.skip("oracle/_ref not built here (needs /root/reference)")
    E = oracle.Elements(which)
    got, (o1, c1, o2, c2) = _exhaustive_hashes(E, oracle)
    for k, val in got.items():
        assert val == h[k], k
    assert (np.packbits(o1[:, 0]) == v["emb_ok_bits"]).all() and (np.packbits(o2[:, 0]) == v["slottype_ok_bits"]).all()
    assert (o1[:1024] == v["emb_first_out"]).all() and (c1[:1024] == v["emb_first_cor"]).all()
    assert (o2[:1024] == v["slottype_first_out"]).all() and (c2[:1024] == v["slottype_first_cor"]).all()
    assert (E.dmr_cach(oracle.all_cach_dibits(0, 4096)) == v["cach_first"]).all()
    sample = np.concatenate([oracle.all_cach_dibits(int(i), 1) for i in v["cach_sample_idx"]])
    assert (E.dmr_cach(sample) == v["cach_sample"]).all()
    assert (E.dmr_embedded_lc(v["elc_prev"], v["elc_frags"], v["elc_nfrags"]) == v["elc_out"]).all()
    f, d7 = E.dmr_lc(v["lc_in"])
    assert (f == v["lc_fields"]).all() and (d7 == v["lc_data7"]).all()
    o, d = E.ysf_fich(v["fich_in"])
    assert (o == v["fich_out"]).all() and (d == v["fich_data"]).all()
    o, w = E.pocsag_codeword(v["cw_in"])
    assert (o == v["cw_out"]).all() and (w == v["cw_words"]).all()
    raw = np.unpackbits(v["dh_in_bits"], axis=1)[:, :660]
    ok, data, text = E.dstar_header(raw)
    assert (ok == v["dh_ok"]).all() and (data == v["dh_data"]).all()
    if which == "ref":                                     # host-side elements exist on the reference side only
        assert (E.dmr_gps(v["dmr_gps_in"]).view(np.uint32) == v["dmr_gps_out"].view(np.uint32)).all()
        c, t, n = E.dmr_talkeralias(v["ta_blocks"], v["ta_order"])
        assert (c == v["ta_complete"]).all() and (t == v["ta_text"]).all() and (n == v["ta_len"]).all()
        ok, ll = E.ysf_gps(v["ysf_gps_in"])
        assert (ok == v["ysf_gps_ok"]).all() and (ll.view(np.uint32) == v["ysf_gps_out"].view(np.uint32)).all()
        assert (text == v["dh_text"]).all()


def test_parse_lc_vs_reference_getters(gold):
    """api.parse_lc (what the Python host hands out for a DH_EV_DMR_LC payload) == Digiham::Dmr::Lc's getters."""
    v, _ = gold
    for lc, f, d7 in zip(v["lc_in"], v["lc_fields"], v["lc_data7"]):
        p = api.parse_lc(lc)
        assert (p["opcode"], p["feature_set_id"], p["source"], p["target"]) == tuple(int(x) for x in f)
        assert bytes(p["data"]) == bytes(d7)


# ------------------------------------------------------------------ the product: element results as decoder events
def _bits(words, nbits):
    w = np.asarray(words, np.uint64)
    return np.stack([((w >> np.uint64(nbits - 1 - k)) & np.uint64(1)).astype(np.uint8) for k in range(nbits)], axis=1)


def _to_dibits(bits):
    return (bits[:, 0::2] << 1 | bits[:, 1::2]).astype(np.uint8)


CACH = [np.array(synth.dmr_cach(s), np.uint8) for s in (0, 1)]
BS_DATA, BS_VOICE = np.array(synth.DMR_SYNC["bs_data"], np.uint8), np.array(synth.DMR_SYNC["bs_voice"], np.uint8)
LEAD = 29


def _run_symbols(ctx, proto, streams, chunk=None):
    """streams [B][n] dibits (or bits) -> per-channel event arrays of a decoder-only engine."""
    B, n = streams.shape
    chunk = n if chunk is None else chunk
    eng = api.Engine(B, chunk, rrc="none", demod="none", proto=proto, ctx=ctx)
    evs = [[] for _ in range(B)]
    for lo in range(0, n, chunk):
        part = np.ascontiguousarray(streams[:, lo:lo + chunk])
        eng.push_symbols(part, np.full(B, part.shape[1], np.uint32))
        e, ec = eng.events()
        for b in range(B):
            evs[b].append(e[b, :ec[b]].copy())
    eng.close()
    return [np.concatenate(e) for e in evs]


def _dmr_streams(bursts, B):
    """bursts [N][144] in transmission order, N = B * K (K even): channel b gets bursts b*K .. (b+1)*K-1, slots alternating."""
    N = len(bursts)
    K = N // B
    assert K * B == N and K % 2 == 0
    bursts = bursts.reshape(B, K, 144).copy()
    bursts[:, 0::2, :12] = CACH[0]
    bursts[:, 1::2, :12] = CACH[1]
    rng = np.random.default_rng(5)
    lead = np.broadcast_to(rng.integers(0, 4, LEAD).astype(np.uint8), (B, LEAD))
    tail = np.zeros((B, 160), np.uint8)                # the phase needs more than one burst in hand (decoder.cpp:25)
    return np.concatenate([lead, bursts.reshape(B, K * 144), tail], axis=1), K


def _events_at(ev, typ, end):
    """events of one type keyed by the symbol index of their burst / frame (those of the zero padding after `end` dropped)"""
    sel = ev[(ev["type"] == typ) & (ev["sym_index"] < end)]
    return {int(e["sym_index"]): e for e in sel}


def _size(ctx, full, small):
    return small if type(ctx.mem).__name__ == "NumpyMemory" else full


def test_slottype_events_vs_reference(ctx, oracle, gold):
    """Every 20-bit slot-type word (all 2^20 with -m gpu) in a data burst: the SLOTTYPE event carries what the reference's
    SlotType::parse / getDataType / getColorCode give (dmr_phase.cpp:244-249, slottype.cpp:9-22)."""
    v, h = gold
    o, c = oracle.Elements("oracle").dmr_slottype(np.arange(1 << 20))
    assert sha(o, c) == h["dmr_slottype_all_2^20"]      # the expectation below IS the reference's table
    step = _size(ctx, 1, 509)
    words = np.arange(0, 1 << 20, step, dtype=np.uint32)
    words = words[:len(words) // 64 * 64]
    B = _size(ctx, 1024, 4)
    rng = np.random.default_rng(6)
    bursts = rng.integers(0, 4, (len(words), 144)).astype(np.uint8)
    st = _to_dibits(_bits(words, 20))
    bursts[:, 61:66], bursts[:, 90:95], bursts[:, 66:90] = st[:, :5], st[:, 5:], BS_DATA
    streams, K = _dmr_streams(bursts, B)
    evs = _run_symbols(ctx, "dmr", streams, chunk=_size(ctx, None, 20000))
    n_ok = 0
    for b in range(B):
        got = _events_at(evs[b], api_ev("DMR_SLOTTYPE"), LEAD + 144 * K)
        exp = {}
        for k in range(K):
            w = int(words[b * K + k])
            if o[w, 0]:
                exp[LEAD + 144 * k] = (k & 1, int(o[w, 2]), int(o[w, 1]))
        assert sorted(got) == sorted(exp), "channel %d: slot-type events at other bursts than the reference decodes" % b
        for pos, (slot, dt, cc) in exp.items():
            e = got[pos]
            assert (int(e["a"]), int(e["b"]), int(e["payload"][0])) == (slot, dt, cc)
        n_ok += len(exp)
    assert n_ok == int(o[words, 0].sum()) and n_ok > 0


def api_ev(name):
    from digiham_amd import _capi
    return {"DMR_SYNC": 1, "DMR_LC": 4, "DMR_SLOTTYPE": 7, "DMR_EMB": 8, "YSF_FICH": 16, "POCSAG_CODEWORD": 48, "DSTAR_HEADER": 64}[name]


def _voice_superframes(mids, rng):
    """mids [S][5][24]: the 24 middle dibits of bursts B..F of S superframes -> [S][6][144] bursts (A carries the voice sync)"""
    S = len(mids)
    sf = rng.integers(0, 4, (S, 6, 144)).astype(np.uint8)
    sf[:, 0, 66:90] = BS_VOICE
    sf[:, 1:, 66:90] = mids
    return sf


def _interleave_slots(sf0, sf1):
    """two slot streams [S][6][144] -> transmission order [S * 12][144] (slot 0 burst, slot 1 burst, ...)"""
    S = len(sf0)
    out = np.empty((S, 6, 2, 144), np.uint8)
    out[:, :, 0], out[:, :, 1] = sf0, sf1
    return out.reshape(S * 12, 144)


def _emb_mid(cc, lcss, frag16=None):
    return np.array(synth.dmr_emb_mid(cc, lcss, [0] * 16 if frag16 is None else list(frag16)), np.uint8)


def test_emb_events_vs_reference(ctx, oracle, gold):
    """Every 16-bit EMB word (all 2^16 with -m gpu) as burst B of a voice superframe, on both slots: the EMB event carries
    the reference's Emb::parse / getLcss / getColorCode (dmr_phase.cpp:117-134, emb.cpp:9-24); bursts C..F are clean."""
    v, h = gold
    o, c = oracle.Elements("oracle").dmr_emb(np.arange(1 << 16))
    assert sha(o, c) == h["dmr_emb_all_2^16"]
    words = np.arange(0, 1 << 16, _size(ctx, 1, 131), dtype=np.uint32)
    B = _size(ctx, 256, 2)
    words = words[:len(words) // (2 * B) * 2 * B]
    rng = np.random.default_rng(7)
    S = len(words) // 2
    clean = _emb_mid(3, 0)
    sfs = []
    for s in (0, 1):
        w = words[s::2]
        mids = np.broadcast_to(clean, (S, 5, 24)).copy()
        e = _to_dibits(_bits(w, 16))
        mids[:, 0, 0:4], mids[:, 0, 20:24] = e[:, :4], e[:, 4:]
        mids[:, 0, 4:20] = rng.integers(0, 4, (S, 16))
        sfs.append(_voice_superframes(mids, rng))
    streams, K = _dmr_streams(_interleave_slots(*sfs), B)
    evs = _run_symbols(ctx, "dmr", streams, chunk=_size(ctx, None, 20000))
    per = S // B
    n_ok = 0
    for b in range(B):
        got = _events_at(evs[b], api_ev("DMR_EMB"), LEAD + 144 * K)
        for i in range(per):
            for s in (0, 1):
                w = int(words[2 * (b * per + i) + s])
                pos = LEAD + 144 * (12 * i + 2 + s)            # burst B of superframe i on slot s
                if o[w, 0]:
                    e = got[pos]
                    assert (int(e["a"]), int(e["b"]), int(e["payload"][0])) == (s, int(o[w, 2]), int(o[w, 1])), (b, i, s, hex(w))
                    n_ok += 1
                else:
                    assert pos not in got, (b, i, s, hex(w))
                for f in range(2, 6):                          # C..F decode cleanly whatever B was
                    e = got[LEAD + 144 * (12 * i + 2 * f + s)]
                    assert (int(e["a"]), int(e["b"]), int(e["payload"][0])) == (s, 0, 3)
    assert n_ok == int(o[words, 0].sum()) and n_ok > 0


def _bytes_to_dibits(b):
    b = np.asarray(b, np.uint8)
    return np.stack([(b >> 6) & 3, (b >> 4) & 3, (b >> 2) & 3, b & 3], axis=-1).reshape(b.shape[:-1] + (-1,))


def test_embedded_lc_events_vs_reference(ctx, gold):
    """The reference's EmbeddedCollector vectors (encoded + noisy LCs, three fragments + a stale quarter, short and
    over-long sequences, random data) as embedded signalling of voice superframes: an LC event (b = 1) with the reference's
    nine bytes appears at the LCSS-stop burst exactly where EmbeddedCollector::getLc succeeded (embedded.cpp:20-94,
    dmr_phase.cpp:136-164).  A first superframe leaves `prev` in the collector, as the golden script did."""
    v, _ = gold
    n = _size(ctx, len(v["elc_out"]) // 64 * 64, 0)
    idx = np.arange(n) if n else np.concatenate([np.arange(0, 4096, 171), np.arange(4096, 6144, 37)])[:80]
    B = _size(ctx, 64, 2)
    idx = idx[:len(idx) // (2 * B) * 2 * B]
    rng = np.random.default_rng(8)
    LCSS = {0: [], 1: [2], 2: [1, 2], 3: [1, 3, 2], 4: [1, 3, 3, 2], 5: [1, 3, 3, 3, 2]}
    sfs, stop_burst = [[], []], {}
    for j, i in enumerate(idx):
        s = j & 1
        nf = int(v["elc_nfrags"][i])
        pm = np.stack([_emb_mid(3, l, _bytes_to_dibits(v["elc_prev"][i, 4 * k:4 * k + 4])) for k, l in enumerate([1, 3, 3, 2])] + [_emb_mid(3, 0)])
        tm = [_emb_mid(3, l, _bytes_to_dibits(v["elc_frags"][i, 4 * k:4 * k + 4])) for k, l in enumerate(LCSS[nf])]
        tm = np.stack(tm + [_emb_mid(3, 0)] * (5 - len(tm)))
        sfs[s].append(_voice_superframes(np.stack([pm, tm]), rng))
        stop_burst[(j // 2, s)] = (i, nf)
    streams, K = _dmr_streams(_interleave_slots(np.concatenate(sfs[0]), np.concatenate(sfs[1])), B)
    evs = _run_symbols(ctx, "dmr", streams, chunk=_size(ctx, None, 20000))
    per = len(idx) // 2 // B
    n_ok = 0
    for b in range(B):
        lcs = {int(e["sym_index"]): e for e in evs[b] if e["type"] == api_ev("DMR_LC") and e["b"] == 1}
        for p in range(per):
            for s in (0, 1):
                i, nf = stop_burst[(b * per + p, s)]
                base = LEAD + 144 * (24 * p + 12 + s)          # burst A of the second (test) superframe
                test_positions = [base + 288 * f for f in range(1, 6)]
                hits = [q for q in test_positions if q in lcs]
                if v["elc_out"][i, 0]:
                    assert hits == [base + 288 * nf], (i, nf, hits)      # the stop burst (a fifth fragment is ignored)
                    e = lcs[hits[0]]
                    assert int(e["a"]) == s and int(e["len"]) == 9 and (e["payload"][:9] == v["elc_out"][i, 1:]).all(), i
                    n_ok += 1
                else:
                    assert hits == [], (i, nf)
    assert n_ok == int(v["elc_out"][idx, 0].sum()) and n_ok > 0


def test_fich_events_vs_reference(ctx, gold):
    """The reference's Fich::parse vectors (encoded FICHs with 0..11 bit errors, random dibits) as the FICH of consecutive
    frames: a FICH event with the reference's 32-bit word appears exactly at the frames the reference decodes
    (ysf_phase.cpp:60-66, fich.cpp:12-52)."""
    v, _ = gold
    n = _size(ctx, len(v["fich_in"]), 96)
    B = _size(ctx, 64, 2)
    idx = (np.arange(n) * (len(v["fich_in"]) // n))[:n // B * B]
    rng = np.random.default_rng(9)
    frames = rng.integers(0, 4, (len(idx), 480)).astype(np.uint8)
    frames[:, :20] = np.array(synth.YSF_SYNC, np.uint8)
    frames[:, 20:120] = v["fich_in"][idx]
    K = len(idx) // B
    lead = np.broadcast_to(rng.integers(0, 4, LEAD).astype(np.uint8), (B, LEAD))
    streams = np.concatenate([lead, frames.reshape(B, K * 480), np.zeros((B, 500), np.uint8)], axis=1)
    evs = _run_symbols(ctx, "ysf", streams, chunk=_size(ctx, None, 30000))
    n_ok = 0
    for b in range(B):
        got = _events_at(evs[b], api_ev("YSF_FICH"), LEAD + 480 * K)
        exp = {LEAD + 480 * k: int(v["fich_data"][idx[b * K + k]]) for k in range(K) if v["fich_out"][idx[b * K + k], 0]}
        assert sorted(got) == sorted(exp), "channel %d" % b
        for pos, word in exp.items():
            assert int.from_bytes(bytes(got[pos]["payload"][:4]), "big") == word
        n_ok += len(exp)
    assert n_ok == int(v["fich_out"][idx, 0].sum()) and n_ok > 0


def test_pocsag_codeword_events_vs_reference(ctx, gold):
    """The reference's Codeword::parse vectors as the 16 codewords of consecutive batches: a CODEWORD event with the
    reference's corrected word appears exactly where Codeword::parse succeeded (pocsag_phase.cpp:54-57, codeword.cpp:9-31)."""
    v, _ = gold
    n = _size(ctx, len(v["cw_in"]), 256)
    B = _size(ctx, 16, 2)
    idx = (np.arange(n) * (len(v["cw_in"]) // n))[:n // (16 * B) * 16 * B]
    sync = np.array(synth._bits_of(synth.POCSAG_SYNC, 32), np.uint8)
    K = len(idx) // B // 16                                   # batches per channel
    cw = (v["cw_in"][idx] != 0).astype(np.uint8).reshape(B, K, 16 * 32)       # the slicer only delivers 0 / 1
    batches = np.concatenate([np.broadcast_to(sync, (B, K, 32)), cw], axis=2)
    rng = np.random.default_rng(10)
    lead = np.broadcast_to(rng.integers(0, 2, LEAD).astype(np.uint8), (B, LEAD))
    streams = np.concatenate([lead, batches.reshape(B, K * 17 * 32), np.broadcast_to(sync, (B, 32)), np.zeros((B, 40), np.uint8)], axis=1)
    evs = _run_symbols(ctx, "pocsag", streams)
    n_ok = 0
    for b in range(B):
        got = _events_at(evs[b], api_ev("POCSAG_CODEWORD"), LEAD + K * 17 * 32)
        exp = {}
        for j in range(K):
            for k in range(16):
                i = idx[(b * K + j) * 16 + k]
                if v["cw_out"][i, 0]:
                    exp[LEAD + 32 * (17 * j + 1 + k)] = (k, int(v["cw_words"][i, 0]))
        assert sorted(got) == sorted(exp), "channel %d" % b
        for pos, (k, word) in exp.items():
            assert int(got[pos]["a"]) == k and int.from_bytes(bytes(got[pos]["payload"][:4]), "big") == word
        n_ok += len(exp)
    assert n_ok == int(v["cw_out"][idx, 0].sum()) and n_ok > 0


def test_dstar_header_events_vs_reference(ctx, gold):
    """The reference's Header::parseFromHeader vectors, each behind a header sync and followed by a terminator: the two
    HEADER events carry the reference's 41 decoded bytes exactly for the vectors it accepts as voice headers
    (dstar_phase.cpp:39-54, header.cpp:22-54)."""
    v, _ = gold
    raw = np.unpackbits(v["dh_in_bits"], axis=1)[:, :660]
    n = _size(ctx, len(raw) // 64 * 64, 24)
    B = _size(ctx, 64, 2)
    idx = (np.arange(n) * (len(raw) // n))[:n // B * B]
    K = len(idx) // B
    hsync = np.array([0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 1, 1, 0, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 0], np.uint8)     # dstar_phase.hpp:18
    term = np.array([1, 0] * 16 + [0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0], np.uint8)               # dstar_phase.hpp:32-41
    rng = np.random.default_rng(11)
    UNIT = 24 + 660 + 72 + 48 + 40
    items = np.zeros((len(idx), UNIT), np.uint8)
    items[:, :24], items[:, 24:684] = hsync, raw[idx]
    items[:, 684:756] = rng.integers(0, 2, (len(idx), 72))
    items[:, 756:804] = term
    lead = np.zeros((B, LEAD), np.uint8)
    streams = np.concatenate([lead, items.reshape(B, K * UNIT), np.zeros((B, 800), np.uint8)], axis=1)
    evs = _run_symbols(ctx, "dstar", streams)
    n_ok = 0
    for b in range(B):
        hd = evs[b][(evs[b]["type"] == api_ev("DSTAR_HEADER")) & (evs[b]["b"] == 0)]
        got = {}
        for e in hd:
            got.setdefault(int(e["sym_index"]), {})[int(e["a"])] = bytes(e["payload"][:e["len"]])
        exp = {}
        for k in range(K):
            i = idx[b * K + k]
            if v["dh_ok"][i] and not (v["dh_data"][i, 0] >> 7) & 1:
                exp[LEAD + UNIT * k + 684] = bytes(v["dh_data"][i])
        assert sorted(got) == sorted(exp), "channel %d" % b
        for pos, h41 in exp.items():
            assert got[pos][0] + got[pos][1] == h41
        n_ok += len(exp)
    assert n_ok > 0


def test_tact_slot_tracking_vs_reference(ctx, gold):
    """The reference's Cach::parse / Tact::getSlot results (golden sample of random 24-bit CACHs, mixed with clean ones)
    drive the TDMA slot of every burst: the `a` field of each burst's SLOTTYPE event must follow dmr_phase.cpp:65-95 fed
    with the REFERENCE's TACT outcome for that CACH."""
    v, _ = gold
    n = _size(ctx, 4096, 128)
    B = _size(ctx, 16, 2)
    K = n // B
    rng = np.random.default_rng(12)
    idx = v["cach_sample_idx"][:n]
    cach = np.stack([((idx >> (22 - 2 * k)) & 3).astype(np.uint8) for k in range(12)], axis=1)
    has, tslot = v["cach_sample"][:n, 0].copy(), v["cach_sample"][:n, 3].copy()
    bursts = rng.integers(0, 4, (n, 144)).astype(np.uint8)
    st = np.array(synth.bits_to_dibits(synth._bits_of(synth.block_encode("golay_20_8", (5 << 4) | 3), 20)), np.uint8)   # cc 5, CSBK
    bursts[:, 61:66], bursts[:, 90:95], bursts[:, 66:90] = st[:5], st[5:], BS_DATA
    bursts[:, :12] = cach
    clean = rng.random(n) < 0.6                                  # runs of clean, alternating CACHs let the stability counter climb
    for i in np.nonzero(clean)[0]:
        bursts[i, :12] = CACH[i & 1]; has[i] = 1; tslot[i] = i & 1
    lead = np.broadcast_to(rng.integers(0, 4, LEAD).astype(np.uint8), (B, LEAD))
    streams = np.concatenate([lead, bursts.reshape(B, K * 144), np.zeros((B, 160), np.uint8)], axis=1)
    evs = _run_symbols(ctx, "dmr", streams)
    for b in range(B):
        got = _events_at(evs[b], api_ev("DMR_SLOTTYPE"), LEAD + 144 * K)
        slot, stab, exp = -1, 0, {}
        for k in range(K):
            i = b * K + k
            nxt = (slot ^ 1) & 0xFF
            if has[i]:
                if tslot[i] != nxt:
                    if stab < 5:
                        stab, slot = 0, int(tslot[i])
                    else:
                        stab -= 1
                        slot = nxt if slot != -1 else slot
                else:
                    stab, slot = min(stab + 1, 100), nxt
            elif slot != -1:
                stab = -100 if stab < -100 else stab - 1
                slot = nxt
            if slot != -1:
                exp[LEAD + 144 * k] = slot
        assert sorted(got) == sorted(exp), "channel %d" % b
        assert all(int(got[p]["a"]) == s and int(got[p]["b"]) == 3 and int(got[p]["payload"][0]) == 5 for p, s in exp.items()), "channel %d" % b
        assert len(set(exp.values())) == 2
