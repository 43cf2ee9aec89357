"""rrc_filter -> gfsk/fsk_demodulator -> dmr/ysf_decoder through the engine ABI vs the oracle.

Runs on the CPU wave emulation (CPU tier) and on the MI355X (-m gpu).  Dibits, decoder bytes and
decoder events must be bit-exact; float filter outputs must be bit-exact in the default (exact)
mode and within 1e-6 in FAST_FIR mode.
"""
import numpy as np
import pytest

from common import assert_matches_oracle, make_channels, rel_err, run_engine, sha
from digiham_amd import api, synth


@pytest.mark.parametrize("proto", ["dmr", "ysf"])
@pytest.mark.parametrize("chunks", [[10 ** 9], [4800], [1000, 37, 12345, 5, 999]])
def test_full_chain_bit_exact(ctx, oracle, proto, chunks):
    x = make_channels(proto, [1, 2, 3, 4, 5], 30 if proto == "dmr" else 10)
    ref = oracle.chain(x, proto=1 if proto == "dmr" else 2)
    chunks = [min(c, x.shape[1]) for c in chunks]
    res = run_engine(ctx, x, proto, chunks)
    assert_matches_oracle(res, ref, x.shape[0], "%s %s" % (proto, chunks[:2]))
    assert sum(len(f) for f in res["frames"]) > 0


@pytest.mark.parametrize("proto", ["dmr", "ysf"])
def test_split_stages_equal_chain_kernel(ctx, oracle, proto):
    """DH_FLAG_SPLIT_STAGES (slicer and decoder as two launches) and the one-wavefront chain kernel agree with the oracle."""
    x = make_channels(proto, [31, 32, 33], 14)
    ref = oracle.chain(x, proto=1 if proto == "dmr" else 2)
    for split in (True, False):
        res = run_engine(ctx, x, proto, [5000, 12345], split_stages=split)
        assert_matches_oracle(res, ref, len(x), "split" if split else "chain")


def test_chain_golden_vectors(ctx, golden):
    g = golden["chain"]
    for name, proto in (("dmr_a", "dmr"), ("dmr_b", "dmr"), ("ysf_a", "ysf")):
        x = g[name + "_x"][None, :]
        res = run_engine(ctx, x, proto, [x.shape[1]])
        assert (res["syms"][0] == g[name + "_syms"]).all(), name
        assert (res["frames"][0] == g[name + "_out"]).all(), name
        assert res["events"][0].tobytes() == g[name + "_events"].tobytes(), name


def test_unfused_rrc_output_is_bit_exact(ctx, oracle):
    """BASELINE config 2 path: RRC output materialised, then the slicer (exact FIR => identical floats)."""
    x = make_channels("dmr", [7, 8, 9], 12)
    ref = oracle.chain(x, proto=0, keep_filtered=True)
    for chunks in ([x.shape[1]], [3000, 50, 2049]):
        res = run_engine(ctx, x, "none", chunks, keep_filtered=True)
        assert (res["filtered"] == ref["filtered"]).all()
        assert_matches_oracle(res, {"syms": ref["syms"], "sym_count": ref["sym_count"]}, x.shape[0])


@pytest.mark.parametrize("chunks", [[10 ** 9], [3000, 50, 2049], [997], [5, 12345, 1, 1, 4000]])
def test_rrc_and_gfsk_in_one_launch(ctx, oracle, chunks):
    """DH_FLAG_KEEP_FILTERED | DH_FLAG_ONE_LAUNCH (wide filter, sps 10): ONE kernel per push -- the error-bounded slicer also delivers
    the filtered samples.  Every filtered sample of every push within 2.5e-6 of the reference's (rrc_filter.cpp:22-34) relative to
    max(|ref|, rms) -- the split-f16 FIR's accuracy, NOT the 1e-6 of BASELINE configs[1] --, the dibits the reference's bit for bit
    (gfsk_demodulator.cpp:24-107), whatever the push sizes; with a decoder behind it (two launches) frames and events as well."""
    x = make_channels("dmr", [7, 8, 9], 12)
    ref = oracle.chain(x, proto=0, keep_filtered=True)
    chunks = [min(c, x.shape[1]) for c in chunks]
    res = run_engine(ctx, x, "none", chunks, keep_filtered=True, one_launch=True)
    assert res["filtered"].shape == ref["filtered"].shape
    assert rel_err(res["filtered"], ref["filtered"]).max() <= 2.5e-6
    assert_matches_oracle(res, {"syms": ref["syms"], "sym_count": ref["sym_count"]}, x.shape[0], "one launch %s" % chunks[:2])
    if chunks[0] > 10 ** 6:
        full = oracle.chain(x, proto=1)
        res = run_engine(ctx, x, "dmr", [x.shape[1]], keep_filtered=True, one_launch=True)
        assert_matches_oracle(res, full, x.shape[0], "one launch + dmr")
        assert rel_err(res["filtered"], ref["filtered"]).max() <= 2.5e-6


@pytest.mark.parametrize("chunks", [[10 ** 9], [3000, 17, 9000, 1, 4096]])
def test_rrc_and_gfsk_in_one_launch_within_1e6(ctx, oracle, chunks):
    """DH_FLAG_KEEP_FILTERED | DH_FLAG_ONE_LAUNCH | DH_FLAG_FAST_FIR: BASELINE configs[1] in ONE kernel per push.  The error-bounded slicer
    filters with the f32 FMA chain (rrc_filter.cpp:22-34 with fused multiply-adds and the reciprocal gain: every filtered sample within
    1e-6 of the reference's relative to max(|ref|, rms)) and still decides every dibit as the reference does (gfsk_demodulator.cpp:24-107):
    the radius of the FMA chain + exact re-evaluation of what it leaves in doubt."""
    x = make_channels("dmr", [7, 8, 9], 12)
    ref = oracle.chain(x, proto=0, keep_filtered=True)
    chunks = [min(c, x.shape[1]) for c in chunks]
    res = run_engine(ctx, x, "none", chunks, keep_filtered=True, one_launch=True, fast_fir=True)
    assert res["filtered"].shape == ref["filtered"].shape
    assert rel_err(res["filtered"], ref["filtered"]).max() <= 1e-6
    assert_matches_oracle(res, {"syms": ref["syms"], "sym_count": ref["sym_count"]}, x.shape[0], "one launch, FMA floats %s" % chunks[:2])


def test_fast_fir_within_1e6(ctx, oracle):
    """FAST_FIR (FMA) is the float-path variant: 1e-6 relative to max(|ref|, rms(ref))."""
    x = make_channels("dmr", [7, 8], 12)
    ref = oracle.chain(x, proto=0, keep_filtered=True)
    res = run_engine(ctx, x, "none", [x.shape[1]], keep_filtered=True, fast_fir=True)
    assert rel_err(res["filtered"], ref["filtered"]).max() <= 1e-6
    # dibits agree except (rarely) at threshold ties; on this input they agree exactly
    agree = np.mean([np.mean(res["syms"][b][:1000] == ref["syms"][b, :1000]) for b in range(2)])
    assert agree > 0.999


@pytest.mark.parametrize("rrc,sps,levels", [("narrow", 20, "gfsk"), ("none", 10, "fsk"), ("none", 40, "fsk"), ("wide", 10, "fsk"), ("none", 5, "gfsk")])
def test_demod_variants(ctx, oracle, rrc, sps, levels):
    """NXDN-style narrow RRC + sps 20, and the 2-level slicer at sps 10 / 40 (D-Star / POCSAG settings), +invert."""
    rng = np.random.default_rng(sps)
    n = 30000
    if levels == "fsk":
        b = rng.integers(0, 2, n // sps + 1)
        x = (np.repeat(b * 2.0 - 1, sps)[:n] * 0.4 + rng.normal(0, 0.08, n)).astype(np.float32)
    else:
        s = rng.integers(0, 4, n // sps + 1)
        x = synth.shape(s, sps=sps)[:n] + rng.normal(0, 0.02, n).astype(np.float32)
    x = np.stack([x, np.roll(x, 3) * 0.5 + 0.1]).astype(np.float32)
    for invert in ([False, True] if levels == "fsk" else [False]):
        ref = oracle.chain(x, rrc={"none": 0, "wide": 1, "narrow": 2}[rrc], levels=2 if levels == "fsk" else 4,
                           invert=invert, sps=sps, proto=0)
        for chunks in ([n], [777, 4096]):
            res = run_engine(ctx, x, "none", chunks, rrc=rrc, demod=levels, sps=sps, invert=invert)
            assert_matches_oracle(res, {"syms": ref["syms"], "sym_count": ref["sym_count"]}, 2, "%s sps%d" % (rrc, sps))


def test_timing_steps_and_flt_min_quirk(ctx, oracle):
    """Sampling-clock offset forces +-1 timing steps; an all-negative signal keeps max at FLT_MIN (gfsk:111)."""
    rng = np.random.default_rng(1)
    s = rng.integers(0, 4, 3000)
    base = synth.shape(s)
    t = np.arange(len(base))
    fast = np.interp(t * 1.0004, t, base).astype(np.float32)       # +400 ppm
    slow = np.interp(t * 0.9996, t, base).astype(np.float32)
    neg = (base - 1.0).astype(np.float32)
    x = np.stack([fast, slow, neg])
    ref = oracle.chain(x, proto=0)
    res = run_engine(ctx, x, "none", [x.shape[1]])
    assert_matches_oracle(res, {"syms": ref["syms"], "sym_count": ref["sym_count"]}, 3)
    res = run_engine(ctx, x, "none", [997])
    assert_matches_oracle(res, {"syms": ref["syms"], "sym_count": ref["sym_count"]}, 3)
    # the drifting channels really did step: symbol counts differ from n/10
    assert int(ref["sym_count"][0]) != int(ref["sym_count"][1])


def test_tiny_and_empty_pushes(ctx, oracle):
    x = make_channels("dmr", [21], 8)
    ref = oracle.chain(x, proto=1)
    res = run_engine(ctx, x, "dmr", [1, 2, 3, 11, 12, 13, 0 + 80, 81, 5000])
    assert_matches_oracle(res, ref, 1)


def test_sync_loss_and_reacquire(ctx, oracle):
    """Noise burst in the middle: FramePhase falls back to SyncPhase and re-acquires (dmr_phase.cpp:183-186)."""
    for proto in ("dmr", "ysf"):
        x = make_channels(proto, [31, 32], 70 if proto == "dmr" else 20, impair=False)
        rng = np.random.default_rng(0)
        n = x.shape[1]
        x[:, n // 4: n // 4 + n // 2] = rng.normal(0, 0.3, (2, n // 2)).astype(np.float32)
        ref = oracle.chain(x, proto=1 if proto == "dmr" else 2)
        res = run_engine(ctx, x, proto, [6000])
        assert_matches_oracle(res, ref, 2, proto)
        types = set(res["events"][0]["type"].tolist())
        assert (3 in types) or (20 in types)          # a META_RESET happened


def test_dmr_slot_filter(ctx, oracle):
    x = make_channels("dmr", [41, 43], 40)
    for filt in (1, 2, 3, 0):
        ref = oracle.chain(x, proto=1, slot_filter=filt)
        res = run_engine(ctx, x, "dmr", [x.shape[1]], slot_filter=filt)
        assert_matches_oracle(res, ref, 2, "filter %d" % filt)


def test_decoder_only_engine(ctx, oracle):
    """push_symbols: the decoder stage alone, fed dibits (what dmr_decoder sees on its stdin)."""
    from digiham_amd import api
    for proto in ("dmr", "ysf"):
        s = [synth.dmr_stream(51, 20), synth.dmr_stream(52, 20)] if proto == "dmr" else [synth.ysf_stream(53, 6), synth.ysf_stream(54, 6)]
        n = min(len(a) for a in s)
        syms = np.stack([a[:n] for a in s])
        eng = api.Engine(2, n, rrc="none", demod="none", proto=proto, ctx=ctx)
        got_f, got_e = [[], []], [[], []]
        for lo in range(0, n, 1000):
            part = np.ascontiguousarray(syms[:, lo:lo + 1000])
            eng.push_symbols(part, np.full(2, part.shape[1], np.uint32))
            f, fc = eng.frames(); e, ec = eng.events()
            for b in range(2):
                got_f[b].append(f[b, :fc[b]].copy()); got_e[b].append(e[b, :ec[b]].copy())
        for b in range(2):
            d = oracle.Decoder(proto)
            o, ev = d.process(syms[b])
            assert (np.concatenate(got_f[b]) == o).all()
            assert np.concatenate(got_e[b]).tobytes() == ev.tobytes()


def test_dmr_sync_loss_at_every_burst_index(ctx, oracle):
    """The frame-parallel DMR decoder takes a push in chunks of up to 64 bursts (one burst per lane; the slot / superframe machine walks
    their summaries) and leaves a chunk where a burst sends the decoder back to its SyncPhase.  Here every channel loses its signal at a
    different burst -- channel c from burst c on, for 8 to 20 bursts of random dibits (sync counters run down, META_RESET, SyncPhase,
    re-acquisition somewhere inside the next chunk) -- in one push, in pushes of 1 000 symbols and in pushes of a few symbols more than
    a burst: decoder bytes and events equal the reference-shaped oracle decoder's (dmr_phase.cpp:35-47, :163-204) for every index."""
    from digiham_amd import api
    rng = np.random.default_rng(2024)
    B, nb = 72, 150
    rows = []
    for c in range(B):
        s = synth.dmr_stream(300 + c, nb, two_slots=bool(c & 1), lead_in=int(rng.integers(0, 50))).copy()
        lead = len(s) - 144 * nb
        gap = int(rng.integers(8, 21))
        s[lead + 144 * c: lead + 144 * (c + gap)] = rng.integers(0, 4, 144 * gap)
        flips = rng.integers(0, len(s), len(s) // 150)            # scattered wrong dibits: the block codes have work to do
        s[flips] ^= rng.integers(1, 4, len(flips)).astype(np.uint8)
        rows.append(s)
    n = min(len(r) for r in rows)
    syms = np.stack([r[:n] for r in rows])
    want = []
    for b in range(B):
        o, ev = oracle.Decoder("dmr").process(syms[b])
        want.append((o, ev.tobytes()))
    for step in (n, 1000, 151):
        eng = api.Engine(B, min(step, n), rrc="none", demod="none", proto="dmr", ctx=ctx)
        got_f, got_e = [[] for _ in range(B)], [[] for _ in range(B)]
        for lo in range(0, n, step):
            part = np.ascontiguousarray(syms[:, lo:lo + step])
            eng.push_symbols(part, np.full(B, part.shape[1], np.uint32))
            f, fc = eng.frames(); e, ec = eng.events()
            for b in range(B):
                got_f[b].append(f[b, :fc[b]].copy()); got_e[b].append(e[b, :ec[b]].copy())
        for b in range(B):
            assert (np.concatenate(got_f[b]) == want[b][0]).all(), (step, b)
            assert np.concatenate(got_e[b]).tobytes() == want[b][1], (step, b)
        eng.close()
    assert any(3 in set(np.frombuffer(w[1], api.EVENT_DTYPE)["type"].tolist()) for w in want)      # META_RESETs happened


def test_ysf_sync_loss_at_every_frame_index(ctx, oracle):
    """The YSF decoder decodes the codewords, voice blocks and sync words of up to 16 frames AHEAD on the grid the current frame starts; a
    frame that sends it back to its SyncPhase invalidates what was decoded ahead.  Every channel loses its signal at a different frame
    (channel c from frame c on, 14 to 20 frames of random dibits), with scattered wrong dibits elsewhere (clean, repaired and Viterbi-decoded
    codewords side by side), in one push and in pushes of 1 000 and of 481 symbols: decoder bytes and events equal the oracle decoder's."""
    from digiham_amd import api
    rng = np.random.default_rng(2025)
    B, nfr = 24, 48
    rows = []
    for c in range(B):
        s = synth.ysf_stream(400 + c, nfr, mode=("vd2", "vd1", "fr")[c % 3] if c % 4 == 3 else "vd2", lead_in=int(rng.integers(0, 60))).copy()
        lead = len(s) - 480 * nfr if len(s) >= 480 * nfr else 0
        gap = int(rng.integers(14, 21))
        s[lead + 480 * c: lead + 480 * (c + gap)] = rng.integers(0, 4, len(s[lead + 480 * c: lead + 480 * (c + gap)]))
        flips = rng.integers(0, len(s), len(s) // 120)
        s[flips] ^= rng.integers(1, 4, len(flips)).astype(np.uint8)
        rows.append(s)
    n = min(len(r) for r in rows)
    syms = np.stack([r[:n] for r in rows])
    want = []
    for b in range(B):
        o, ev = oracle.Decoder("ysf").process(syms[b])
        want.append((o, ev.tobytes()))
    for step in (n, 1000, 481):
        eng = api.Engine(B, min(step, n), rrc="none", demod="none", proto="ysf", ctx=ctx)
        got_f, got_e = [[] for _ in range(B)], [[] for _ in range(B)]
        for lo in range(0, n, step):
            part = np.ascontiguousarray(syms[:, lo:lo + step])
            eng.push_symbols(part, np.full(B, part.shape[1], np.uint32))
            f, fc = eng.frames(); e, ec = eng.events()
            for b in range(B):
                got_f[b].append(f[b, :fc[b]].copy()); got_e[b].append(e[b, :ec[b]].copy())
        for b in range(B):
            assert (np.concatenate(got_f[b]) == want[b][0]).all(), (step, b)
            assert np.concatenate(got_e[b]).tobytes() == want[b][1], (step, b)
        eng.close()
    assert any(20 in set(np.frombuffer(w[1], api.EVENT_DTYPE)["type"].tolist()) for w in want)     # META_RESETs happened


def test_dmr_event_row_overflow_truncates_like_the_burst_serial_decoder(ctx, oracle):
    """An event row that is too small: the reference-shaped decoder (oracle) drops the events that do not fit, finishes the burst in which
    that happened and stops; dh_engine_sync reports DH_ECAPACITY.  The frame-parallel DMR decoder must cut at the same burst -- its
    pass C notices the overflow on the per-burst event counts of a whole chunk and walks the chunk again up to that burst.  Streams
    built to emit FIVE events per burst (every burst claims slot 0 in its TACT: slot switch + sync + slot type + BPTC + LC) against a
    row sized for four, the overflow falling at different bursts of different chunks per channel."""
    import ctypes as C
    from digiham_amd import api
    from oracle import oracle as O
    rng = np.random.default_rng(5)
    rows = []
    for c in range(6):
        five = [160, 120, 90, 64, 40, 30][c]                    # bursts with five events, then four per burst: the row overflows at another burst per channel
        s = list(rng.integers(0, 4, 7 * c))
        for i in range(160):
            lc = synth.dmr_lc(0, 0, 0, int(rng.integers(1, 1 << 24)), int(rng.integers(1, 1 << 24)))
            slot = 0 if i < five else (i - five + 1) & 1
            s += list(synth.dmr_data_burst(slot, 1, 1, bytes(lc) + bytes(3), "bs_data", rng))
        rows.append(np.array(s, np.uint8))
    n = min(len(r) for r in rows)
    syms = np.stack([r[:n] for r in rows])
    B = len(rows)
    eng = api.Engine(B, n, rrc="none", demod="none", proto="dmr", ctx=ctx)
    eng.push_symbols(np.ascontiguousarray(syms), np.full(B, n, np.uint32))
    with pytest.raises(RuntimeError, match="DH_ECAPACITY"):
        eng.sync()
    lib = ctx.lib
    p, stride, cnt = C.c_void_p(), C.c_size_t(), C.c_void_p()
    assert lib.dh_engine_events(eng._h, C.byref(p), C.byref(stride), C.byref(cnt)) == 0
    cap = stride.value
    erows = np.empty((B, cap), api.EVENT_DTYPE); ecnt = np.empty(B, np.uint32)
    assert lib.dh_copy_to_host(erows.ctypes.data_as(C.c_void_p), p, erows.nbytes) == 0
    assert lib.dh_copy_to_host(ecnt.ctypes.data_as(C.c_void_p), cnt, ecnt.nbytes) == 0
    p2, stride2, cnt2 = C.c_void_p(), C.c_size_t(), C.c_void_p()
    assert lib.dh_engine_frames(eng._h, C.byref(p2), C.byref(stride2), C.byref(cnt2)) == 0
    frows = np.empty((B, stride2.value), np.uint8); fcnt = np.empty(B, np.uint32)
    assert lib.dh_copy_to_host(frows.ctypes.data_as(C.c_void_p), p2, frows.nbytes) == 0
    assert lib.dh_copy_to_host(fcnt.ctypes.data_as(C.c_void_p), cnt2, fcnt.nbytes) == 0
    overflowed = 0
    for b in range(B):
        d = O.Decoder("dmr")
        out = np.zeros(n + 256, np.uint8); oev = np.zeros(cap, O.EVENT_DTYPE)
        no, ne = C.c_size_t(), C.c_size_t()
        O.lib().orc_decoder_process(d._h, O._p(syms[b]), C.c_size_t(n), O._p(out), C.c_size_t(out.size), C.byref(no), O._p(oev), C.c_size_t(cap), C.byref(ne))
        assert ecnt[b] == ne.value, (b, int(ecnt[b]), ne.value)
        assert erows[b, :ecnt[b]].tobytes() == oev[:ne.value].tobytes(), b
        assert fcnt[b] == no.value and (frows[b, :fcnt[b]] == out[:no.value]).all(), b
        overflowed += int(ne.value == cap)
    assert overflowed >= 5
    eng.close()


def test_lc_fields_from_events(ctx):
    """The LC words carried by DH_EV_DMR_LC events decode to the source / target ids the generator put in
    (Digiham::Dmr::Lc getters, lc.cpp:26-43) -- for the voice-header LC (BPTC) and the embedded LC alike."""
    from digiham_amd import api
    rng = np.random.default_rng(77)
    bursts = synth.dmr_call(rng, 0, cc=5, dst=1234, src=5678901, n_superframes=3)
    idle = [synth.dmr_idle_burst(1, 5, rng) for _ in bursts]
    s = list(rng.integers(0, 4, 341))            # the slicer's AGC needs ~100 symbols to settle (zeroed volume ring)
    for a, b in zip(bursts, idle):
        s += a + b
    s += list(rng.integers(0, 4, 300))
    x = synth.shape(np.array(s, np.uint8))[None, :]
    res = run_engine(ctx, x, "dmr", [x.shape[1]])
    lcs = [e for e in res["events"][0] if e["type"] == 4]
    assert {int(e["b"]) for e in lcs} == {0, 1}            # from the voice header and from the embedded signalling
    for e in lcs:
        f = api.parse_lc(e["payload"])
        assert (f["opcode"], f["target"], f["source"]) == (0, 1234, 5678901)


@pytest.mark.parametrize("proto", ["dmr", "ysf", "nxdn", "dstar", "pocsag"])
def test_noise_only_channels_match_oracle(ctx, oracle, proto):
    """Channels that carry nothing but noise: the slicers' timing fallbacks and the decoders' false syncs (NXDN's
    10-dibit sync word fires on noise and decodes frames) must still be the reference's."""
    rng = np.random.default_rng(77)
    n = {"nxdn": 60000, "pocsag": 120000}.get(proto, 30000)
    x = (rng.normal(0, 0.25, (3, n)) + np.array([[0.0], [0.1], [-0.3]])).astype(np.float32)
    ekw = {"nxdn": dict(rrc="narrow", sps=20), "dstar": dict(rrc="none", demod="fsk", sps=10),
           "pocsag": dict(rrc="none", demod="fsk", sps=40, invert=True)}.get(proto, {})
    okw = {"dmr": dict(proto=1), "ysf": dict(proto=2), "nxdn": dict(proto=3, rrc=2, sps=20),
           "dstar": dict(proto=5, rrc=0, levels=2, sps=10), "pocsag": dict(proto=4, rrc=0, levels=2, sps=40, invert=True)}[proto]
    ref = oracle.chain(x, **okw)
    res = run_engine(ctx, x, proto, [n // 3, n - n // 3], **ekw)
    assert_matches_oracle(res, ref, 3, proto + " noise")


def test_custom_rrc_table(ctx, oracle):
    """DH_RRC_CUSTOM: RrcFilter(nZeros, gain, coeffs[]) with the caller's table (include/rrc_filter.hpp:12) -- short,
    long (161 taps) and non-symmetric tables, ragged pushes, several channels; the filtered floats are bit-exact, and a
    demodulator behind the custom filter slices exactly what the oracle's pipe slices."""
    from digiham_amd import api
    rng = np.random.default_rng(41)
    x = make_channels("dmr", [3, 4, 5], 10)
    for nz, gain in ((40, 3.217), (160, 11.5), (1, 2.0), (7, 0.37)):
        taps = rng.normal(0, 1, nz + 1).astype(np.float32)
        ref = np.stack([oracle.Rrc(taps=taps, gain=gain).process(row) for row in x])
        eng = api.Engine(x.shape[0], 4096, rrc="custom", taps=taps, gain=gain, demod="none", proto="none", ctx=ctx)
        got, pos = [], 0
        for c in [4096, 1, 1500, 4096, 333] * 50:
            c = min(c, x.shape[1] - pos)
            if c == 0:
                break
            eng.push(np.ascontiguousarray(x[:, pos:pos + c]))
            got.append(eng.filtered()[:, :c].copy())
            pos += c
        eng.close()
        assert np.concatenate(got, axis=1).tobytes() == ref.tobytes(), nz
    # a custom filter in front of the slicer: here the wide design's own table, so the whole pipe must equal the built-in one
    wide, g = oracle.rrc_taps(False)
    eng = api.Engine(x.shape[0], x.shape[1], rrc="custom", taps=wide, gain=g, demod="gfsk", sps=10, proto="dmr", ctx=ctx)
    eng.push(x)
    s, sc = eng.symbols(); f, fc = eng.frames()
    eng.close()
    ref = oracle.chain(x, proto=1)
    for b in range(x.shape[0]):
        assert sc[b] == ref["sym_count"][b] and (s[b, :sc[b]] == ref["syms"][b, :sc[b]]).all()
        assert fc[b] == ref["out_count"][b] and (f[b, :fc[b]] == ref["out"][b, :fc[b]]).all()


@pytest.mark.parametrize("kw,oproto", [(dict(proto="dmr"), 1), (dict(proto="ysf"), 2), (dict(proto="none", keep_filtered=True), 0),
                                       (dict(proto="none", keep_filtered=True, one_launch=True), 0)])
def test_ragged_pushes_every_channel_at_its_own_pace(ctx, oracle, kw, oproto):
    """dh_engine_push_ragged: each push brings a different number of samples per channel (some none at all); every channel's
    concatenated outputs are those of its whole stream -- what N module instances sharing one launch need
    (include/digiham/shared_engine.hpp)."""
    proto = "ysf" if kw["proto"] == "ysf" else "dmr"
    x = make_channels(proto, [21, 22, 23, 24, 25], 14)
    B, n = x.shape
    ref = oracle.chain(x, proto=oproto, keep_filtered=bool(kw.get("keep_filtered")))
    rng = np.random.default_rng(5)
    cap = 4096
    eng = api.Engine(B, cap, ctx=ctx, **kw)
    pos = np.zeros(B, np.int64)
    syms, frames, filt = [[] for _ in range(B)], [[] for _ in range(B)], [[] for _ in range(B)]
    while (pos < n).any():
        want = rng.choice([0, 1, 7, 333, 1024, 1025, 2500, cap], B)
        cnt = np.minimum(want, n - pos).astype(np.uint32)
        buf = np.zeros((B, cap), np.float32)
        for b in range(B):
            buf[b, :cnt[b]] = x[b, pos[b]:pos[b] + cnt[b]]
            buf[b, cnt[b]:] = np.nan                               # whatever lies behind a channel's count is not its business
        eng.push(buf, n=int(cnt.max()) if cnt.max() else 0, counts=cnt)
        pos += cnt
        s, sc = eng.symbols()
        for b in range(B):
            syms[b].append(s[b, :sc[b]].copy())
        if kw["proto"] != "none":
            f, fc = eng.frames()
            for b in range(B):
                frames[b].append(f[b, :fc[b]].copy())
        if kw.get("keep_filtered"):
            y = eng.filtered()
            for b in range(B):
                filt[b].append(y[b, :cnt[b]].copy())
    eng.close()
    for b in range(B):
        gs = np.concatenate(syms[b])
        assert len(gs) == ref["sym_count"][b] and (gs == ref["syms"][b, :len(gs)]).all(), b
        if kw["proto"] != "none":
            gf = np.concatenate(frames[b])
            assert len(gf) == ref["out_count"][b] and (gf == ref["out"][b, :len(gf)]).all(), b
        if kw.get("one_launch"):                    # (DH_FLAG_ONE_LAUNCH: the matrix-core FIR's floats, 2.5e-6; every sample of every push delivered)
            y = np.concatenate(filt[b])
            assert y.shape == ref["filtered"][b].shape and rel_err(y, ref["filtered"][b]).max() <= 2.5e-6, b
        elif kw.get("keep_filtered"):
            assert (np.concatenate(filt[b]).view(np.uint32) == ref["filtered"][b].view(np.uint32)).all(), b


@pytest.mark.parametrize("proto,oproto,demod,levels", [("dmr", 1, "fsk", 2), ("dstar", 5, "gfsk", 4)])
def test_decoder_behind_the_other_slicer_runs_as_two_launches(ctx, oracle, proto, oproto, demod, levels):
    """The chain kernels are built for their pipe's slicer (4 levels for DMR / YSF / NXDN, 2 for D-Star: k_chain,
    dh_rrc_demod_channel<..., LV>); a decoder configured behind the other one is a legal, if pointless, engine and must take
    the two-launch route with the generic slicer -- same symbols and decoder bytes as the oracle's."""
    x = make_channels("dmr", [41, 42], 6)
    kw = dict(rrc="none", demod=demod, sps=10)
    ref = oracle.chain(x, proto=oproto, rrc=0, levels=levels)
    res = run_engine(ctx, x, proto, [x.shape[1] // 2, x.shape[1] - x.shape[1] // 2], **kw)
    assert_matches_oracle(res, ref, 2, "%s behind %s" % (proto, demod))
