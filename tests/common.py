"""Shared helpers for the parity tests."""
import hashlib

import numpy as np

from digiham_amd import api, synth

CODES = [("hamming_7_4", 7), ("hamming_13_9", 13), ("hamming_15_11", 15), ("hamming_16_11", 16),
         ("quadratic_residue", 16), ("golay_20_8", 20), ("golay_24_12", 24)]


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def make_channels(proto, seeds, n_units, impair=True):
    """A [B][n] float32 batch of synthetic channels with assorted impairments."""
    chans = []
    for i, seed in enumerate(seeds):
        if proto == "dmr":
            s = synth.dmr_stream(seed, n_units, two_slots=(seed % 2 == 1))
        else:
            s = synth.ysf_stream(seed, n_units, mode=["vd2", "vd1", "fr", "vd2", "datafr"][seed % 5])
        x = synth.shape(s)
        if impair:
            x = synth.impair(x, seed, snr_db=[None, 25, 14, 30, 18][i % 5], dc=[0, 0.2, -0.3, 0.05, 0][i % 5],
                             delay=(seed * 7) % 23, gain=[1, 0.3, 2, 1, 0.6][i % 5])
        chans.append(x)
    n = min(len(c) for c in chans)
    return np.stack([c[:n] for c in chans])


def run_engine(ctx, x, proto, chunks, **kw):
    """Push x[B][n] through an engine in the given chunk sizes; returns per-channel concatenated outputs."""
    B, n = x.shape
    eng = api.Engine(B, max(chunks), proto=proto, ctx=ctx, **kw)
    syms = [[] for _ in range(B)]
    frames = [[] for _ in range(B)]
    evs = [[] for _ in range(B)]
    filt = []
    pos = i = 0
    while pos < n:
        c = min(chunks[i % len(chunks)], n - pos)
        i += 1
        eng.push(np.ascontiguousarray(x[:, pos:pos + c]))
        pos += c
        if eng.has_demod:
            s, sc = eng.symbols()
            for b in range(B):
                syms[b].append(s[b, :sc[b]].copy())
        if eng.has_proto:
            f, fc = eng.frames()
            e, ec = eng.events()
            for b in range(B):
                frames[b].append(f[b, :fc[b]].copy())
                evs[b].append(e[b, :ec[b]].copy())
        if eng.keep_filtered:
            filt.append(eng.filtered()[:, :c].copy())
    eng.sync()
    eng.close()
    cat = lambda parts, dt: [np.concatenate(p) if p else np.zeros(0, dt) for p in parts]
    return {"syms": cat(syms, np.uint8), "frames": cat(frames, np.uint8), "events": cat(evs, api.EVENT_DTYPE),
            "filtered": np.concatenate(filt, axis=1) if filt else None}


def assert_matches_oracle(res, ref, B, what=""):
    for b in range(B):
        rs = ref["syms"][b, :ref["sym_count"][b]]
        assert len(res["syms"][b]) == len(rs) and (res["syms"][b] == rs).all(), "%s ch %d: dibits differ" % (what, b)
        if "out" in ref and res["frames"][b] is not None and ref["out_count"] is not None:
            rf = ref["out"][b, :ref["out_count"][b]]
            assert len(res["frames"][b]) == len(rf) and (res["frames"][b] == rf).all(), "%s ch %d: frame bytes differ" % (what, b)
            re = ref["events"][b, :ref["event_count"][b]]
            assert len(res["events"][b]) == len(re), "%s ch %d: event count %d vs %d" % (what, b, len(res["events"][b]), len(re))
            assert res["events"][b].tobytes() == re.tobytes(), "%s ch %d: events differ" % (what, b)


def rel_err(a, ref):
    """|a - ref| / max(|ref|, rms(ref)) -- the 1e-6 float tolerance of BASELINE.md section 4."""
    ref = ref.astype(np.float64)
    rms = np.sqrt(np.mean(ref ** 2)) + 1e-30
    return np.abs(a.astype(np.float64) - ref) / np.maximum(np.abs(ref), rms)
