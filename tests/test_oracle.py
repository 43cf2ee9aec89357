"""CPU tier: the oracle against the reference's own FEC code and the committed golden vectors."""
import numpy as np
import pytest

from common import CODES, sha


def test_oracle_fec_matches_golden_hashes(oracle, golden):
    """Exhaustive for n <= 20 bits; Golay(24,12) on the stride-16 subset (the GPU tier does all 2^24)."""
    for code, bits in CODES:
        h = golden["hashes"]["codes"][code]
        w = np.arange(1 << bits, dtype=np.uint32)
        if bits == 24:
            cw, ok = oracle.block_decode(code, w[::16])
            assert sha(cw.astype(np.uint32), ok) == h["stride16_sha256"]
        else:
            cw, ok = oracle.block_decode(code, w)
            assert sha(cw.astype(np.uint32), ok) == h["sha256_all"]
            assert int(ok.sum()) == h["n_ok"]
        for i, (c, k) in enumerate(h["first16"]):
            if bits != 24 or i % 16 == 0:
                pass
        assert [[int(c), int(k)] for c, k in zip(*oracle.block_decode(code, np.arange(16)))] == h["first16"]


def test_oracle_fec_matches_golden_vectors(oracle, golden):
    g = golden["fec"]
    out, ok = oracle.bptc_196_96(g["bptc_in"])
    out[ok == 0] = 0
    assert (out == g["bptc_out"]).all() and (ok == g["bptc_ok"]).all()
    for nd in (100, 180):
        o, m = oracle.trellis(g["trellis%d_in" % nd], nd)
        assert (o == g["trellis%d_out" % nd]).all() and (m == g["trellis%d_metric" % nd]).all()
    for cnt in (4, 10, 20):
        assert (oracle.crc16(g["crc_in"], cnt) == g["crc%d" % cnt]).all()
    for nb in (100, 104, 160):
        assert (oracle.whitening(g["crc_in"], nb)[:, :(nb + 7) // 8] == g["whiten%d" % nb]).all()
    assert (oracle.hamming_distance(g["hd_a"], g["hd_b"]) == g["hd"]).all()


def test_oracle_fec_vs_compiled_reference(oracle):
    """Direct comparison with oracle/_ref (the reference's C sources compiled in place), when present."""
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(7)
    for code, bits in CODES:
        w = rng.integers(0, 1 << bits, 50000)
        a, ao = oracle.block_decode(code, w)
        b, bo = oracle.block_decode(code, w, "ref")
        assert (a == b).all() and (ao == bo).all(), code
    p = rng.integers(0, 256, (4000, 25), dtype=np.uint8)
    o1, k1 = oracle.bptc_196_96(p)
    o2, k2 = oracle.bptc_196_96(p, "ref")
    assert (k1 == k2).all() and (o1[k1 == 1] == o2[k2 == 1]).all()
    for nd, nb in ((100, 25), (180, 45)):
        x = rng.integers(0, 256, (500, nb), dtype=np.uint8)
        o1, m1 = oracle.trellis(x, nd)
        o2, m2 = oracle.trellis(x, nd, "ref")
        assert (o1 == o2).all() and (m1 == m2).all()


def test_oracle_chain_golden(oracle, golden):
    """Drift guard: the committed chain vectors were produced by this oracle (not by the reference)."""
    g = golden["chain"]
    for name, proto in (("dmr_a", 1), ("dmr_b", 1), ("ysf_a", 2)):
        r = oracle.chain(g[name + "_x"][None, :], proto=proto, keep_filtered=True)
        ns, no, ne = int(r["sym_count"][0]), int(r["out_count"][0]), int(r["event_count"][0])
        assert bytes.fromhex(sha(r["filtered"][0])) == g[name + "_filtered_sha256"].tobytes()
        assert (r["syms"][0, :ns] == g[name + "_syms"]).all()
        assert (r["out"][0, :no] == g[name + "_out"]).all()
        assert r["events"][0, :ne].tobytes() == g[name + "_events"].tobytes()


def test_oracle_encoders_round_trip(oracle):
    rng = np.random.default_rng(3)
    for code, k in (("hamming_7_4", 4), ("hamming_13_9", 9), ("hamming_15_11", 11), ("hamming_16_11", 11),
                    ("quadratic_residue", 7), ("golay_20_8", 8), ("golay_24_12", 12)):
        n = dict(CODES)[code]
        t = {"quadratic_residue": 2, "golay_20_8": 3, "golay_24_12": 3}.get(code, 1)
        for _ in range(200):
            info = int(rng.integers(0, 1 << k))
            cw = oracle.encode(code, info)
            assert cw >> (n - k) == info
            err = 0
            for b in rng.choice(n, t, replace=False):
                err |= 1 << int(b)
            fixed, ok = oracle.block_decode(code, [cw ^ err])
            assert ok[0] == 1 and int(fixed[0]) == cw


def test_streaming_oracle_is_chunk_invariant(oracle):
    """The reference modules only look at [ptr, ptr+N): feeding the oracle in pieces must not change anything."""
    from digiham_amd import synth
    s = synth.dmr_stream(5, 20)
    x = synth.impair(synth.shape(s), 5, snr_db=20, delay=3)
    whole = oracle.chain(x[None, :], proto=1)
    f = oracle.Rrc().process(x)
    dem, dec = oracle.Demod(10, 4), oracle.Decoder("dmr")
    syms, out = [], []
    pos = 0
    for c in [1, 7, 300, 4096, 11, 100000]:
        d = dem.process(f[pos:pos + c]); pos += c
        syms.append(d)
        o, _ = dec.process(d)
        out.append(o)
    syms, out = np.concatenate(syms), np.concatenate(out)
    ns, no = int(whole["sym_count"][0]), int(whole["out_count"][0])
    assert len(syms) == ns and (syms == whole["syms"][0, :ns]).all()
    assert len(out) == no and (out == whole["out"][0, :no]).all()
