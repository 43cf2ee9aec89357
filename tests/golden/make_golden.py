"""Generate the committed golden vectors (run in the development container only).

FEC vectors come from the REFERENCE's own C code (oracle/_ref/libdigiham_ref_fec.so, built by
oracle/Makefile from /root/reference/src/{dmr_decoder,ysf_decoder,lib}/*.c, unmodified).
Chain-level vectors (signal -> dibits -> frames) come from the oracle restatement, because the
reference's C++ stages cannot be built here (csdr absent); they pin the oracle against drift,
not against the reference, and are labelled as such in the file.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from digiham_amd import synth           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    assert O.ref() is not None, "build oracle/_ref first (make -C oracle)"
    hashes = {"source": "reference C code via oracle/_ref/libdigiham_ref_fec.so", "codes": {}}
    for code, bits in [("hamming_7_4", 7), ("hamming_13_9", 13), ("hamming_15_11", 15), ("hamming_16_11", 16),
                       ("quadratic_residue", 16), ("golay_20_8", 20), ("golay_24_12", 24)]:
        w = np.arange(1 << bits, dtype=np.uint32)
        cw, ok = O.block_decode(code, w, "ref")
        entry = {"bits": bits, "sha256_all": sha(cw.astype(np.uint32), ok), "n_ok": int(ok.sum()),
                 "first16": [[int(c), int(k)] for c, k in zip(cw[:16], ok[:16])]}
        if bits == 24:      # strided subset for the CPU tier (the GPU tier checks all 2^24)
            entry["stride16_sha256"] = sha(cw[::16].astype(np.uint32), ok[::16])
        hashes["codes"][code] = entry
        print(code, entry["n_ok"], entry["sha256_all"][:16])

    rng = np.random.default_rng(20260928)
    vec = {}
    # BPTC: valid codewords with 0..7 flipped bits + random payloads
    info = rng.integers(0, 256, (1536, 12), dtype=np.uint8)
    cw = np.stack([O.bptc_encode(i) for i in info])
    for r in range(len(cw)):
        for bp in rng.choice(196, r % 8, replace=False):
            cw[r, bp // 8] ^= 0x80 >> (bp % 8)
    payload = np.concatenate([cw, rng.integers(0, 256, (512, 25), dtype=np.uint8)])
    out, ok = O.bptc_196_96(payload, "ref")
    out[ok == 0] = 0
    vec["bptc_in"], vec["bptc_out"], vec["bptc_ok"] = payload, out, ok
    # Viterbi: half encoded + noise, half random
    for nd in (100, 180):
        nb = (nd + 3) // 4
        bits = rng.integers(0, 256, (512, (nd + 7) // 8), dtype=np.uint8)
        enc = np.stack([O.trellis_encode(b, nd) for b in bits])
        for r in range(len(enc)):
            for bp in rng.choice(nd * 2, r % 24, replace=False):
                enc[r, bp // 8] ^= 0x80 >> (bp % 8)
        x = np.concatenate([enc, rng.integers(0, 256, (512, nb), dtype=np.uint8)])
        o, m = O.trellis(x, nd, "ref")
        vec["trellis%d_in" % nd], vec["trellis%d_out" % nd], vec["trellis%d_metric" % nd] = x, o, m
    d = rng.integers(0, 256, (256, 20), dtype=np.uint8)
    vec["crc_in"] = d
    for cnt in (4, 10, 20):
        vec["crc%d" % cnt] = O.crc16(d, cnt, "ref")
    for nb in (100, 104, 160):
        w = O.whitening(d, nb, "ref")
        vec["whiten%d" % nb] = w[:, :(nb + 7) // 8]
    a = rng.integers(0, 4, (256, 24), dtype=np.uint8)
    b = rng.integers(0, 4, (256, 24), dtype=np.uint8)
    vec["hd_a"], vec["hd_b"], vec["hd"] = a, b, O.hamming_distance(a, b, "ref")
    np.savez_compressed(os.path.join(OUT, "fec_ref.npz"), **vec)
    json.dump(hashes, open(os.path.join(OUT, "fec_ref_hashes.json"), "w"), indent=1)

    # chain-level (oracle-generated: drift guard only)
    chain = {}
    for name, proto, seed, kw in [("dmr_a", "dmr", 101, dict(snr_db=20, dc=0.1, delay=4, gain=0.7)),
                                  ("dmr_b", "dmr", 102, dict(snr_db=None, dc=-0.9, delay=0, gain=1.0)),
                                  ("ysf_a", "ysf", 103, dict(snr_db=16, dc=0.0, delay=9, gain=2.0))]:
        s = synth.dmr_stream(seed, 30, two_slots=True) if proto == "dmr" else synth.ysf_stream(seed, 9)
        x = synth.impair(synth.shape(s), seed, **kw)
        r = O.chain(x[None, :], proto=1 if proto == "dmr" else 2, keep_filtered=True)
        ns, no, ne = int(r["sym_count"][0]), int(r["out_count"][0]), int(r["event_count"][0])
        chain[name + "_tx"] = s
        chain[name + "_x"] = x
        chain[name + "_filtered_sha256"] = np.frombuffer(bytes.fromhex(sha(r["filtered"][0])), np.uint8)
        chain[name + "_syms"] = r["syms"][0, :ns]
        chain[name + "_out"] = r["out"][0, :no]
        chain[name + "_events"] = r["events"][0, :ne].view(np.uint8).reshape(ne, 32)
        print(name, len(x), ns, no, ne)
    np.savez_compressed(os.path.join(OUT, "chain_oracle.npz"), **chain)


if __name__ == "__main__":
    main()
