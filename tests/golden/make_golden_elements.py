"""Generate tests/golden/elements_ref.{npz,json} (run in the development container only).

Every expected value comes from the REFERENCE's own burst / frame element classes, compiled where they lie under
/root/reference into oracle/_ref/libdigiham_ref_{dmr,ysf,pocsag,dstar}.so (oracle/Makefile target `ref`; glue
oracle/ref_{dmr,ysf,pocsag,dstar}.cpp) -- none of those sources needs csdr:

    DMR     Cach::parse + Tact (all 2^24 CACHs), Emb::parse (all 2^16 words), SlotType::parse (all 2^20 words):
            SHA-256 of the complete output tables + their first rows;
            EmbeddedCollector::collect/getLc, Lc getters, Gps::parse, TalkerAliasCollector
    YSF     Fich::parse, Gps::parse, DataCollector / DataFrame
    POCSAG  Codeword::parse + getters
    D-Star  Header::parseFromHeader + toString()

Inputs are built with the independent encoders of digiham_amd/synth.py, with bit errors, plus random words.

    python tests/golden/make_golden_elements.py
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


def dibits_to_bytes(d):
    d = np.asarray(d, np.uint8).reshape(-1, 4)
    return (d[:, 0] << 6 | d[:, 1] << 4 | d[:, 2] << 2 | d[:, 3]).astype(np.uint8)


def flip_bits(a, rng, nflips, nbits_per_item):
    """flip nflips distinct bits of a flat uint8 array whose items carry nbits_per_item bits each (MSB first)"""
    a = np.array(a, np.uint8)
    total = a.size * nbits_per_item
    for bp in rng.choice(total, nflips, replace=False):
        a.flat[bp // nbits_per_item] ^= 1 << (nbits_per_item - 1 - bp % nbits_per_item)
    return a


def hashes(E):
    h = {}
    s = hashlib.sha256()
    for start in range(0, 1 << 24, 1 << 20):
        s.update(E.dmr_cach(O.all_cach_dibits(start, 1 << 20)).tobytes())
    h["dmr_cach_all_2^24"] = s.hexdigest()
    o, c = E.dmr_emb(np.arange(1 << 16))
    h["dmr_emb_all_2^16"] = hashlib.sha256(o.tobytes() + c.tobytes()).hexdigest()
    h["dmr_emb_ok_count"] = int(o[:, 0].sum())
    o, c = E.dmr_slottype(np.arange(1 << 20))
    h["dmr_slottype_all_2^20"] = hashlib.sha256(o.tobytes() + c.tobytes()).hexdigest()
    h["dmr_slottype_ok_count"] = int(o[:, 0].sum())
    return h


def main():
    E = O.Elements("ref")
    for name in ("dmr", "ysf", "pocsag", "dstar"):
        assert O.ref_lib(name) is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20260929)
    v = {}
    h = hashes(E)

    # ---- DMR: first rows of the exhaustive tables + random samples of them
    v["cach_first"] = E.dmr_cach(O.all_cach_dibits(0, 4096))
    idx = rng.integers(0, 1 << 24, 4096).astype(np.uint32)
    v["cach_sample_idx"] = idx
    v["cach_sample"] = E.dmr_cach(np.concatenate([O.all_cach_dibits(int(i), 1) for i in idx]))
    o, c = E.dmr_emb(np.arange(1 << 16))
    v["emb_ok_bits"] = np.packbits(o[:, 0])
    v["emb_first_out"], v["emb_first_cor"] = o[:1024], c[:1024]
    o, c = E.dmr_slottype(np.arange(1 << 20))
    v["slottype_ok_bits"] = np.packbits(o[:, 0])
    v["slottype_first_out"], v["slottype_first_cor"] = o[:1024], c[:1024]

    # ---- DMR embedded LC: encoded (+ bit errors) / three fragments with the stale fourth / short and long / random
    prev, frags, nfr = [], [], []
    for i in range(6144):
        lc9 = bytes(rng.integers(0, 256, 9).tolist())
        f = np.concatenate([dibits_to_bytes(x) for x in synth.dmr_embedded_lc_fragments(lc9)])   # 16 bytes
        p = rng.integers(0, 256, 16).astype(np.uint8)
        fr = np.zeros(20, np.uint8)
        if i < 4096:
            fr[:16] = flip_bits(f, rng, i % 8, 8); n = 4
        elif i < 5120:
            fr[:12] = f[:12]; n = 3
            if i % 2 == 0:
                p[12:] = f[12:]                       # the buffer still holds a matching last quarter (embedded.cpp:33)
            if i % 5 == 0:
                fr[:12] = flip_bits(fr[:12], rng, 1, 8)
        elif i < 5632:
            fr[:16] = f; fr[16:] = rng.integers(0, 256, 4); n = (0, 1, 2, 5)[i % 4]
        else:
            fr[:] = rng.integers(0, 256, 20); n = 4
        prev.append(p); frags.append(fr); nfr.append(n)
    v["elc_prev"], v["elc_frags"], v["elc_nfrags"] = np.array(prev, np.uint8), np.array(frags, np.uint8), np.array(nfr, np.uint8)
    v["elc_out"] = E.dmr_embedded_lc(v["elc_prev"], v["elc_frags"], v["elc_nfrags"])

    # ---- DMR LC getters, GPS
    v["lc_in"] = rng.integers(0, 256, (1024, 9)).astype(np.uint8)
    v["lc_fields"], v["lc_data7"] = E.dmr_lc(v["lc_in"])
    g = rng.integers(0, 256, (2048, 7)).astype(np.uint8)
    g[0] = 0; g[1] = 255; g[2] = [0, 0, 0, 0, 0x80, 0, 0]; g[3] = [1, 0, 0, 0, 0, 0, 0]
    v["dmr_gps_in"], v["dmr_gps_out"] = g, E.dmr_gps(g)

    # ---- DMR talker alias: four formats, every length, complete / partial / out-of-order block sequences
    orders = [(0, 1, 2, 3), (0, 1, 9, 9), (0, 9, 9, 9), (1, 2, 3, 0), (3, 2, 1, 0), (1, 9, 9, 9), (0, 2, 3, 1), (0, 1, 2, 9), (2, 0, 1, 3)]
    blocks, order = [], []
    for i in range(3072):
        fmt, length = i % 4, (i // 4) % 32
        b = np.zeros(28, np.uint8)
        if fmt == 0:                                   # 7 bit: any bit pattern is a valid 7-bit string
            b[:] = rng.integers(0, 256, 28)
        elif fmt == 1:                                 # 8 bit (ISO-8859-1 through ICU); now and then an embedded NUL
            b[1:] = rng.integers(0x20, 0x100, 27)
            if i % 16 == 1:
                b[1 + int(rng.integers(0, 27))] = 0
        elif fmt == 2:                                 # UTF-8 bytes, copied verbatim
            txt = "".join(chr(int(c)) for c in rng.choice([0x41, 0x62, 0x33, 0x20, 0xE4, 0xF6, 0x20AC, 0x3042], 27)).encode("utf-8")[:27]
            b[1:1 + len(txt)] = np.frombuffer(txt, np.uint8)
        else:                                          # UTF-16BE, no surrogates (std::wstring_convert would throw)
            cp = rng.choice([0x41, 0x7A, 0x30, 0xE9, 0x3A9, 0x4E2D, 0x20AC, 0x20], 13)
            for k, c in enumerate(cp):
                b[1 + 2 * k], b[2 + 2 * k] = int(c) >> 8, int(c) & 255
            b[27] = rng.integers(0, 256)
        b[0] = (fmt << 6) | (length << 1) | int(rng.integers(0, 2))
        blocks.append(b); order.append(orders[(i // 128) % len(orders)])
    v["ta_blocks"], v["ta_order"] = np.array(blocks, np.uint8), np.array(order, np.uint8)
    v["ta_complete"], v["ta_text"], v["ta_len"] = E.dmr_talkeralias(v["ta_blocks"], v["ta_order"])
    assert (v["ta_len"] != 255).all() and (v["ta_complete"] < 16).all()

    # ---- YSF FICH: encoded with 0..11 bit errors, random dibits
    fich = []
    for i in range(2560):
        d = np.array(synth.ysf_fich_dibits(int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 8)), int(rng.integers(0, 8))), np.uint8)
        fich.append(flip_bits(d, rng, i % 12, 2))
    fich += list(rng.integers(0, 4, (512, 100)).astype(np.uint8))
    v["fich_in"] = np.array(fich, np.uint8)
    v["fich_out"], v["fich_data"] = E.ysf_fich(v["fich_in"])

    # ---- YSF GPS: plausible position bytes, every branch of gps.cpp:5-82 (rows keep (d[4] & 0xF0) in {0x50, 0x30}, see ref_ysf.cpp)
    d = np.zeros((6144, 9), np.uint8)
    for i in range(len(d)):
        low = rng.integers(0, 10, 6) if i % 11 else rng.integers(0, 16, 6)
        d[i, :6] = low
        d[i, 0] |= rng.integers(0, 16) << 4; d[i, 1] |= rng.integers(0, 16) << 4; d[i, 2] |= rng.integers(0, 16) << 4
        d[i, 3] |= (0x50, 0x30, 0x50, 0x30, 0x40)[i % 5]
        d[i, 4] |= (0x50, 0x30)[(i // 5) % 2]
        d[i, 5] |= (0x50, 0x30, 0x30, 0x50, 0x50, 0x30, 0x70)[i % 7]
        d[i, 6] = rng.integers(0x20, 0x84)
        d[i, 7] = rng.integers(0x20, 0x68)
        d[i, 8] = rng.integers(0x16, 0x84)
    v["ysf_gps_in"] = d
    v["ysf_gps_ok"], v["ysf_gps_out"] = E.ysf_gps(d)

    # ---- YSF DataCollector / DataFrame scripts
    chunks, offs = np.zeros((1536, 80), np.uint8), np.full((1536, 8), 9, np.uint8)
    scripts = [(0, 1), (0, 1, 0, 1), (1, 0, 1), (0, 0, 1), (0,), (1,), (0, 1, 1), (1, 1, 0, 1), (0, 1, 0)]
    for i in range(len(chunks)):
        sc = scripts[i % len(scripts)]
        offs[i, :len(sc)] = sc
        chunks[i] = rng.integers(0, 256, 80)
        frame = np.zeros(20, np.uint8)
        frame[:18] = rng.integers(0, 256, 18)
        frame[1:4] = [(0x22, 0x62, 0x5f), (0x22, 0x61, 0x5f), (0x47, 0x63, 0x5f), (0x47, 0x64, 0x5f), (1, 2, 3)][i % 5 if i % 3 else 0]
        frame[4] = rng.choice([0x24, 0x25, 0x26, 0x28, 0x29, 0x2a, 0x2b, 0x2d, 0x2e, 0x30, 0x31, 0x32, 0x33, 0x34, 0x35, 0x00, 0x7f])
        frame[5:14] = v["ysf_gps_in"][rng.integers(0, len(d))]
        frame[18] = 0x03 if i % 13 else 0x02
        frame[19] = (int(frame[:19].sum()) + (0 if i % 7 else 1)) & 255
        # the last (0, 1) pair of the script carries the frame
        last0 = max((k for k in range(len(sc) - 1) if sc[k] == 0 and sc[k + 1] == 1), default=None)
        if last0 is not None:
            chunks[i, 10 * last0:10 * last0 + 10] = frame[:10]
            chunks[i, 10 * last0 + 10:10 * last0 + 20] = frame[10:]
    v["ydata_chunks"], v["ydata_offsets"] = chunks, offs
    v["ydata_has2"], v["ydata_frame"], v["ydata_radio"], v["ydata_latlon"] = E.ysf_data(chunks, offs)
    assert (v["ydata_frame"][:, 3] == 0).all()

    # ---- POCSAG codewords: encoded with 0..3 bit errors (any non-zero byte is a 1), random, idle
    bits = []
    for i in range(3072):
        w = synth.pocsag_codeword(int(rng.integers(0, 1 << 21))) if i % 9 else synth.POCSAG_IDLE
        b = np.array(synth._bits_of(w, 32), np.uint8)
        b = flip_bits(b, rng, i % 4, 1)
        if i % 3 == 0:
            b = b * rng.choice([1, 2, 3, 128, 255], 32).astype(np.uint8)
        bits.append(b)
    bits += list(rng.integers(0, 2, (1024, 32)).astype(np.uint8))
    v["cw_in"] = np.array(bits, np.uint8)
    v["cw_out"], v["cw_words"] = E.pocsag_codeword(v["cw_in"])

    # ---- D-Star radio headers: encoded with 0..23 bit errors (Viterbi limit: 10 errors, header.cpp:36), random bits
    raw = []
    alphabet = [chr(c) for c in range(0x30, 0x5B)] + [" ", " ", "/", "\xe4", "\xd6", "\xfc"]
    cs = lambda n: "".join(rng.choice(alphabet, n))
    for i in range(768):
        h41 = bytearray(synth.dstar_header_bytes(cs(8), cs(8), cs(8), cs(int(rng.integers(3, 9))), cs(4) if i % 3 else "", flags=tuple(int(x) for x in rng.integers(0, 256, 3))))
        if i % 37 == 5:                                # an embedded NUL: Converter::convertToUtf8 cuts the field there (charset.cpp:22)
            h41[27 + int(rng.integers(0, 8))] = 0
            c = synth.dstar_crc(bytes(h41[:39])); h41[39], h41[40] = c & 255, c >> 8
        if i % 41 == 7:                                # bad FCS
            h41[40] ^= 0x10
        raw.append(flip_bits(synth.dstar_header_bits(bytes(h41)), rng, i % 24, 1))
    raw += list(rng.integers(0, 2, (128, 660)).astype(np.uint8))
    raw = np.array(raw, np.uint8)
    v["dh_in_bits"] = np.packbits(raw, axis=1)                          # 660 bits -> 83 bytes
    v["dh_ok"], v["dh_data"], v["dh_text"] = E.dstar_header(raw)

    np.savez_compressed(os.path.join(OUT, "elements_ref.npz"), **v)
    with open(os.path.join(OUT, "elements_ref_hashes.json"), "w") as f:
        json.dump(h, f, indent=1, sort_keys=True)
    print(json.dumps(h, indent=1))
    print("embedded LC ok %d / %d" % (v["elc_out"][:, 0].sum(), len(v["elc_out"])))
    print("talker alias complete(any) %d / %d, non-empty %d" % ((v["ta_complete"] != 0).sum(), len(v["ta_len"]), (v["ta_len"] > 0).sum()))
    print("fich ok %d / %d" % (v["fich_out"][:, 0].sum(), len(v["fich_in"])))
    print("ysf gps ok %d / %d" % (v["ysf_gps_ok"].sum(), len(d)))
    print("ysf data frames %d, with gps %d / %d" % (v["ydata_frame"][:, 0].sum(), v["ydata_frame"][:, 2].sum(), len(chunks)))
    print("pocsag codewords ok %d / %d" % (v["cw_out"][:, 0].sum(), len(v["cw_in"])))
    print("dstar headers ok %d / %d" % (v["dh_ok"].sum(), len(raw)))
    print("npz bytes:", os.path.getsize(os.path.join(OUT, "elements_ref.npz")))


if __name__ == "__main__":
    main()
