"""Deterministic synthetic DMR / YSF signal generator (host-side plumbing).

Builds what digiham's ``rrc_filter`` sees in ``examples/dmr-decoder.sh:13-19``:
48 kS/s float32 FM-discriminator audio of a 4800 Bd 4-level FSK signal, i.e.
dibit symbols mapped to levels {1:+3, 0:+1, 2:-1, 3:-3} (the inverse of the
slicer at gfsk_demodulator.cpp:90-104), 10 samples per symbol, TX pulse-shaped
with the same 81-tap root-raised-cosine the receiver uses.

Frame builders follow the layouts the reference decoders parse (file:line of
the *decoding* side is cited at each builder); FEC encoders are systematic
encoders for the ETSI TS 102 361-1 Annex B / YSF generator matrices.

Nothing here touches the GPU; bench.py lifts the symbol streams to the device
and does the pulse shaping there with torch (plumbing, not the product).
"""
import numpy as np

# ----------------------------------------------------------------------------- FEC encoders
# P parts of the systematic generator matrices G = [I | P]
_P = {
    "hamming_7_4": (7, 4, [0x5, 0x7, 0x6, 0x3]),
    "hamming_13_9": (13, 9, [0xF, 0xE, 0x7, 0xA, 0x5, 0xB, 0xC, 0x6, 0x3]),
    "hamming_15_11": (15, 11, [0x9, 0xD, 0xF, 0xE, 0x7, 0xA, 0x5, 0xB, 0xC, 0x6, 0x3]),
    "hamming_16_11": (16, 11, [0x13, 0x1A, 0x1F, 0x1C, 0x0E, 0x15, 0x0B, 0x16, 0x19, 0x0D, 0x07]),
    "golay_24_12": (24, 12, [0xC75, 0x63B, 0xF68, 0x7B4, 0x3DA, 0xD99, 0x6CD, 0x367, 0xDC6, 0xA97, 0x93E, 0x8EB]),
    "golay_20_8": (20, 8, [0x3DA, 0xD99, 0x6CD, 0x367, 0xDC6, 0xA97, 0x93E, 0x8EB]),
    "quadratic_residue": (16, 7, [0x04F, 0x11E, 0x1B7, 0x1E2, 0x1C9, 0x0E5, 0x073]),
}


def block_encode(code, info):
    n, k, p = _P[code]
    info = int(info) & ((1 << k) - 1)
    par = 0
    for j in range(k):
        if (info >> (k - 1 - j)) & 1:
            par ^= p[j]
    return (info << (n - k)) | par


def _bits_of(value, nbits):
    return [(value >> (nbits - 1 - i)) & 1 for i in range(nbits)]


def bytes_to_bits(b):
    return [(int(x) >> (7 - i)) & 1 for x in b for i in range(8)]


def bits_to_bytes(bits):
    bits = list(bits) + [0] * (-len(bits) % 8)
    return bytes(sum(bits[i + j] << (7 - j) for j in range(8)) for i in range(0, len(bits), 8))


def bits_to_dibits(bits):
    assert len(bits) % 2 == 0
    return [bits[i] * 2 + bits[i + 1] for i in range(0, len(bits), 2)]


def bptc_196_96_encode_bits(info_bits):
    """96 info bits -> 196 transmit bits (decoder: bptc_196_96.c:5-59)."""
    assert len(info_bits) == 96
    rows = []
    it = iter(info_bits)
    for r in range(9):
        d = [0, 0, 0] + [next(it) for _ in range(8)] if r == 0 else [next(it) for _ in range(11)]
        rows.append(_bits_of(block_encode("hamming_15_11", int("".join(map(str, d)), 2)), 15))
    rows += [[0] * 15 for _ in range(4)]
    for c in range(15):
        d = int("".join(str(rows[r][c]) for r in range(9)), 2)
        cw = _bits_of(block_encode("hamming_13_9", d), 13)
        for r in range(9, 13):
            rows[r][c] = cw[r]
    deint = [0] + [b for row in rows for b in row]
    tx = [0] * 196
    for i in range(196):
        tx[(i * 181) % 196] = deint[i]
    return tx


def trellis_encode_bits(bits):
    """rate-1/2 K=5, G1 = 1+D^3+D^4, G2 = 1+D+D^2+D^4 (decoder: trellis.c:8-25) -> dibits."""
    state = 0
    out = []
    for b in bits:
        s0, s1, s2, s3 = state & 1, (state >> 1) & 1, (state >> 2) & 1, (state >> 3) & 1
        out.append(((b ^ s1 ^ s0) << 1) | (b ^ s3 ^ s2 ^ s0))
        state = (b << 3) | (state >> 1)
    return out


def crc16_ccitt(data):
    """decoder: crc16.c:3-18 (init 0, poly 0x1021, inverted)."""
    crc = 0
    for byte in data:
        for i in range(8):
            fb = ((byte >> (7 - i)) & 1) ^ ((crc >> 15) & 1)
            crc = (crc << 1) & 0xFFFF
            if fb:
                crc ^= 0x1021
    return crc ^ 0xFFFF


def pn9_bits(n):
    """decoder: whitening.c:6-22."""
    wsr = 0x1C9
    out = []
    for _ in range(n):
        wb = wsr & 1
        out.append(wb)
        fb = ((wsr >> 4) & 1) ^ wb
        wsr = ((wsr & 0x1FE) >> 1) | (fb << 8)
    return out


# ----------------------------------------------------------------------------- DMR
def _hex_to_dibits(h):
    bits = _bits_of(int(h, 16), 4 * len(h))
    return bits_to_dibits(bits)


DMR_SYNC = {  # ETSI TS 102 361-1 table 9.2 (decoder: dmr_phase.hpp:25-28)
    "bs_data": _hex_to_dibits("DFF57D75DF5D"), "bs_voice": _hex_to_dibits("755FD7DF75F7"),
    "ms_data": _hex_to_dibits("D5D7F77FD757"), "ms_voice": _hex_to_dibits("7F7D5DD57DFD"),
}
_TACT_POS = [0, 4, 8, 12, 14, 18, 22]           # decoder: cach.cpp:7


def dmr_cach(slot, lcss=0, at=1, rng=None):
    """12 dibits; TACT = Hamming(7,4) of [AT, TC, LCSS1, LCSS0] (decoder: cach.cpp:11-31, tact.cpp:14-24)."""
    tact = _bits_of(block_encode("hamming_7_4", (at << 3) | (slot << 2) | lcss), 7)
    bits = list(rng.integers(0, 2, 24)) if rng is not None else [0] * 24
    for b, pos in zip(tact, _TACT_POS):
        bits[pos] = b
    return bits_to_dibits([int(b) for b in bits])


def dmr_embedded_lc_fragments(lc9):
    """9 LC bytes -> four 32-bit fragments as 16 dibits each (decoder: embedded.cpp:32-94)."""
    bits = bytes_to_bits(lc9)
    cs = sum(lc9) % 31
    csb = _bits_of(cs, 5)
    rows = [bits[0:11], bits[11:22], bits[22:32] + [csb[0]], bits[32:42] + [csb[1]], bits[42:52] + [csb[2]],
            bits[52:62] + [csb[3]], bits[62:72] + [csb[4]]]
    rows = [_bits_of(block_encode("hamming_16_11", int("".join(map(str, r)), 2)), 16) for r in rows]
    rows.append([sum(r[c] for r in rows) & 1 for c in range(16)])
    stream = [rows[r][c] for c in range(16) for r in range(8)]          # column by column
    return [bits_to_dibits(stream[i * 32:(i + 1) * 32]) for i in range(4)]


def dmr_voice_burst(slot, payload_dibits, mid, rng=None):
    """144 dibits: CACH 12 | 54 | 24 (sync or EMB+embedded) | 54 (decoder: dmr_phase.cpp:207-227)."""
    assert len(payload_dibits) == 108 and len(mid) == 24
    return dmr_cach(slot, rng=rng) + list(payload_dibits[:54]) + list(mid) + list(payload_dibits[54:])


def dmr_emb_mid(cc, lcss, frag16, pi=0):
    """EMB (QR(16,7) of CC|PI|LCSS) split around 16 embedded dibits (decoder: dmr_phase.cpp:117-145, emb.cpp:18-24)."""
    emb = _bits_of(block_encode("quadratic_residue", (cc << 3) | (pi << 2) | lcss), 16)
    return bits_to_dibits(emb[:8]) + list(frag16) + bits_to_dibits(emb[8:])


def dmr_data_burst(slot, cc, data_type, info12, sync="bs_data", rng=None):
    """Data burst: 98 info dibits + slot type Golay(20,8) around the sync (decoder: dmr_phase.cpp:235-284)."""
    tx = bits_to_dibits(bptc_196_96_encode_bits(bytes_to_bits(info12)))
    st = bits_to_dibits(_bits_of(block_encode("golay_20_8", (cc << 4) | data_type), 20))
    return dmr_cach(slot, rng=rng) + tx[:49] + st[:5] + DMR_SYNC[sync] + st[5:] + tx[49:]


def dmr_lc(flco, fid, opts, dst, src):
    """9-byte full LC + 3 parity bytes (parity not checked: lc.cpp:8-11)."""
    return bytes([flco & 0x3F, fid, opts, (dst >> 16) & 255, (dst >> 8) & 255, dst & 255,
                  (src >> 16) & 255, (src >> 8) & 255, src & 255])


def dmr_call(rng, slot, cc=1, dst=1234, src=5678901, n_superframes=4, sync_kind="bs"):
    """One voice call on one slot as a list of 144-dibit bursts: LC header, superframes, terminator."""
    lc = dmr_lc(0, 0, 0, dst, src)
    bursts = [dmr_data_burst(slot, cc, 1, lc + bytes(3), sync_kind + "_data", rng)]
    frags = dmr_embedded_lc_fragments(lc)
    for _ in range(n_superframes):
        for f in range(6):
            payload = list(rng.integers(0, 4, 108))
            if f == 0:
                mid = DMR_SYNC[sync_kind + "_voice"]
            elif f <= 4:
                mid = dmr_emb_mid(cc, [1, 3, 3, 2][f - 1], frags[f - 1])
            else:
                mid = dmr_emb_mid(cc, 0, [0] * 16)
            bursts.append(dmr_voice_burst(slot, payload, mid, rng))
    bursts.append(dmr_data_burst(slot, cc, 2, lc + bytes(3), sync_kind + "_data", rng))
    return bursts


def dmr_idle_burst(slot, cc, rng):
    return dmr_data_burst(slot, cc, 9, bytes(rng.integers(0, 256, 12).tolist()), "bs_data", rng)


def dmr_stream(seed, n_bursts, two_slots=True, cc=1, lead_in=37):
    """A BS-style TDMA dibit stream: bursts alternate slot 0 / slot 1.  Slot 0 carries
    back-to-back voice calls; slot 1 carries idle data bursts or (two_slots) its own calls."""
    rng = np.random.default_rng(seed)
    q = [[], []]
    while len(q[0]) < n_bursts:
        q[0] += dmr_call(rng, 0, cc, dst=int(rng.integers(1, 1 << 24)), src=int(rng.integers(1, 1 << 24)),
                         n_superframes=int(rng.integers(2, 6)))
        q[0] += [dmr_idle_burst(0, cc, rng) for _ in range(int(rng.integers(0, 3)))]
    while len(q[1]) < n_bursts:
        if two_slots:
            q[1] += [dmr_idle_burst(1, cc, rng) for _ in range(int(rng.integers(1, 4)))]
            q[1] += dmr_call(rng, 1, cc, dst=int(rng.integers(1, 1 << 24)), src=int(rng.integers(1, 1 << 24)),
                             n_superframes=int(rng.integers(1, 4)))
        else:
            q[1].append(dmr_idle_burst(1, cc, rng))
    out = list(rng.integers(0, 4, lead_in))
    for i in range(n_bursts):
        out += q[i & 1][i >> 1]
    return np.array(out, np.uint8)


# ----------------------------------------------------------------------------- YSF
YSF_SYNC = _hex_to_dibits("D471C9634D")           # decoder: ysf_phase.hpp:21


def ysf_fich_dibits(frame_type, data_type, frame_number, frame_total=7):
    """FICH word -> CRC16 -> 4x Golay(24,12) -> conv. code -> 5x20 interleave (decoder: fich.cpp:12-66)."""
    word = (frame_type << 30) | (frame_number << 19) | (frame_total << 16) | (data_type << 8)
    be = word.to_bytes(4, "big")
    bits = bytes_to_bits(be) + _bits_of(crc16_ccitt(be), 16)
    coded = []
    for i in range(4):
        coded += _bits_of(block_encode("golay_24_12", int("".join(map(str, bits[i * 12:(i + 1) * 12])), 2)), 24)
    code = trellis_encode_bits(coded + [0, 0, 0, 0])
    tx = [0] * 100
    for i in range(100):
        tx[(i * 20) % 100 + (i * 20) // 100] = code[i]
    return tx


def _ysf_dch_code(data, nbytes):
    """whiten -> CRC16 -> 4 tail bits -> conv. code (decoder: ysf_phase.cpp:258-269 / :332-346)."""
    bits = bytes_to_bits(data[:nbytes])
    pn = pn9_bits(len(bits))
    wh = bits_to_bytes([b ^ p for b, p in zip(bits, pn)])
    bits = bytes_to_bits(wh) + _bits_of(crc16_ccitt(wh), 16) + [0, 0, 0, 0]
    return trellis_encode_bits(bits)


_V2_MAP = [0, 3, 6, 9, 12, 15, 18, 21, 24, 27, 30, 33, 36, 39, 41, 43, 45, 47,
           1, 4, 7, 10, 13, 16, 19, 22, 25, 28, 31, 34, 37, 40, 42, 44, 46, 48,
           2, 5, 8, 11, 14, 17, 20, 23, 26, 29, 32, 35, 38]   # decoder: ysf_phase.hpp:46-51


def ysf_v2_voice_dibits(ambe49):
    """49 AMBE bits -> 52 dibits (decoder: ysf_phase.cpp:180-256)."""
    voice = [ambe49[_V2_MAP[i]] for i in range(49)]
    tri = [b for b in voice[:27] for _ in range(3)] + voice[27:] + [0]
    pn = pn9_bits(104)
    wh = [a ^ b for a, b in zip(tri, pn)]
    inter = [0] * 104
    for k in range(104):
        inter[(k * 4) % 104 + (k * 4) // 104] = wh[k]
    return bits_to_dibits(inter)


def ysf_frame(rng, frame_type, data_type=2, frame_number=0, dch=None, csd=None, corrupt_fich=False):
    """One 480-dibit frame: sync 20 | FICH 100 | payload 360 (decoder: ysf_phase.cpp:45-172)."""
    payload = list(rng.integers(0, 4, 360))
    if frame_type == 1 and data_type == 2:                      # V/D mode type 2
        code = _ysf_dch_code(dch if dch is not None else bytes(rng.integers(32, 127, 10).tolist()), 10)
        for i in range(100):
            payload[(i % 5) * 72 + i // 5] = code[i]
        for blk in range(5):
            payload[blk * 72 + 20: blk * 72 + 72] = ysf_v2_voice_dibits(list(rng.integers(0, 2, 49)))
    elif frame_type in (0, 2):                                  # header / terminator: CSD1 + CSD2
        csd = csd if csd is not None else [bytes(rng.integers(32, 127, 20).tolist()) for _ in range(2)]
        for half in range(2):
            code = _ysf_dch_code(csd[half], 20)
            for i in range(180):
                streampos = (i % 9) * 20 + i // 9
                payload[half * 36 + (streampos // 36) * 72 + streampos % 36] = code[i]
    fich = ysf_fich_dibits(frame_type, data_type, frame_number)
    if corrupt_fich:
        fich = list(rng.integers(0, 4, 100))
    return YSF_SYNC + fich + payload


def ysf_stream(seed, n_frames, mode="vd2", lead_in=53):
    """Header, communication frames (FN cycling 0..7), terminator, repeated."""
    rng = np.random.default_rng(seed)
    dt = {"vd1": 0, "vd2": 2, "fr": 3, "datafr": 1}[mode]
    out = list(rng.integers(0, 4, lead_in))
    n = 0
    while n < n_frames:
        out += ysf_frame(rng, 0, dt); n += 1
        for i in range(int(rng.integers(6, 14))):
            if n >= n_frames:
                break
            out += ysf_frame(rng, 1, dt, i & 7); n += 1
        if n < n_frames:
            out += ysf_frame(rng, 2, dt); n += 1
    return np.array(out, np.uint8)


# ----------------------------------------------------------------------------- NXDN48
NXDN_SYNC = [3, 0, 3, 1, 3, 3, 1, 1, 2, 1]         # decoder: nxdn_phase.cpp:15-16


def nxdn_scramble(dibits):
    """decoder: scrambler.cpp:9-25 (its own inverse): PN output 1 flips the high bit of the dibit."""
    sr = 0x0E4
    out = []
    for d in dibits:
        wb = sr & 1
        out.append((int(d) & 3) ^ (wb << 1))
        wb = ((sr >> 4) & 1) ^ wb
        sr = ((sr & 0x1FE) >> 1) | (wb << 8)
    return out


def nxdn_lich_dibits(lich7):
    """7-bit LICH + parity over its 4 MSBs, one bit per dibit on the high bit (decoder: lich.cpp:5-31)."""
    bits = _bits_of(lich7, 7)
    bits.append(bits[0] ^ bits[1] ^ bits[2] ^ bits[3])
    return [(b << 1) | 1 for b in bits]


def _nxdn_crc(bits, width, init, poly):
    """decoder: sacch.cpp:70-84 (CRC-6) / facch1.cpp:63-75 (CRC-12)."""
    crc = init
    mask = (1 << width) - 1
    for b in bits:
        cb = ((crc >> (width - 1)) & 1) ^ b
        if cb:
            crc ^= poly
        crc = ((crc << 1) & (mask & ~1)) | cb
    return _bits_of(crc, width)


def _nxdn_channel_encode(info_bits, crc_bits, punct, rows, cols):
    bits = list(info_bits) + list(crc_bits) + [0, 0, 0, 0]
    coded = [b for d in trellis_encode_bits(bits) for b in ((d >> 1) & 1, d & 1)]
    kept = [b for i, b in enumerate(coded) if not punct(i)]
    tx = [0] * len(kept)
    for i in range(rows):                       # decoder de-interleaves out[k * rows + i] = in[i * cols + k]
        for k in range(cols):
            tx[i * cols + k] = kept[k * rows + i]
    return bits_to_dibits(tx)


def nxdn_sacch_dibits(structure, ran, data18):
    """One SACCH fragment: 2 bit structure (3 - index), 6 bit RAN, 18 data bits -> 30 dibits (decoder: sacch.cpp:24-84)."""
    info = _bits_of(3 - structure, 2) + _bits_of(ran, 6) + list(data18)
    return _nxdn_channel_encode(info, _nxdn_crc(info, 6, 0x3F, 0x13), lambda i: (i + 1) % 6 == 0, 12, 5)


def nxdn_facch1_dibits(info80):
    """80 information bits -> 72 dibits (decoder: facch1.cpp:8-75)."""
    info = list(info80)
    return _nxdn_channel_encode(info, _nxdn_crc(info, 12, 0xFFF, 0x407), lambda i: (i - 1) % 4 == 0, 16, 9)


def nxdn_vcall_bits(call_type, src, dst):
    """72 bits of a VCALL layer-3 message as the decoder reads them (sacch.cpp:133-152): message type 0x01 in the low
    6 bits of byte 0, call type in the top 3 bits of byte 2, source in bytes 3-4, destination in bytes 5-6."""
    msg = bytes([0x01, 0x00, (call_type & 7) << 5, (src >> 8) & 255, src & 255, (dst >> 8) & 255, dst & 255, 0, 0])
    return bytes_to_bits(msg)


def nxdn_frame(rng, lich7, sacch_dibits, blocks):
    """sync + scrambled (LICH, SACCH, two 72-dibit blocks): 192 dibits (decoder: nxdn_phase.cpp:43-170)."""
    body = nxdn_lich_dibits(lich7) + list(sacch_dibits) + list(blocks[0]) + list(blocks[1])
    assert len(body) == 182
    return NXDN_SYNC + nxdn_scramble(body)


def nxdn_stream(seed, n_frames, lead_in=29, src=None, dst=None):
    """Calls of voice frames (LICH: RDCH, SACCH superframe, both blocks voice) carrying a VCALL in the SACCH superframe,
    opened by a frame whose blocks are FACCH1 (VCALL) and closed by one with TX_RELEASE, noise in between."""
    rng = np.random.default_rng(seed)
    out = list(rng.integers(0, 4, lead_in))
    n = 0
    while n < n_frames:
        s = int(rng.integers(1, 65535)) if src is None else src
        t = int(rng.integers(1, 65535)) if dst is None else dst
        vcall = nxdn_vcall_bits(1 if s & 1 else 4, s, t)
        ran = int(rng.integers(0, 64))
        fa = nxdn_facch1_dibits(vcall + [0] * 8)
        frames = [(0x50, 0, [fa, fa])]
        for i in range(int(rng.integers(9, 17))):
            voice = [list(rng.integers(0, 4, 72)), list(rng.integers(0, 4, 72))]
            frames.append((0x56, i & 3, voice))
        rel = nxdn_facch1_dibits(bytes_to_bits(bytes([0x08])) + [0] * 72)
        frames.append((0x50, 0, [rel, rel]))
        for lich, idx, blocks in frames:
            if n >= n_frames:
                break
            sac = nxdn_sacch_dibits(idx, ran, vcall[18 * idx:18 * idx + 18])
            out += nxdn_frame(rng, lich, sac, blocks)
            n += 1
        out += list(rng.integers(0, 4, int(rng.integers(15, 60))))
    return np.array(out, np.uint8)


# ----------------------------------------------------------------------------- POCSAG
POCSAG_SYNC = 0x7CD215D8          # decoder: pocsag_phase.hpp:15
POCSAG_IDLE = 0x7A89C197          # decoder: codeword.hpp:23


def bch_31_21_encode(data21):
    """Systematic BCH(31,21), generator x^10+x^9+x^8+x^6+x^5+x^3+1 (decoder: bch_31_21.c:3-14)."""
    w = (data21 & 0x1FFFFF) << 10
    r = w
    for j in range(30, 9, -1):
        if r & (1 << j):
            r ^= 0x769 << (j - 10)
    return w | (r & 0x3FF)


def pocsag_codeword(data21):
    """21 information bits -> 32-bit codeword with even parity (decoder: codeword.cpp:9-32)."""
    w = bch_31_21_encode(data21) << 1
    return w | (bin(w).count("1") & 1)


def pocsag_alpha_payloads(text):
    """7-bit characters, LSB first, cut into 20-bit message payloads (decoder: message.cpp:29-38)."""
    bits = [(ord(c) >> k) & 1 for c in text for k in range(7)]
    bits += [0] * (-len(bits) % 20)
    return [int("".join(map(str, bits[i:i + 20])), 2) for i in range(0, len(bits), 20)]


def pocsag_batches(messages):
    """messages: list of (address 21 bit, function, text).  Returns the codewords of as many batches as needed; every
    message starts in the frame its address selects (address & 7), idle codewords fill the rest."""
    words = []
    for address, function, text in messages:
        frame = address & 7
        while (len(words) % 16) // 2 != frame or len(words) % 2:
            words.append(POCSAG_IDLE)
        words.append(pocsag_codeword(((address >> 3) & 0x3FFFF) << 2 | (function & 3)))
        for pl in pocsag_alpha_payloads(text):
            words.append(pocsag_codeword(1 << 20 | pl))
        words.append(POCSAG_IDLE)
    words += [POCSAG_IDLE] * (-len(words) % 16)
    return words


def pocsag_stream(seed, n_messages, lead_in=37):
    """Transmissions of preamble + batches carrying random alphanumeric pages, noise bits in between."""
    rng = np.random.default_rng(seed)
    out = list(rng.integers(0, 2, lead_in))
    sent = []
    n = 0
    while n < n_messages:
        msgs = []
        for _ in range(int(rng.integers(1, 4))):
            if n >= n_messages:
                break
            text = "".join(chr(int(c)) for c in rng.integers(32, 127, int(rng.integers(3, 40))))
            msgs.append((int(rng.integers(8, 1 << 21)), 3, text))
            n += 1
        sent += msgs
        words = pocsag_batches(msgs)
        bits = [1, 0] * 288
        for i in range(0, len(words), 16):
            for w in [POCSAG_SYNC] + words[i:i + 16]:
                bits += _bits_of(w, 32)
        # the decoder keeps its batch grid for three more batches after a transmission ends (pocsag_phase.cpp:40-52), so a
        # transmission that follows sooner is missed: mostly long gaps, sometimes a short one
        gap = int(rng.integers(1750, 1900)) if rng.random() < 0.8 else int(rng.integers(40, 600))
        out += bits + list(rng.integers(0, 2, gap))
    return np.array(out, np.uint8), sent


# ----------------------------------------------------------------------------- D-Star
DSTAR_FRAME_SYNC = [1, 1, 1, 0, 1, 1, 0, 0, 1, 0, 1, 0, 0, 0, 0]                     # JARL D-STAR 2.1.1 frame sync
DSTAR_VOICE_SYNC = [1, 0] * 5 + [1, 1, 0, 1, 0, 0, 0] * 2                             # data-frame sync pattern
DSTAR_TERMINATOR = [1, 0] * 16 + [0, 0, 0, 1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 1, 1, 0]     # end pattern (+ one bit)


def dstar_pn(n):
    """Whitening sequence x^7 + x^4 + 1 from the all-ones state."""
    sr = 0x7F
    out = np.zeros(n, np.uint8)
    for i in range(n):
        wb = (sr & 1) ^ ((sr >> 3) & 1)
        out[i] = wb
        sr = ((sr & 0x7E) >> 1) | (wb << 6)
    return out


def dstar_crc(data):
    """CRC-CCITT, reflected (0x8408), preset and inverted: the radio header's P_FCS."""
    c = 0xFFFF
    for b in bytes(data):
        for i in range(8):
            c ^= (b >> i) & 1
            c = (c >> 1) ^ 0x8408 if c & 1 else c >> 1
    return c ^ 0xFFFF


def dstar_header_bytes(rpt2, rpt1, your, my, suffix="", flags=(0, 0, 0)):
    """The 41-byte radio header: 3 flag bytes, destination / departure repeater, companion, own callsign + suffix, FCS."""
    f = lambda t, n: t.encode("latin-1")[:n].ljust(n, b" ")
    h = bytes(flags) + f(rpt2, 8) + f(rpt1, 8) + f(your, 8) + f(my, 8) + f(suffix, 4)
    c = dstar_crc(h)
    return h + bytes([c & 0xFF, c >> 8])


def dstar_header_bits(h41):
    """Header bytes -> the 660 transmitted bits: rate-1/2 K=3 convolutional code (bits LSB first, two tail bits),
    24-column interleave, whitening."""
    bits = np.concatenate([np.unpackbits(np.frombuffer(bytes(h41), np.uint8), bitorder="little"), [0, 0]]).astype(np.uint8)
    coded = np.zeros(660, np.uint8)
    b1 = b2 = 0
    for n, b in enumerate(bits):
        coded[2 * n] = b ^ b1 ^ b2
        coded[2 * n + 1] = b ^ b2
        b2, b1 = b1, int(b)
    tx = np.zeros(660, np.uint8)
    for i in range(12):
        for k in range(28):
            tx[i * 28 + k] = coded[k * 24 + i]
    for i in range(12, 24):
        for k in range(27):
            tx[12 + i * 27 + k] = coded[k * 24 + i]
    return tx ^ dstar_pn(660)


def dstar_slow_blocks(message=None, header=None, simple=b""):
    """6-byte slow-data blocks (mini header + 5 bytes): the 20-character message (0x40-0x43), the header copy (0x55 / 0x51),
    simple data such as DPRS / NMEA sentences (0x3n)."""
    blocks = []
    if message is not None:
        m = message.encode("latin-1")[:20].ljust(20, b" ")
        blocks += [bytes([0x40 + i]) + m[5 * i:5 * i + 5] for i in range(4)]
    if header is not None:
        for i in range(0, 41, 5):
            part = bytes(header[i:i + 5])
            blocks.append(bytes([0x50 + len(part)]) + part.ljust(5, b"\x66"))
    for i in range(0, len(simple), 5):
        part = bytes(simple[i:i + 5])
        blocks.append(bytes([0x30 + len(part)]) + part.ljust(5, b"\x66"))
    return blocks


def dstar_dprs_sentence(text):
    body = (text + "\r").encode("latin-1")
    return b"$$CRC%04X," % dstar_crc(body) + body


def dstar_gga_sentence(lat, lon):
    """$GPGGA with the given position (degrees, positive north / east)."""
    def dm(v, w):
        a = abs(v); d = int(a); m = (a - d) * 60
        return "%0*d%07.4f" % (w, d, m)
    body = "GPGGA,123519,%s,%s,%s,%s,1,08,0.9,545.4,M,46.9,M,," % (dm(lat, 2), "N" if lat >= 0 else "S", dm(lon, 3), "E" if lon >= 0 else "W")
    cs = 0
    for ch in body.encode():
        cs ^= ch
    return ("$%s*%02X\r\n" % (body, cs)).encode()


def dstar_voice_frames(rng, superframes, trailing_sync=True):
    """Voice superframes of 21 frames (72 random AMBE bits + a 24-bit data frame): a sync frame, then 20 whitened
    slow-data frames carrying that superframe's (up to 10) blocks, filler 0x66 after them.  The receiver evaluates a
    superframe's slow data at the NEXT sync frame, hence the trailing one.  Returns (bits, voice bytes)."""
    pn24 = dstar_pn(24)
    bits, voice = [], []
    for sf, blocks in enumerate(list(superframes) + ([None] if trailing_sync else [])):
        queue = list(blocks or [])
        for f in range(21 if blocks is not None else 1):
            v = rng.integers(0, 2, 72).astype(np.uint8)
            voice.append(np.packbits(v, bitorder="little"))
            bits.append(v)
            if f == 0:
                bits.append(np.array(DSTAR_VOICE_SYNC, np.uint8))
                continue
            if f % 2 == 1:
                cur = queue.pop(0) if queue else b"\x66" * 6
            half = cur[:3] if f % 2 == 1 else cur[3:]
            bits.append(np.unpackbits(np.frombuffer(half, np.uint8), bitorder="little") ^ pn24)
    return np.concatenate(bits), voice


def dstar_transmission(rng, my="DL1ABC", your="CQCQCQ", rpt1="DB0XYZ B", rpt2="DB0XYZ G", suffix="ID51", message="hello d-star world",
                       n_superframes=3, simple=b"", with_header=True, inline_header=True, data_flag=False):
    """Bit sync + frame sync + radio header, then superframes that alternate message (+ simple data) and the header copy,
    then the end pattern in place of the last data frame."""
    h = dstar_header_bytes(rpt2, rpt1, your, my, suffix, flags=(0x80 if data_flag else 0, 0, 0))
    bits = []
    if with_header:
        bits += [np.array([1, 0] * 32 + DSTAR_FRAME_SYNC, np.uint8), dstar_header_bits(h)]
    pending = dstar_slow_blocks(None, None, simple)
    sfs = []
    for i in range(n_superframes):
        if i % 2 == 1 and inline_header:
            sfs.append(dstar_slow_blocks(None, h))
        else:
            blocks = dstar_slow_blocks(message) + pending[:6]
            pending = pending[6:]
            sfs.append(blocks)
    vb, voice = dstar_voice_frames(rng, sfs)
    if not with_header:
        vb = vb[72:]                                                   # late entry: the receiver first sees a data-frame sync
        voice = voice[1:]
    last = rng.integers(0, 2, 72).astype(np.uint8)
    voice.append(np.packbits(last, bitorder="little"))
    bits += [vb, last, np.array(DSTAR_TERMINATOR, np.uint8)]
    return np.concatenate(bits), h, voice


def dstar_stream(seed, n_transmissions, lead_in=41, ber=0.0):
    """Transmissions (header + voice superframes with slow data + terminator) separated by noise bits; some enter late
    (no header), some carry a DPRS / NMEA position, some have a damaged or a data header.  Returns (bits, infos)."""
    rng = np.random.default_rng(seed)
    out = [rng.integers(0, 2, lead_in).astype(np.uint8)]
    infos = []
    for t in range(n_transmissions):
        kind = int(rng.integers(0, 8))
        my = "".join(chr(int(c)) for c in rng.integers(65, 91, 6))
        msg = "".join(chr(int(c)) for c in rng.integers(32, 127, int(rng.integers(5, 21))))
        simple = b""
        if kind in (1, 5):
            simple = dstar_dprs_sentence("%s>APDPRS,DSTAR*:!4916.45N/01131.00E>test %d" % (my, t))
        elif kind == 2:
            simple = dstar_gga_sentence(float(rng.uniform(-80, 80)), float(rng.uniform(-170, 170)))
        nsf = int(rng.integers(1, 5))
        if simple:
            nsf = max(nsf, 2 * (-(-len(simple) // 30)) - 1)            # 30 simple-data bytes per message superframe
        b, h, voice = dstar_transmission(rng, my=my, message=msg, n_superframes=nsf, simple=simple,
                                         with_header=kind != 3, inline_header=kind != 4, data_flag=kind == 6)
        if kind == 7:                                                  # a header beyond repair: 40 errors in its 660 bits
            at = 64 + 15 + rng.choice(660, 40, replace=False)
            b = b.copy(); b[at] ^= 1
        infos.append({"kind": kind, "my": my, "message": msg, "header": h, "simple": simple, "frames": len(voice)})
        out += [b, rng.integers(0, 2, int(rng.integers(30, 400))).astype(np.uint8)]
    bits = np.concatenate(out)
    if ber > 0:
        bits = bits ^ (rng.random(bits.size) < ber).astype(np.uint8)
    return bits, infos


def fsk_shape(bits, sps=40, amplitude=0.4, invert=False):
    """Two-level FSK discriminator audio: bit 1 above the centre (below with `invert`, as POCSAG is received:
    examples/pocsag-decoder.sh, fsk_demodulator -i), lightly low-passed edges."""
    lv = np.where(np.asarray(bits) > 0, 1.0, -1.0) * (-1.0 if invert else 1.0)
    x = np.repeat(lv, sps)
    w = max(sps // 8, min(5, sps // 2), 1)              # without sloped edges every sampling phase looks alike to the slicer
    k = np.ones(w) / w
    return (amplitude * np.convolve(x, k, mode="same")).astype(np.float32)


# ----------------------------------------------------------------------------- waveform
LEVELS = np.array([1.0, 3.0, -1.0, -3.0], np.float32) / 3.0      # dibit 0,1,2,3 (gfsk_demodulator.cpp:90-104)


def wide_rrc_taps():
    from . import _taps
    return _taps.wide()


def shape(symbols, sps=10, taps=None, amplitude=0.5, circular=False, levels=LEVELS):
    """Dibits -> pulse-shaped float32 audio (TX RRC), `sps` samples per symbol."""
    taps = wide_rrc_taps() if taps is None else taps
    imp = np.zeros(len(symbols) * sps, np.float64)
    imp[::sps] = levels[np.asarray(symbols)]
    g = taps.astype(np.float64)
    g = g / g.sum() * sps                                          # unity gain for a constant symbol run
    if circular:
        n = len(imp)
        y = np.real(np.fft.ifft(np.fft.fft(imp) * np.fft.fft(g, n)))
        y = np.roll(y, -(len(g) // 2))
    else:
        y = np.convolve(imp, g)[len(g) // 2:][:len(imp)]
    return (amplitude * y).astype(np.float32)


def impair(x, seed, snr_db=None, dc=0.0, gain=1.0, delay=0):
    """Per-channel impairments: sample delay, gain, DC offset, additive white noise."""
    rng = np.random.default_rng(seed)
    y = np.roll(x, delay).astype(np.float32) * np.float32(gain) + np.float32(dc)
    if snr_db is not None:
        p = float(np.mean(x.astype(np.float64) ** 2))
        sigma = np.sqrt(p / (10 ** (snr_db / 10)))
        y = y + rng.normal(0, sigma, len(y)).astype(np.float32)
    return y.astype(np.float32)
