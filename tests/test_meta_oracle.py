"""the metadata LINES of the product's collectors (include/digiham/*_meta.hpp, fed with the
product's decoder events) against oracle/meta.py -- an independent restatement of the reference's MetaCollector logic --
fed with the oracle decoder's events, on a few hundred randomised symbol streams per run: calls with talker aliases in all
four formats, GPS, terminators, two slots, TACT-forced slot switches, symbol errors up to sync loss; YSF headers / V/D2
data channels with and without GPS frames / terminators; NXDN calls.  Both tiers."""
import os
import subprocess

import numpy as np
import pytest

from digiham_amd import api, synth
from oracle import meta as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def meta_exe(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("meta") / "meta_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "host_cpp", "meta_test.cpp"), "-o", exe], check=True)
    return exe


def _alias_lcs(rng):
    """LCs of a talker alias: header + blocks, one of the four formats, sometimes out of order / incomplete / padded with NULs"""
    fmt = int(rng.integers(0, 4))
    if fmt == 0:
        text = bytes(rng.integers(0x20, 0x7F, int(rng.integers(1, 31))).astype(np.uint8))
        bits = []
        for ch in text:
            bits += [(ch >> (6 - i)) & 1 for i in range(7)]
        length = len(text)
        payload_bits = bits
    elif fmt == 1:
        text = bytes(rng.integers(0x20, 0x100, int(rng.integers(1, 27))).astype(np.uint8))
        if rng.integers(0, 3) == 0:
            text = text[:len(text) // 2] + bytes(len(text) - len(text) // 2)        # NUL padding (YSF bridges do that)
        length, payload_bits = len(text), [b for ch in text for b in [(ch >> (7 - i)) & 1 for i in range(8)]]
    elif fmt == 2:
        s = "".join(chr(int(c)) for c in rng.choice([0x41, 0x62, 0xE4, 0x20AC, 0x4E2D, 0x30], int(rng.integers(1, 12))))
        text = s.encode("utf-8")[:27]
        length, payload_bits = min(31, len(text) + int(rng.integers(0, 2))), [b for ch in text for b in [(ch >> (7 - i)) & 1 for i in range(8)]]
    else:
        s = "".join(chr(int(c)) for c in rng.choice([0x41, 0x62, 0xE4, 0x20AC, 0x4E2D], int(rng.integers(1, 13))))
        text = s.encode("utf-16-be")
        length, payload_bits = len(s), [b for ch in text for b in [(ch >> (7 - i)) & 1 for i in range(8)]]
    head_bits = [(fmt >> 1) & 1, fmt & 1] + [(length >> (4 - i)) & 1 for i in range(5)]
    if fmt == 0:
        allbits = head_bits + payload_bits                       # 7-bit characters follow the seven header bits directly
    else:
        allbits = head_bits + [0] + payload_bits                 # one reserved bit, then whole bytes
    allbits = (allbits + [0] * 224)[:224]
    data = bytes(int("".join(map(str, allbits[i:i + 8])), 2) for i in range(0, 224, 8))
    nblocks = 1 + min(3, max(0, (len(payload_bits) + (7 if fmt == 0 else 8) + 55) // 56 - 1))
    order = list(range(nblocks))
    if rng.integers(0, 5) == 0:
        rng.shuffle(order)
    if rng.integers(0, 6) == 0 and nblocks > 1:
        order = order[:-1]
    return [bytes([4 + b, 0]) + data[7 * b:7 * b + 7] for b in order]


def _dmr_stream(rng):
    cc = int(rng.integers(0, 16))
    out = list(rng.integers(0, 4, int(rng.integers(20, 80))))
    slots = [0] if rng.integers(0, 3) else [0, 1]
    n_calls = int(rng.integers(1, 4))
    bursts = {0: [], 1: []}
    for s in (0, 1):
        if s not in slots:
            continue
        for _ in range(n_calls):
            dst, src = int(rng.integers(1, 1 << 24)), int(rng.integers(1, 1 << 24))
            head = synth.dmr_lc(int(rng.choice([0, 3])), 0, 0, dst, src)
            lcs = [head]
            for _ in range(int(rng.integers(0, 3))):
                lcs += _alias_lcs(rng)
            if rng.integers(0, 2):
                lcs.append(bytes([8, 0]) + bytes(rng.integers(0, 256, 7).astype(np.uint8)))
            if rng.integers(0, 2):
                lcs.append(head)
            sync_kind = str(rng.choice(["bs", "ms"]))
            bursts[s].append(synth.dmr_data_burst(s, cc, 1, head + bytes(3), sync_kind + "_data", rng))
            for lc9 in lcs:
                frags = synth.dmr_embedded_lc_fragments(lc9)
                for f in range(6):
                    mid = synth.DMR_SYNC[sync_kind + "_voice"] if f == 0 else synth.dmr_emb_mid(cc, [1, 3, 3, 2][f - 1], frags[f - 1]) if f <= 4 \
                        else synth.dmr_emb_mid(cc, 0, [0] * 16)
                    bursts[s].append(synth.dmr_voice_burst(s, list(rng.integers(0, 4, 108)), mid, rng))
            if rng.integers(0, 4):
                bursts[s].append(synth.dmr_data_burst(s, cc, 2, head + bytes(3), sync_kind + "_data", rng))
            for _ in range(int(rng.integers(0, 4))):
                bursts[s].append(synth.dmr_idle_burst(s, cc, rng))
    n = max(len(bursts[0]), len(bursts[1]))
    for i in range(n):
        for s in (0, 1):
            b = bursts[s][i] if i < len(bursts[s]) else synth.dmr_idle_burst(s, cc, rng)
            out += b
        if rng.integers(0, 40) == 0:
            out += synth.dmr_idle_burst(0, cc, rng)              # one burst too many: the TACT then disagrees with the alternation
    out = np.array(out + list(rng.integers(0, 4, 300)), np.uint8)
    ber = float(rng.choice([0, 0, 0.002, 0.01, 0.04]))
    if ber:
        hit = rng.random(out.size) < ber
        out[hit] ^= rng.integers(1, 4, int(hit.sum())).astype(np.uint8)
    if rng.integers(0, 4) == 0:                                  # a hole: sync is lost and found again
        a = int(rng.integers(0, max(1, out.size - 1500)))
        out[a:a + int(rng.integers(300, 1500))] = rng.integers(0, 4, 1)[0]
    return out


def _ysf_stream(rng):
    def field():
        s = bytes(rng.choice(list(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789-/"), int(rng.integers(0, 11))))
        pad = b" " if rng.integers(0, 4) else b"\n"
        return (s + pad * 10)[:10] if rng.integers(0, 6) else bytes(rng.integers(0x20, 0x100, 10).astype(np.uint8))
    out = list(rng.integers(0, 4, int(rng.integers(20, 90))))
    for _ in range(int(rng.integers(1, 4))):
        dt = int(rng.choice([0, 1, 2, 2, 2, 3]))
        if rng.integers(0, 5):
            csd = (field() + field(), field() + field())
            out += synth.ysf_frame(rng, 0, dt, csd=csd)
        gps = bytearray(20)
        gps[1:4] = bytes([0x22, 0x62, 0x5F]) if rng.integers(0, 4) else bytes(rng.integers(0, 256, 3).astype(np.uint8))
        gps[4] = 0x28
        gps[5:14] = bytes([0x30 + int(rng.integers(0, 10)) for _ in range(3)]) + bytes([int(rng.choice([0x50, 0x30])) + int(rng.integers(0, 10)) for _ in range(3)]) \
            + bytes(rng.integers(0x1c, 0x7f, 3).astype(np.uint8))
        gps[18] = 0x03 if rng.integers(0, 6) else 0x02
        gps[19] = (sum(gps[:19]) + (0 if rng.integers(0, 6) else 1)) & 0xFF
        dch = {0: field(), 1: field(), 2: field(), 3: field(), 6: bytes(gps[:10]), 7: bytes(gps[10:])}
        for i in range(int(rng.integers(4, 20))):
            fn = i & 7
            if rng.integers(0, 25) == 0:
                continue                                         # a lost frame: the data collector sees a sequence error
            out += synth.ysf_frame(rng, 1, dt, fn, dch=dch.get(fn) if dt == 2 else None)
        if rng.integers(0, 4):
            out += synth.ysf_frame(rng, 2, dt)
        out += list(rng.integers(0, 4, int(rng.integers(0, 700))))
    out = np.array(out + [0] * 600, np.uint8)
    ber = float(rng.choice([0, 0, 0.002, 0.01]))
    if ber:
        hit = rng.random(out.size) < ber
        out[hit] ^= rng.integers(1, 4, int(hit.sum())).astype(np.uint8)
    return out


def _dstar_sentence(rng):
    """One slow-data sentence: DPRS / NMEA, valid or broken in one of the ways the parser has a branch for"""
    kind = int(rng.integers(0, 14))
    call = "".join(chr(int(c)) for c in rng.integers(65, 91, 6))
    if kind <= 2:
        s = synth.dstar_dprs_sentence("%s>APDPRS,DSTAR*:!%04d.%02dN/%05d.%02dE>%s" % (call, rng.integers(0, 9000), rng.integers(0, 100), rng.integers(0, 18000),
                                                                                     rng.integers(0, 100), "".join(chr(int(c)) for c in rng.integers(0x20, 0x100, int(rng.integers(0, 12))))))
        if kind == 1:
            s = s[:5] + s[5:9].lower() + s[9:]                       # the checksum's hex digits in lower case
        if kind == 2 and rng.integers(0, 2):
            s = s[:12] + bytes([s[12] ^ 1]) + s[13:]                 # CRC failure
        return s
    if kind <= 6:
        s = synth.dstar_gga_sentence(float(rng.uniform(-89, 89)), float(rng.uniform(-179, 179)))
        if kind == 4:
            s = s.replace(b"\r\n", b"\r")                            # termination may be \r or \r\n
        if kind == 5:
            s = s[:8] + bytes([s[8] ^ 2]) + s[9:]                    # checksum failure
        if kind == 6:
            s = s.replace(b"*", b"*0x"[:int(rng.integers(1, 4))], 1) if rng.integers(0, 2) else s[:-4].lower() + s[-4:]
        return s
    if kind == 7:                                                    # another NMEA sentence: checked, then ignored
        body = "GPRMC,123519,A,4807.038,N,01131.000,E,022.4,084.4,230394,003.1,W"
        cs = 0
        for ch in body.encode():
            cs ^= ch
        return ("$%s*%02X\r\n" % (body, cs)).encode()
    if kind == 8:                                                    # GGA with fields missing or not numeric; the checksum holds
        body = str(rng.choice(["GPGGA,1,,N,,E", "GPGGA,1,x,N,1.0,E", "GPGGA,1,4807.038", "GPGGA,1,4807.038,S,01131.000,W,", "GP", "G", "GPGGA,,,,,,,,,"]))
        cs = 0
        for ch in body.encode():
            cs ^= ch
        return ("$%s*%02X\r" % (body, cs)).encode()
    if kind == 9:
        return b"$$CRC" + bytes(rng.integers(0x20, 0x7F, int(rng.integers(0, 12))).astype(np.uint8)) + b"\r"
    if kind == 10:
        return b"$" + bytes(rng.integers(0x20, 0x7F, int(rng.integers(0, 20))).astype(np.uint8)) + b"\r\n"
    if kind == 11:
        return bytes(rng.integers(0, 256, int(rng.integers(1, 25))).astype(np.uint8)) + b"\r"
    if kind == 12:
        return b"$GPGGA,no star\r\n*\r*1\r"
    return b"\r\n\r"


def _dstar_stream(rng):
    def call(n):
        k = int(rng.integers(0, n + 1))
        if rng.integers(0, 5) == 0:
            return "".join(chr(int(c)) for c in rng.choice([0x41, 0xC4, 0xE9, 0x20, 0x31, 0x00], k))
        return "".join(chr(int(c)) for c in rng.choice(list(b"ABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 /"), k))
    out = [rng.integers(0, 2, int(rng.integers(30, 200))).astype(np.uint8)]
    for _ in range(int(rng.integers(1, 5))):
        simple = b"".join(_dstar_sentence(rng) for _ in range(int(rng.integers(0, 4))))
        msg = "".join(chr(int(c)) for c in rng.integers(0x20, 0x100, int(rng.integers(0, 21))))
        if rng.integers(0, 6) == 0:
            msg = msg[:len(msg) // 2] + "\0" + msg[len(msg) // 2 + 1:]
        nsf = max(int(rng.integers(1, 6)), 2 * (-(-len(simple) // 30)) - 1 + int(rng.integers(0, 2)))
        b, _, _ = synth.dstar_transmission(rng, my=call(8), your=call(8), rpt1=call(8), rpt2=call(8), suffix=call(4), message=msg, n_superframes=nsf,
                                           simple=simple, with_header=bool(rng.integers(0, 5)), inline_header=bool(rng.integers(0, 4)),
                                           data_flag=rng.integers(0, 8) == 0)
        if rng.integers(0, 5) == 0:
            b = b[:len(b) - int(rng.integers(48, 2000))]           # the transmission fades away: no terminator, syncs run out
        out += [b, rng.integers(0, 2, int(rng.integers(30, 3000))).astype(np.uint8)]
    bits = np.concatenate(out)
    ber = float(rng.choice([0, 0, 0, 0.001, 0.004]))
    if ber:
        bits = bits ^ (rng.random(bits.size) < ber).astype(np.uint8)
    return bits


def _product_lines(ctx, exe, proto, syms, chunk):
    eng = api.Engine(1, len(syms), rrc="none", demod="none", proto=proto, ctx=ctx)
    batches = b""
    for lo in range(0, len(syms), chunk):
        part = np.ascontiguousarray(syms[None, lo:lo + chunk])
        eng.push_symbols(part, np.array([part.shape[1]], np.uint32))
        e, ec = eng.events()
        batches += np.uint32(ec[0]).tobytes() + e[0, :ec[0]].tobytes()
    eng.close()
    return subprocess.run([exe, proto], input=batches, capture_output=True, check=True).stdout.split(b"\n")[:-1]


def _oracle_lines(oracle, proto, syms):
    _, ev = oracle.Decoder(proto).process(syms)
    return list(M.lines(proto, ev))


@pytest.mark.parametrize("proto,count", [("dmr", 120), ("ysf", 80), ("nxdn", 30), ("dstar", 100)])
def test_collector_lines_equal_the_oracle_lines(ctx, oracle, meta_exe, proto, count):
    rng = np.random.default_rng({"dmr": 1, "ysf": 2, "nxdn": 3, "dstar": 4}[proto])
    total = 0
    kinds = set()
    for case in range(count):
        if proto == "dmr":
            syms = _dmr_stream(rng)
        elif proto == "ysf":
            syms = _ysf_stream(rng)
        elif proto == "dstar":
            syms = _dstar_stream(rng)
        else:
            syms = synth.nxdn_stream(int(rng.integers(0, 1 << 30)), int(rng.integers(8, 40)), src=int(rng.integers(1, 65535)), dst=int(rng.integers(1, 65535)))
            if rng.integers(0, 3) == 0:
                hit = rng.random(syms.size) < 0.01
                syms = syms.copy(); syms[hit] ^= 1
        ref = _oracle_lines(oracle, proto, syms)
        got = _product_lines(ctx, meta_exe, proto, syms, int(rng.choice([len(syms), 5000, 1777])))
        assert got == ref, "%s case %d:\n%s\n--- oracle ---\n%s" % (proto, case, b"\n".join(got[:40]).decode(errors="replace"), b"\n".join(ref[:40]).decode(errors="replace"))
        total += len(ref)
        for l in ref:
            for key in (b"talkeralias:", b"lat:", b"sync:data", b"type:direct", b"type:group", b"mode:DN", b"mode:V1", b"source:", b"type:conference", b"type:individual", b"dprs:", b"message:", b"ourcall:",
                        b"departure:"):
                if key in l:
                    kinds.add(key)
    assert total > 3 * count                                      # the streams do produce metadata
    if proto == "dmr":
        assert {b"talkeralias:", b"lat:", b"sync:data", b"type:group", b"type:direct"} <= kinds
    if proto == "ysf":
        assert {b"lat:", b"mode:DN", b"source:"} <= kinds
    if proto == "dstar":
        assert {b"lat:", b"dprs:", b"message:", b"ourcall:", b"departure:", b"sync:data"} <= kinds
