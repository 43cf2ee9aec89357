"""Metadata lines: an independent restatement of the reference's MetaCollector logic (TEST INFRASTRUCTURE ONLY).

Turns a decoder event stream (one record per call the reference's phases make into their MetaCollector, as produced by
oracle/decoders.c -- or by the product, which must emit the same records) into the `k:v;k:v` lines the reference writes:

  base      src/lib/meta.cpp:8-17 (keys in std::map order, ';' separated), :71-99 (hold / release / dirty)
  DMR       src/dmr_decoder/dmr_meta.cpp:11-178 (Slot, withSlot, reset), dmr_phase.cpp:80,104-115,178-202,229-233,277-296,304-339
            (call sites), talkeralias.cpp:23-143, gps.cpp:7-16, lc.cpp:26-43
  YSF       src/ysf_decoder/ysf_meta.cpp:13-105, ysf_phase.cpp:50,73,87,112,133,139-164,258-305,351-361, data.cpp:24-88, gps.cpp:5-82
  NXDN      src/nxdn_decoder/nxdn_meta.cpp:6-76, nxdn_phase.cpp:50,115,138,153, sacch.cpp:141-155
  D-Star    src/dstar_decoder/dstar_meta.cpp:9-135, dstar_phase.cpp:50,97,105,110,205-290, header.cpp:150-181, crc.cpp:6-23

PARITY UNPINNED: those translation units include <csdr/module.hpp> or ICU and cannot be built in this image; this file
follows them line by line and is what the product's collectors (include/digiham/*_meta.hpp) are compared with.
"""
import numpy as np

F32 = np.float32


def _b(v):
    return v if isinstance(v, bytes) else v.encode("utf-8")


def serialize(d):
    """StringSerializer::serializeMetaData (meta.cpp:8-17): std::map iterates its keys in byte order.  Lines are BYTES: the
    reference cuts talker aliases at a byte count, possibly inside a UTF-8 character."""
    return b";".join(_b(k) + b":" + _b(d[k]) for k in sorted(d))


def f2s(x):
    """std::to_string(float): %f"""
    return "%f" % float(x)


def latin1(raw):
    """Converter::convertToUtf8 (charset.cpp:10-28): ICU from ISO-8859-1, the result is read as a C string (cut at the first NUL)"""
    raw = bytes(raw)
    if not raw:
        return ""
    out = raw.decode("latin-1")
    cut = out.find("\0")
    return out if cut < 0 else out[:cut]


# ------------------------------------------------------------------------------------------------- DMR
class _TalkerAlias:
    """talkeralias.cpp:23-143"""

    def __init__(self):
        self.data = bytearray(28)
        self.blocks = 0

    def reset(self):
        self.blocks = 0

    def set_block(self, block, data7):
        self.data[block * 7:block * 7 + 7] = bytes(data7[:7])
        self.blocks |= 1 << block

    def has_header(self):
        return bool(self.blocks & 1)

    def fmt(self):
        return self.data[0] >> 6

    def length(self):
        return (self.data[0] & 0b00111110) >> 1

    def collected_bytes(self):
        i = 0
        while i < 4:
            mask = (1 << (i + 1)) - 1
            if (self.blocks & mask) != mask:
                break
            i += 1
        return i * 7

    @staticmethod
    def _seven(s):
        r = [(s[0] & 0xFE) >> 1, (s[0] & 1) << 6 | (s[1] & 0xFC) >> 2, (s[1] & 3) << 5 | (s[2] & 0xF8) >> 3,
             (s[2] & 7) << 4 | (s[3] & 0xF0) >> 4, (s[3] & 0x0F) << 3 | (s[4] & 0xE0) >> 5, (s[4] & 0x1F) << 2 | (s[5] & 0xC0) >> 6,
             (s[5] & 0x3F) << 1 | (s[6] & 0x80) >> 7, s[6] & 0x7F]
        return bytes(r)

    def contents(self):
        """bytes (the reference's std::string)"""
        if not self.has_header():
            return b""
        n = self.collected_bytes()
        f = self.fmt()
        if f == 0:                                   # 7 bit: first character is trash (built from the header bits)
            res = b"".join(self._seven(self.data[i:i + 7]) for i in range(0, n, 7))[1:]
        elif f == 1:                                 # 8 bit: ISO-8859-1 -> UTF-8, read back as a C string
            res = latin1(self.data[1:n]).encode("utf-8")
        elif f == 2:                                 # UTF-8 as it is
            res = bytes(self.data[1:n])
        else:                                        # UTF-16 big endian -> UTF-8
            chars = (n - 1) // 2
            res = bytes(self.data[1:1 + 2 * chars]).decode("utf-16-be", errors="surrogatepass").encode("utf-8", errors="surrogatepass")
        return res[:self.length()] if len(res) > self.length() else res

    def complete(self):
        if not self.has_header():
            return False
        n = self.collected_bytes()
        f = self.fmt()
        if f == 0:
            return (n * 7) // 8 - 1 >= self.length()
        if f == 1:
            return n - 1 >= self.length()
        if f == 2:
            return len(self.contents()) >= self.length()
        return (n - 1) // 2 >= self.length()


def dmr_gps(d):
    """gps.cpp:7-16 (float arithmetic)"""
    lat = ((d[4] & 0x7F) << 16) | (d[5] << 8) | d[6]
    if d[4] & 0x80:
        lat = -lat
    lon = (d[1] << 16) | (d[2] << 8) | d[3]
    if d[0] & 1:
        lon = -lon
    return F32(F32(180.0) / F32(1 << 24)) * F32(lat), F32(F32(360.0) / F32(1 << 25)) * F32(lon)


class _Slot:
    def __init__(self):
        self.sync, self.type, self.source, self.target, self.alias, self.coord, self.dirty = -1, -1, 0, 0, b"", None, False

    def _set(self, name, v):
        if getattr(self, name) == v:
            return
        setattr(self, name, v)
        self.dirty = True

    def soft_reset(self):
        self._set("type", -1); self._set("source", 0); self._set("target", 0); self._set("alias", b""); self._set("coord", None)

    def reset(self):
        self.soft_reset(); self._set("sync", -1)

    def collect(self):
        r = {}
        if self.sync > 0:
            r["sync"] = {1: "data", 2: "voice"}.get(self.sync, "unknown")
        if self.type > 0:
            r["type"] = {1: "direct", 2: "group"}.get(self.type, "unknown")
        if self.source > 0:
            r["source"] = str(self.source)
        if self.target > 0:
            r["target"] = str(self.target)
        if self.alias:
            r["talkeralias"] = self.alias
        if self.coord is not None:
            r["lat"], r["lon"] = f2s(self.coord[0]), f2s(self.coord[1])
        return r


class DmrLines:
    """SYNCTYPE_DATA = 1, SYNCTYPE_VOICE = 2 (dmr_phase.hpp); META_TYPE_DIRECT = 1, META_TYPE_GROUP = 2 (dmr_meta.hpp)"""

    def __init__(self):
        self.slots = [_Slot(), _Slot()]
        self.alias = [_TalkerAlias(), _TalkerAlias()]
        self.sync_types = [-1, -1]
        self.pending_alias_reset = [None, None]      # a slot reset by the TACT switch (:80): its collector is cleared at its next burst (144 symbols on) unless that burst is in voice mode
        self.lines = []

    def _send(self, i):                              # sendMetaDataForSlot (dmr_meta.cpp:157-170)
        s = self.slots[i]
        if not s.dirty:
            return
        d = {"protocol": "DMR", "slot": str(i)}
        d.update(s.collect())
        self.lines.append(serialize(d))
        s.dirty = False

    def _frame_of_slot(self, slot, sym_index, voice_sync_now):
        """bookkeeping of talkerAliasCollector[slot]->reset() (dmr_phase.cpp:233: every burst of a slot that is not in voice mode)"""
        p = self.pending_alias_reset[slot]
        if p is not None:
            if not (voice_sync_now and sym_index == p):
                self.alias[slot].reset()
            self.pending_alias_reset[slot] = None

    def consume(self, ev):
        t, a, b = int(ev["type"]), int(ev["a"]) & 1, int(ev["b"])
        idx = int(ev["sym_index"])
        pay = bytes(bytearray(ev["payload"][:int(ev["len"])]))
        if t == 1:                                   # SYNC (:104-115)
            self._frame_of_slot(a, idx, b == 2)
            self.sync_types[a] = b
            s = self.slots[a]
            s._set("sync", b)
            if pay and pay[0]:
                s.soft_reset()
            self._send(a)
            if b != 2:
                self.alias[a].reset()                # :233 in the same burst
        elif t == 2:                                 # SLOT_RESET
            if b == 1:                               # :80: the other slot; its burst comes next
                self.sync_types[a] = -1
                self.slots[a].reset(); self._send(a)
                self.pending_alias_reset[a] = (idx + 144) & 0xFFFFFFFF
            else:                                    # :178 / :196 (sync lost) or :295 (a burst outside voice and data mode)
                self._frame_of_slot(a, idx, False)
                self.sync_types[a] = -1
                self.slots[a].reset(); self._send(a)
                self.alias[a].reset()                # the same burst then runs :233
        elif t == 3:                                 # META_RESET (:184, :202): MetaCollector::reset, then a new FramePhase later
            for i in range(2):
                self.slots[i].reset()
            for i in range(2):
                self._send(i)
            self.alias = [_TalkerAlias(), _TalkerAlias()]
            self.sync_types = [-1, -1]
            self.pending_alias_reset = [None, None]
        elif t == 7:                                 # SLOTTYPE: a data burst (:233 has run)
            self._frame_of_slot(a, idx, False)
            self.alias[a].reset()
        elif t == 5:                                 # SOFT_RESET (:279-282)
            self.slots[a].soft_reset(); self._send(a)
        elif t == 4 and len(pay) >= 9:               # LC -> handleLc (:304-339)
            self._frame_of_slot(a, idx, False)
            op = pay[0] & 0x3F
            s = self.slots[a]
            if op in (0, 3):                         # LC_OPCODE_GROUP, LC_OPCODE_UNIT_TO_UNIT (lc.hpp)
                s._set("type", 2 if op == 0 else 1)
                s._set("target", pay[3] << 16 | pay[4] << 8 | pay[5])
                s._set("source", pay[6] << 16 | pay[7] << 8 | pay[8])
                self._send(a)
            elif 4 <= op <= 7:                       # talker alias header / blocks 1-3
                al = self.alias[a]
                al.set_block(op - 4, pay[2:9])
                if al.complete():
                    text = al.contents().rstrip(b"\0")
                    s._set("alias", text)
                self._send(a)
            elif op == 8:                            # LC_GPS_INFO
                s._set("coord", tuple(dmr_gps(pay[2:9])))
                self._send(a)


# ------------------------------------------------------------------------------------------------- YSF
def ysf_string(raw10):
    """treatYsfString (ysf_phase.cpp:351-361): cut at the first newline, then at the first space before it"""
    raw = bytes(raw10[:10])
    n = 10
    for c in (b"\n", b" "):
        k = raw[:n].find(c)
        if k >= 0:
            n = k
    return latin1(raw[:n])


def ysf_gps(d):
    """gps.cpp:5-82 (float arithmetic; `lon` starts from 0 where the reference leaves it uninitialised)"""
    for i in range(6):
        if (d[i] & 0x0F) > 9:
            return None
    lat = F32((d[0] & 0x0F) * 10 + (d[1] & 0x0F))
    for k, div in ((2, 6), (3, 60), (4, 600), (5, 6000)):
        lat = F32(lat + F32(F32(d[k] & 0x0F) / F32(div)))
    direction = d[3] & 0xF0
    if direction == 0x30:
        lat = -lat
    elif direction != 0x50:
        return None
    lon = F32(0)
    b, c = d[4] & 0xF0, d[6]
    if b == 0x50:
        if 0x76 <= c < 0x7F:
            lon = F32(c - 0x76)
        elif 0x6C <= c < 0x75:
            lon = F32(100 + (c - 0x6C))
        elif 0x26 <= c < 0x6B:
            lon = F32(110 + (c - 0x26))
        else:
            return None
    elif b == 0x30:
        if 0x26 <= c < 0x7F:
            lon = F32(10 + (c - 0x26))
        else:
            return None
    b = d[7]
    if 0x58 < b <= 0x61:
        lon = F32(lon + F32(F32(b - 0x58) / F32(60)))
    elif 0x26 <= b <= 0x57:
        lon = F32(lon + F32(F32(10 + (b - 0x26)) / F32(60)))
    else:
        return None
    b = d[8]
    if 0x1C <= b < 0x7F:
        lon = F32(lon + F32(F32(b - 0x1C) / F32(6000)))
    else:
        return None
    direction = d[5] & 0xF0
    if direction == 0x50:
        lon = -lon
    elif direction != 0x30:
        return None
    if lat > 90 or lat < -90 or lon > 180 or lon < -180:
        return None
    return (lat, lon)


class YsfLines:
    KEYS = {"mode": "mode", "destination": "target", "source": "source", "up": "up", "down": "down", "radio": "radio"}

    def __init__(self):
        self.f = {k: "" for k in self.KEYS}
        self.coord = None
        self.held = 0
        self.dirty = False
        self.lines = []
        self.dc_next = 0                              # DataCollector (data.cpp:44-88)
        self.dc = bytearray(20)
        self.in_header = False

    def _send(self):                                  # MetaCollector::sendMetaData() (meta.cpp:93-99)
        if self.held:
            self.dirty = True
            return
        d = {"protocol": "YSF"}
        for k, key in self.KEYS.items():
            if self.f[k]:
                d[key] = self.f[k]
        if self.coord is not None:
            d["lat"], d["lon"] = f2s(self.coord[0]), f2s(self.coord[1])
        self.lines.append(serialize(d))

    def _hold(self):
        self.held += 1

    def _release(self):
        self.held -= 1
        if self.held == 0:
            if self.dirty:
                self._send()
            self.dirty = False

    def _set(self, k, v):
        if self.f[k] == v:
            return
        self.f[k] = v
        self._send()

    def _set_gps(self, c):
        if self.coord == c:
            return
        self.coord = c
        self._send()

    def _reset(self):
        self._hold()
        for k in ("mode", "destination", "source", "up", "down", "radio"):
            self._set(k, "")
        self._set_gps(None)
        self._release()

    def _end_header(self):
        if self.in_header:
            self._release()
            self.in_header = False

    def consume(self, ev):
        t, a, b = int(ev["type"]), int(ev["a"]), int(ev["b"])
        pay = bytes(bytearray(ev["payload"][:int(ev["len"])]))
        if t != 19:
            self._end_header()
        if t == 20:                                   # META_RESET: b = 0 sync lost (:50), 1 header channel (:141-142), 2 terminator (:163)
            self._reset()
            if b == 1:
                self._hold()
                self.in_header = True
        elif t == 17:                                 # setMode (:73, :87, :112, :133) by FICH data type
            self._set("mode", {0: "V1", 1: "FR data", 2: "DN", 3: "VW"}[b & 3])
        elif t == 19:                                 # header DCH (:143-155): CSD1 = destination + source, CSD2 = down + up
            if a == 0:
                self._set("destination", ysf_string(pay[0:10])); self._set("source", ysf_string(pay[10:20]))
            else:
                self._set("down", ysf_string(pay[0:10])); self._set("up", ysf_string(pay[10:20]))
        elif t == 18:                                 # V/D2 DCH of frame number a (:270-305)
            fn = a
            if fn < 6:
                if fn == 0:
                    self._set("destination", ysf_string(pay))
                elif fn == 1:
                    self._set("source", ysf_string(pay))
                elif fn == 2:
                    self._set("down", ysf_string(pay))
                elif fn == 3:
                    self._set("up", ysf_string(pay))
                self.dc_next = 0
            if 6 <= fn < 8:
                off = fn - 6
                if off != self.dc_next:
                    self.dc_next = 0
                else:
                    self.dc_next = off + 1
                    self.dc[off * 10:off * 10 + 10] = pay[:10]      # (the reference copies only when in sequence)
            if self.dc_next >= 2:
                d = self.dc
                if d[18] == 0x03 and (sum(d[:19]) & 0xFF) == d[19]:
                    cmd = d[1] << 16 | d[2] << 8 | d[3]
                    self._set_gps(ysf_gps(d[5:14]) if cmd == 0x22625F else None)

    def finish(self):
        self._end_header()


# ------------------------------------------------------------------------------------------------- NXDN
class NxdnLines:
    def __init__(self):
        self.sync, self.type, self.source, self.destination = "", "", 0, 0
        self.held, self.dirty, self.lines = 0, False, []

    def _send(self):
        if self.held:
            self.dirty = True
            return
        d = {"protocol": "NXDN"}
        if self.sync:
            d["sync"] = self.sync
        if self.type:
            d["type"] = self.type
        if self.source:
            d["source"] = str(self.source)
        if self.destination:
            d["destination"] = str(self.destination)
        self.lines.append(serialize(d))

    def _set(self, k, v):
        if getattr(self, k) == v:
            return
        setattr(self, k, v)
        self._send()

    def consume(self, ev):
        t = int(ev["type"])
        pay = bytes(bytearray(ev["payload"][:int(ev["len"])]))
        if t == 37:                                   # reset (nxdn_meta.cpp:69-76)
            self.held += 1
            self._set("sync", ""); self._set("type", ""); self._set("source", 0); self._set("destination", 0)
            self.held -= 1
            if self.held == 0:
                if self.dirty:
                    self._send()
                self.dirty = False
        elif t == 35:
            self._set("sync", "voice")
        elif t == 34 and len(pay) >= 7:               # setFromSacch (:54-67)
            if pay[0] & 0x3F == 0x01:
                ct = pay[2] >> 5
                self._set("type", "conference" if ct == 1 else "individual" if ct == 4 else "")
                self._set("source", pay[3] << 8 | pay[4])
                self._set("destination", pay[5] << 8 | pay[6])


# ------------------------------------------------------------------------------------------------- D-Star
def dstar_crc_ok(data, to_check):
    """Crc::isCrcValid (src/dstar_decoder/crc.cpp:6-23)"""
    c = 0xFFFF
    for byte in data:
        for i in range(8):
            c ^= (byte >> i) & 1
            c = (c >> 1) ^ 0x8408 if c & 1 else c >> 1
    return (c ^ 0xFFFF) == to_check


def hex_extract_u16(text):
    """`std::stringstream ss; ss << std::hex << text; ss >> (uint16_t) v` (dstar_phase.cpp:222-225, :262-265) as libstdc++'s
    num_get does it: white space skipped, an optional sign, an optional 0x, hex digits; no digit -> 0."""
    i, n = 0, len(text)
    while i < n and text[i] in b" \t\n\v\f\r":
        i += 1
    neg = False
    if i < n and text[i] in b"+-":
        neg = text[i] == 0x2D
        i += 1
    found_zero, sep_pos, had_x = False, 0, False       # num_get::_M_extract_int, base 16 (checked against libstdc++ 11)
    while i < n:
        if text[i] == 0x30 and not found_zero:
            found_zero, sep_pos = True, sep_pos + 1
        elif found_zero and text[i] in b"xX" and not had_x:
            found_zero, sep_pos, had_x = False, 0, True
        else:
            break
        i += 1
    v = 0
    while i < n and text[i] in b"0123456789abcdefABCDEF":
        v = v * 16 + int(chr(text[i]), 16)
        sep_pos += 1
        i += 1
    if not sep_pos and not found_zero:
        return 0                                       # failbit, value 0 (C++11)
    if v > 0xFFFF:
        return 0xFFFF                                  # overflow: failbit, the largest value
    return (-v) & 0xFFFF if neg else v


def stof(text):
    """std::stof: strtof's longest valid prefix after white space; None where it throws"""
    import re
    m = re.match(rb"[ \t\n\v\f\r]*[+-]?(?:0[xX](?:[0-9a-fA-F]+\.?[0-9a-fA-F]*|\.[0-9a-fA-F]+)(?:[pP][+-]?[0-9]+)?|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?|[iI][nN][fF](?:[iI][nN][iI][tT][yY])?|[nN][aA][nN])", text)
    if not m:
        return None
    t = m.group(0).strip().decode()
    try:
        v = float.fromhex(t) if "x" in t.lower() else float(t)
    except (ValueError, OverflowError):
        return None
    with np.errstate(over="ignore"):
        r = F32(v)
    if (np.isinf(r) and not np.isinf(v)) or (v != 0 and abs(float(r)) < 1.1754943508222875e-38):
        return None                                    # out_of_range (glibc's strtof reports overflow and underflow)
    return r


class DstarLines:
    """dstar_meta.cpp:9-135 on the calls of dstar_phase.cpp:50,97,105,110,205-290 (header.cpp:150-181 for the fields)"""
    KEYS = (("sync", "sync"), ("departure", "departure"), ("destination", "destination"), ("ourcall", "ourcall"),
            ("yourcall", "yourcall"), ("message", "message"), ("dprs", "dprs"))

    def __init__(self):
        self.v = {k: b"" for k, _ in self.KEYS}
        self.coord = None
        self.header = bytearray(41)
        self.simple = b""
        self.held, self.dirty, self.lines = 0, False, []

    def _send(self):                                   # meta.cpp:93-99
        if self.held:
            self.dirty = True
            return
        d = {"protocol": "DSTAR"}                      # dstar_meta.cpp:100-135
        for k, name in self.KEYS:
            if self.v[k]:
                d[name] = self.v[k]
        if self.coord is not None:
            d["lat"], d["lon"] = f2s(self.coord[0]), f2s(self.coord[1])
        self.lines.append(serialize(d))

    def _hold(self):
        self.held += 1

    def _release(self):                                # meta.cpp:79-91
        self.held -= 1
        if self.held == 0:
            if self.dirty:
                self._send()
            self.dirty = False

    def _set(self, k, v):
        v = _b(v)
        if self.v[k] == v:
            return
        self.v[k] = v
        self._send()

    def _set_gps(self, c):                             # dstar_meta.cpp:71-81
        if self.coord is None and c is None:
            return
        if self.coord is not None and c is not None and self.coord[0] == c[0] and self.coord[1] == c[1]:
            return
        self.coord = c
        self._send()

    def _field(self, at, n):
        return latin1(self.header[at:at + n]).rstrip(" ")

    def _from_header(self):                            # dstar_meta.cpp:15-27
        self._hold()
        self._set("sync", "data" if (self.header[0] >> 7) & 1 else "voice")
        self._set("departure", self._field(11, 8))
        self._set("destination", self._field(3, 8))
        own, suffix = self._field(27, 8), self._field(35, 4)
        self._set("ourcall", own + "/" + suffix if suffix != "" else own)
        self._set("yourcall", self._field(19, 8))
        self._release()

    def _nmea(self, s):                                # dstar_phase.cpp:248-290
        star = s.rfind(b"*")
        if star < 0 or star + 2 > len(s):
            return
        if star < 1:
            return                                     # (the reference's substr(1, npos-ish) case; see the product's note)
        body = s[1:star]
        if len(body) < 2:
            return                                     # where body.substr(2, 3) would throw
        sentence = body[2:5]
        x = 0
        for ch in body:
            x ^= ch
        if x != hex_extract_u16(s[star + 1:star + 3]):
            return
        fields = body.split(b",")
        if body.endswith(b","):
            fields = fields[:-1]                       # getline() yields no empty item behind a trailing separator
        if sentence == b"GGA":
            if len(fields) < 6:
                return                                 # (undefined behaviour in the reference)
            la, lo = stof(fields[2]), stof(fields[4])
            if la is None or lo is None:
                return                                 # (an exception ends the reference's process)
            with np.errstate(all="ignore"):
                if not np.isfinite(la) or not np.isfinite(lo) or abs(float(la)) >= 2 ** 31 or abs(float(lo)) >= 2 ** 31:
                    return                             # (int) of such a float is undefined; the streams do not carry one
                lat = F32(int(la) // 100 if la >= 0 else -((-int(la)) // 100))
                lat = F32(lat + F32(F32(la - F32(lat * F32(100))) / F32(60)))
                if fields[3] == b"S":
                    lat = F32(-lat)
                lon = F32(int(lo) // 100 if lo >= 0 else -((-int(lo)) // 100))
                lon = F32(lon + F32(F32(lo - F32(lon * F32(100))) / F32(60)))
                if fields[5] == b"W":
                    lon = F32(-lon)
            self._set_gps((lat, lon))

    def _parse_simple(self):                           # dstar_phase.cpp:218-245
        while True:
            pos = self.simple.find(b"\r")
            if pos < 0:
                break
            s = self.simple[:pos + 1]
            if len(s) >= 10 and s[:5] == b"$$CRC" and s[9:10] == b",":
                if dstar_crc_ok(s[10:], hex_extract_u16(s[5:9])):
                    self._set("dprs", s[10:len(s) - 1])
            elif len(s) > 5 and s[:1] == b"$":
                self._nmea(s)
            self.simple = self.simple[pos + 1 + (1 if self.simple[pos + 1:pos + 2] == b"\n" else 0):]

    def consume(self, ev):
        t = int(ev["type"])
        pay = bytes(bytearray(ev["payload"][:int(ev["len"])]))
        if t == 64:                                    # the 41 header bytes come as two records
            if int(ev["a"]) == 0:
                self.header[:24] = pay[:24]
            else:
                self.header[24:41] = pay[:17]
                self._from_header()
        elif t == 65:                                  # a new VoicePhase (dstar_phase.hpp:77)
            self.simple = b""
        elif t == 66:
            self._set("sync", "voice")
        elif t == 67:                                  # :207
            self._set("message", latin1(pay[:20]))
        elif t == 68:                                  # :178
            self.simple += pay
        elif t == 69:
            self._parse_simple()
        elif t == 70:                                  # dstar_meta.cpp:83-94
            self._hold()
            for k, _ in self.KEYS:
                self._set(k, "")
            self._set_gps(None)
            self._release()


def lines(proto, events):
    m = {"dmr": DmrLines, "ysf": YsfLines, "nxdn": NxdnLines, "dstar": DstarLines}[proto]()
    for ev in events:
        m.consume(ev)
    if hasattr(m, "finish"):
        m.finish()
    return m.lines
