"""ctypes/numpy front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  It loads

* ``oracle/liboracle.so``  -- the C restatement of the reference algorithms, and
* ``oracle/_ref/libdigiham_ref_fec.so`` (optional) -- the reference's own
  pure-C FEC sources compiled unmodified (see oracle/Makefile).

Nothing under ``digiham_amd/`` imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

EVENT_DTYPE = np.dtype([("sym_index", "<u4"), ("type", "u1"), ("a", "u1"), ("b", "u1"),
                        ("len", "u1"), ("payload", "u1", (24,))])
assert EVENT_DTYPE.itemsize == 32


def build(force=False):
    """(Re)build liboracle.so and, when /root/reference is present, _ref/."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.exists("/root/reference/src"):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _declare(_LIB)
    return _LIB


def ref():
    """The compiled reference FEC, or None when it has not been built."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libdigiham_ref_fec.so")
        if not os.path.exists(so):
            return None
        _REF = C.CDLL(so)
    return _REF


def _declare(L):
    vp = C.c_void_p
    L.orc_rrc_new.restype = vp; L.orc_rrc_new.argtypes = [C.c_int]
    L.orc_rrc_new_custom.restype = vp; L.orc_rrc_new_custom.argtypes = [C.c_uint, C.c_double, vp]
    L.orc_rrc_free.argtypes = [vp]
    L.orc_rrc_process.argtypes = [vp, vp, vp, C.c_size_t]
    L.orc_rrc_taps.restype = C.POINTER(C.c_float)
    L.orc_rrc_taps.argtypes = [C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_double)]
    L.orc_demod_new.restype = vp; L.orc_demod_new.argtypes = [C.c_uint, C.c_int, C.c_int]
    L.orc_demod_free.argtypes = [vp]
    L.orc_demod_process.restype = C.c_size_t
    L.orc_demod_process.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orc_dvfilter_new.restype = vp
    L.orc_dvfilter_free.argtypes = [vp]
    L.orc_dvfilter_process.argtypes = [vp, vp, vp, C.c_size_t]
    L.orc_dmr_new.restype = vp
    L.orc_ysf_new.restype = vp
    L.orc_nxdn_new.restype = vp
    L.orc_pocsag_new.restype = vp
    L.orc_dstar_new.restype = vp
    L.orc_dstar_scramble.argtypes = [vp, vp, vp, C.c_size_t]
    L.orc_dstar_crc.restype = C.c_uint16
    L.orc_dstar_crc.argtypes = [vp, C.c_size_t]
    L.orc_dstar_header_crc_ok.argtypes = [vp]
    L.orc_dstar_header_parse.argtypes = [vp, vp]
    L.orc_bch_31_21_encode.restype = C.c_uint32
    L.orc_bch_31_21_encode.argtypes = [C.c_uint32]
    L.orc_nxdn_trellis_decode.restype = C.c_uint
    L.orc_nxdn_trellis_decode.argtypes = [vp, vp, C.c_size_t]
    L.orc_nxdn_scramble.argtypes = [vp, vp, vp, C.c_size_t]
    L.orc_nxdn_lich_parse.argtypes = [vp]
    L.orc_nxdn_sacch_parse.argtypes = [vp, vp]
    L.orc_nxdn_facch1_parse.argtypes = [vp, vp]
    L.orc_decoder_free.argtypes = [vp]
    L.orc_dmr_set_slot_filter.argtypes = [vp, C.c_uint8]
    L.orc_decoder_process.restype = C.c_size_t
    L.orc_decoder_process.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t),
                                      vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.orc_chain_run.restype = C.c_int
    L.orc_bptc_196_96_encode.argtypes = [vp, vp]
    L.orc_trellis_encode.argtypes = [vp, C.c_int, vp]
    for name, rt in [("orc_hamming_7_4_encode", C.c_uint8), ("orc_hamming_13_9_encode", C.c_uint16),
                     ("orc_hamming_15_11_encode", C.c_uint16), ("orc_hamming_16_11_encode", C.c_uint16),
                     ("orc_golay_20_8_encode", C.c_uint32), ("orc_golay_24_12_encode", C.c_uint32),
                     ("orc_quadratic_residue_encode", C.c_uint16)]:
        getattr(L, name).restype = rt
        getattr(L, name).argtypes = [C.c_uint32]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# ------------------------------------------------------------------ FEC (batch)
_WORD = {"hamming_7_4": np.uint8, "hamming_13_9": np.uint16, "hamming_15_11": np.uint16,
         "hamming_16_11": np.uint16, "quadratic_residue": np.uint16,
         "golay_20_8": np.uint32, "golay_24_12": np.uint32, "bch_31_21": np.uint32}


def _impl(which):
    if which == "oracle":
        return lib(), "orc_batch_"
    r = ref()
    if r is None:
        raise RuntimeError("oracle/_ref/libdigiham_ref_fec.so not built")
    return r, "ref_batch_"


def block_decode(code, words, which="oracle"):
    """Decode an array of codewords; returns (corrected_words, ok_flags)."""
    L, pre = _impl(which)
    w = np.ascontiguousarray(words, dtype=_WORD[code]).copy()
    ok = np.zeros(w.shape, np.uint8)
    getattr(L, pre + code)(_p(w), _p(ok), C.c_size_t(w.size))
    return w, ok


def bptc_196_96(payloads, which="oracle"):
    L, pre = _impl(which)
    p = np.ascontiguousarray(payloads, np.uint8).reshape(-1, 25)
    out = np.zeros((p.shape[0], 12), np.uint8)
    ok = np.zeros(p.shape[0], np.uint8)
    getattr(L, pre + "bptc_196_96")(_p(p), _p(out), _p(ok), C.c_size_t(p.shape[0]))
    return out, ok


def trellis(packed, ndibits, which="oracle"):
    L, pre = _impl(which)
    p = np.ascontiguousarray(packed, np.uint8)
    n, stride = p.shape
    ob = (ndibits + 7) // 8
    out = np.zeros((n, ob), np.uint8)
    metric = np.zeros(n, np.uint8)
    getattr(L, pre + "trellis")(_p(p), C.c_size_t(stride), C.c_uint8(ndibits), _p(out), C.c_size_t(ob),
                                _p(metric), C.c_size_t(n))
    return out, metric


def crc16(data, count, which="oracle"):
    L, pre = _impl(which)
    d = np.ascontiguousarray(data, np.uint8)
    n, stride = d.shape
    out = np.zeros(n, np.uint16)
    getattr(L, pre + "crc16")(_p(d), C.c_size_t(stride), C.c_int(count), _p(out), C.c_size_t(n))
    return out


def whitening(data, nbits, which="oracle"):
    L, pre = _impl(which)
    d = np.ascontiguousarray(data, np.uint8)
    n, stride = d.shape
    out = np.zeros_like(d)
    getattr(L, pre + "whitening")(_p(d), _p(out), C.c_size_t(stride), C.c_uint8(nbits), C.c_size_t(n))
    return out


def hamming_distance(a, b, which="oracle"):
    L, pre = _impl(which)
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    n, size = a.shape
    out = np.zeros(n, np.uint32)
    getattr(L, pre + "hamming_distance")(_p(a), _p(b), C.c_size_t(size), _p(out), C.c_size_t(n))
    return out


def encode(code, value):
    return int(getattr(lib(), "orc_%s_encode" % code)(C.c_uint32(int(value))))


def bptc_encode(info12):
    info = np.ascontiguousarray(info12, np.uint8)
    out = np.zeros(25, np.uint8)
    lib().orc_bptc_196_96_encode(_p(info), _p(out))
    return out


def trellis_encode(bits_packed, nbits):
    b = np.ascontiguousarray(bits_packed, np.uint8)
    out = np.zeros((nbits + 3) // 4, np.uint8)
    lib().orc_trellis_encode(_p(b), C.c_int(nbits), _p(out))
    return out


# ------------------------------------------------------------------ DSP
def rrc_taps(narrow=False):
    nz = C.c_uint(); g = C.c_double()
    p = lib().orc_rrc_taps(int(narrow), C.byref(nz), C.byref(g))
    return np.ctypeslib.as_array(p, (nz.value + 1,)).copy(), g.value


class Rrc:
    """RrcFilter: the wide / narrow design, or any table (taps = nZeros + 1 coefficients, gain)."""

    def __init__(self, narrow=False, taps=None, gain=None):
        if taps is None:
            self._h = lib().orc_rrc_new(int(narrow))
        else:
            t = np.ascontiguousarray(taps, np.float32)
            self._h = lib().orc_rrc_new_custom(len(t) - 1, float(gain), _p(t))

    def process(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        lib().orc_rrc_process(self._h, _p(x), _p(y), C.c_size_t(x.size))
        return y

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_rrc_free(self._h); self._h = None


def frontend(x, mode, dcblock=True, state=None):
    """Front-end oracle over one channel: x = int16 audio (mode "audio") or interleaved int16 I / Q ("iq").
    Returns (float32 samples, state) -- pass the state back in to continue the stream."""
    a = np.ascontiguousarray(x, np.int16).ravel()
    n = a.size if mode == "audio" else a.size // 2
    st = np.zeros(4, np.float32) if state is None else np.ascontiguousarray(state, np.float32).copy()
    out = np.zeros(n, np.float32)
    lib().orc_frontend_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]
    lib().orc_frontend_process(_p(st), _p(a), C.c_size_t(n), _p(out), 1 if mode == "audio" else 2, int(bool(dcblock)))
    return out, st


class Demod:
    """Streaming GfskDemodulator (levels=4) / FskDemodulator (levels=2): keeps the unread tail like a csdr reader."""

    def __init__(self, sps=10, levels=4, invert=False):
        self._h = lib().orc_demod_new(sps, levels, int(invert))
        self._tail = np.zeros(0, np.float32)

    def process(self, x):
        x = np.concatenate([self._tail, np.asarray(x, np.float32)])
        out = np.zeros(x.size // 1 + 16, np.uint8) if x.size < 64 else np.zeros(x.size // 4 + 16, np.uint8)
        n = C.c_size_t()
        used = lib().orc_demod_process(self._h, _p(x), C.c_size_t(x.size), _p(out), C.c_size_t(out.size), C.byref(n))
        self._tail = x[used:].copy()
        return out[:n.value].copy()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_demod_free(self._h); self._h = None


class DvFilter:
    def __init__(self):
        self._h = lib().orc_dvfilter_new()

    def process(self, x):
        x = np.ascontiguousarray(x, np.int16)
        y = np.empty_like(x)
        lib().orc_dvfilter_process(self._h, _p(x), _p(y), C.c_size_t(x.size))
        return y

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_dvfilter_free(self._h); self._h = None


class Decoder:
    """Streaming Dmr::Decoder / Ysf::Decoder; returns (output bytes, events) per call."""

    def __init__(self, proto):
        self._h = {"dmr": lib().orc_dmr_new, "ysf": lib().orc_ysf_new, "nxdn": lib().orc_nxdn_new, "pocsag": lib().orc_pocsag_new,
                   "dstar": lib().orc_dstar_new}[proto]()
        self._tail = np.zeros(0, np.uint8)

    def set_slot_filter(self, f):
        lib().orc_dmr_set_slot_filter(self._h, f)

    def process(self, syms):
        s = np.concatenate([self._tail, np.asarray(syms, np.uint8)])
        out = np.zeros(s.size + 256, np.uint8)
        ev = np.zeros(s.size // 20 + 64, EVENT_DTYPE)
        no = C.c_size_t(); ne = C.c_size_t()
        used = lib().orc_decoder_process(self._h, _p(s), C.c_size_t(s.size), _p(out), C.c_size_t(out.size),
                                         C.byref(no), _p(ev), C.c_size_t(ev.size), C.byref(ne))
        self._tail = s[used:].copy()
        return out[:no.value].copy(), ev[:ne.value].copy()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_decoder_free(self._h); self._h = None


class ChainCfg(C.Structure):
    _fields_ = [("rrc", C.c_int), ("levels", C.c_int), ("invert", C.c_int), ("sps", C.c_uint),
                ("proto", C.c_int), ("slot_filter", C.c_int)]


def chain(x, rrc=1, levels=4, invert=False, sps=10, proto=1, slot_filter=3, threads=1, keep_filtered=False):
    """Run the whole reference pipe over x[channels][n]; returns a dict of per-channel results."""
    x = np.ascontiguousarray(x, np.float32)
    if x.ndim == 1:
        x = x[None, :]
    B, n = x.shape
    cfg = ChainCfg(rrc, levels, int(invert), sps, proto, slot_filter)
    filt = np.zeros_like(x) if (keep_filtered and rrc) else None
    sym_stride = n // max(sps - 1, 1) + 16
    syms = np.zeros((B, sym_stride), np.uint8)
    sym_count = np.zeros(B, np.uint32)
    out_stride = sym_stride // 4 + 128
    out = np.zeros((B, out_stride), np.uint8)
    out_count = np.zeros(B, np.uint32)
    ev_stride = sym_stride // 40 + 64
    ev = np.zeros((B, ev_stride), EVENT_DTYPE)
    ev_count = np.zeros(B, np.uint32)
    rc = lib().orc_chain_run(C.byref(cfg), _p(x), C.c_size_t(B), C.c_size_t(n), C.c_size_t(n),
                             _p(filt) if filt is not None else None,
                             _p(syms), C.c_size_t(sym_stride), _p(sym_count),
                             _p(out), C.c_size_t(out_stride), _p(out_count),
                             _p(ev), C.c_size_t(ev_stride), _p(ev_count), C.c_int(threads))
    if rc != 0:
        raise RuntimeError("orc_chain_run failed: %d" % rc)
    return {"filtered": filt, "syms": syms, "sym_count": sym_count, "out": out, "out_count": out_count,
            "events": ev, "event_count": ev_count}


# ------------------------------------------------------------------ NXDN frame elements
_REF_NXDN = None


def ref_nxdn():
    """oracle/_ref/libdigiham_ref_nxdn.so: the reference's own scrambler / LICH / SACCH / FACCH1 / trellis classes."""
    global _REF_NXDN
    if _REF_NXDN is None:
        so = os.path.join(_HERE, "_ref", "libdigiham_ref_nxdn.so")
        if not os.path.exists(so):
            return None
        L = C.CDLL(so)
        vp = C.c_void_p
        L.ref_nxdn_scramble.argtypes = [vp, vp, C.c_size_t]
        L.ref_nxdn_lich_parse.argtypes = [vp]
        L.ref_nxdn_trellis_decode.restype = C.c_uint
        L.ref_nxdn_trellis_decode.argtypes = [vp, vp, C.c_size_t]
        L.ref_nxdn_sacch_parse.argtypes = [vp, vp]
        L.ref_nxdn_facch1_parse.argtypes = [vp, vp]
        _REF_NXDN = L
    return _REF_NXDN


def nxdn_scramble(dibits, which="oracle"):
    """Scrambler from its reset state over a run of dibits (scrambler.cpp:9-25)."""
    d = np.ascontiguousarray(dibits, np.uint8)
    out = np.zeros_like(d)
    if which == "ref":
        ref_nxdn().ref_nxdn_scramble(_p(d), _p(out), d.size)
    else:
        sr = C.c_uint16(0x0E4)
        lib().orc_nxdn_scramble(C.byref(sr), _p(d), _p(out), d.size)
    return out


def nxdn_lich(dibits8, which="oracle"):
    d = np.ascontiguousarray(dibits8, np.uint8)
    return (ref_nxdn().ref_nxdn_lich_parse if which == "ref" else lib().orc_nxdn_lich_parse)(_p(d))


def nxdn_trellis(packed, len_bits, which="oracle"):
    d = np.ascontiguousarray(packed, np.uint8)
    out = np.zeros((len_bits + 15) // 16, np.uint8)
    fn = ref_nxdn().ref_nxdn_trellis_decode if which == "ref" else lib().orc_nxdn_trellis_decode
    metric = fn(_p(d), _p(out), len_bits)
    return out, int(metric)


def nxdn_sacch(dibits30, which="oracle"):
    d = np.ascontiguousarray(dibits30, np.uint8)
    out = np.zeros(5, np.uint8)
    ok = (ref_nxdn().ref_nxdn_sacch_parse if which == "ref" else lib().orc_nxdn_sacch_parse)(_p(d), _p(out))
    return bool(ok), out


def nxdn_facch1(dibits72, which="oracle"):
    d = np.ascontiguousarray(dibits72, np.uint8)
    out = np.zeros(12, np.uint8)
    ok = (ref_nxdn().ref_nxdn_facch1_parse if which == "ref" else lib().orc_nxdn_facch1_parse)(_p(d), _p(out))
    return bool(ok), out


# ------------------------------------------------------------------ D-Star elements
_REF_DSTAR = None


def ref_dstar():
    """oracle/_ref/libdigiham_ref_dstar.so: the reference's own D-Star Scrambler and Crc classes."""
    global _REF_DSTAR
    if _REF_DSTAR is None:
        so = os.path.join(_HERE, "_ref", "libdigiham_ref_dstar.so")
        if not os.path.exists(so):
            return None
        L = C.CDLL(so)
        vp = C.c_void_p
        L.ref_dstar_scramble.argtypes = [vp, vp, C.c_size_t]
        L.ref_dstar_crc_valid.argtypes = [vp, C.c_size_t, C.c_ushort]
        _REF_DSTAR = L
    return _REF_DSTAR


def dstar_scramble(bits, which="oracle"):
    """Scrambler from its reset state over a run of bits (dstar_decoder/scrambler.cpp:7-21)."""
    d = np.ascontiguousarray(bits, np.uint8)
    out = np.zeros_like(d)
    if which == "ref":
        ref_dstar().ref_dstar_scramble(_p(d), _p(out), d.size)
    else:
        sr = C.c_uint8(0x7F)
        lib().orc_dstar_scramble(C.byref(sr), _p(d), _p(out), d.size)
    return out


def dstar_crc(data):
    d = np.ascontiguousarray(data, np.uint8)
    return int(lib().orc_dstar_crc(_p(d), d.size))


def dstar_crc_valid(data, checksum, which="oracle"):
    d = np.ascontiguousarray(data, np.uint8)
    if which == "ref":
        return bool(ref_dstar().ref_dstar_crc_valid(_p(d), d.size, checksum))
    return dstar_crc(d) == checksum


def dstar_header_parse(raw660):
    d = np.ascontiguousarray(raw660, np.uint8)
    out = np.zeros(41, np.uint8)
    ok = lib().orc_dstar_header_parse(_p(d), _p(out))
    return bool(ok), out


# ------------------------------------------------------------------ burst / frame element parsers (batch)
_REF_LIBS = {}


def ref_lib(name):
    """oracle/_ref/libdigiham_ref_<name>.so (the reference's own classes compiled in place), or None when not built."""
    if name not in _REF_LIBS:
        so = os.path.join(_HERE, "_ref", "libdigiham_ref_%s.so" % name)
        _REF_LIBS[name] = C.CDLL(so) if os.path.exists(so) else None
    return _REF_LIBS[name]


class Elements:
    """The element parsers as batch calls.  which = "oracle": oracle/elements.c (orc_el_*); which = "ref": the
    reference's own Cach / Emb / SlotType / EmbeddedCollector / Lc / Gps / TalkerAliasCollector / Fich / DataCollector /
    Codeword / Header classes through oracle/ref_{dmr,ysf,pocsag,dstar}.cpp (ref_el_*).  Same signatures on both sides;
    the host-side-only elements (GPS, talker alias, YSF data frames) exist on the "ref" side only."""
    _HOME = {"dmr_cach": "dmr", "dmr_emb": "dmr", "dmr_slottype": "dmr", "dmr_embedded_lc": "dmr", "dmr_lc": "dmr",
             "dmr_gps": "dmr", "dmr_talkeralias": "dmr", "ysf_fich": "ysf", "ysf_gps": "ysf", "ysf_data": "ysf",
             "pocsag_codeword": "pocsag", "dstar_header": "dstar"}

    def __init__(self, which="oracle"):
        self.which = which

    def _fn(self, name):
        if self.which == "oracle":
            return getattr(lib(), "orc_el_" + name)
        L = ref_lib(self._HOME[name])
        if L is None:
            raise RuntimeError("oracle/_ref/libdigiham_ref_%s.so not built" % self._HOME[name])
        return getattr(L, "ref_el_" + name)

    def dmr_cach(self, raw):
        """raw [n][12] dibits -> [n][8] = has_tact, tact, busy, slot, lcss, payload[3]"""
        raw = np.ascontiguousarray(raw, np.uint8).reshape(-1, 12)
        out = np.zeros((len(raw), 8), np.uint8)
        self._fn("dmr_cach")(_p(raw), C.c_size_t(len(raw)), _p(out))
        return out

    def dmr_emb(self, words):
        w = np.ascontiguousarray(words, np.uint16).ravel()
        out, cor = np.zeros((len(w), 4), np.uint8), np.zeros(len(w), np.uint16)
        self._fn("dmr_emb")(_p(w), C.c_size_t(len(w)), _p(out), _p(cor))
        return out, cor

    def dmr_slottype(self, words):
        w = np.ascontiguousarray(words, np.uint32).ravel()
        out, cor = np.zeros((len(w), 4), np.uint8), np.zeros(len(w), np.uint32)
        self._fn("dmr_slottype")(_p(w), C.c_size_t(len(w)), _p(out), _p(cor))
        return out, cor

    def dmr_embedded_lc(self, prev, frags, nfrags):
        prev = np.ascontiguousarray(prev, np.uint8).reshape(-1, 16)
        frags = np.ascontiguousarray(frags, np.uint8).reshape(-1, 20)
        nfrags = np.ascontiguousarray(nfrags, np.uint8).ravel()
        out = np.zeros((len(prev), 10), np.uint8)
        self._fn("dmr_embedded_lc")(_p(prev), _p(frags), _p(nfrags), C.c_size_t(len(prev)), _p(out))
        return out

    def dmr_lc(self, lc):
        lc = np.ascontiguousarray(lc, np.uint8).reshape(-1, 9)
        fields, data7 = np.zeros((len(lc), 4), np.uint32), np.zeros((len(lc), 7), np.uint8)
        self._fn("dmr_lc")(_p(lc), C.c_size_t(len(lc)), _p(fields), _p(data7))
        return fields, data7

    def dmr_gps(self, d):
        d = np.ascontiguousarray(d, np.uint8).reshape(-1, 7)
        out = np.zeros((len(d), 2), np.float32)
        self._fn("dmr_gps")(_p(d), C.c_size_t(len(d)), _p(out))
        return out

    def dmr_talkeralias(self, blocks, order):
        blocks = np.ascontiguousarray(blocks, np.uint8).reshape(-1, 28)
        order = np.ascontiguousarray(order, np.uint8).reshape(-1, 4)
        n = len(blocks)
        complete, text, ln = np.zeros(n, np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.uint8)
        self._fn("dmr_talkeralias")(_p(blocks), _p(order), C.c_size_t(n), _p(complete), _p(text), _p(ln))
        return complete, text, ln

    def ysf_fich(self, dibits):
        d = np.ascontiguousarray(dibits, np.uint8).reshape(-1, 100)
        out, data = np.zeros((len(d), 4), np.uint8), np.zeros(len(d), np.uint32)
        self._fn("ysf_fich")(_p(d), C.c_size_t(len(d)), _p(out), _p(data))
        return out, data

    def ysf_gps(self, d):
        d = np.ascontiguousarray(d, np.uint8).reshape(-1, 9)
        ok, out = np.zeros(len(d), np.uint8), np.zeros((len(d), 2), np.float32)
        self._fn("ysf_gps")(_p(d), C.c_size_t(len(d)), _p(ok), _p(out))
        return ok, out

    def ysf_data(self, chunks, offsets):
        chunks = np.ascontiguousarray(chunks, np.uint8).reshape(-1, 80)
        offsets = np.ascontiguousarray(offsets, np.uint8).reshape(-1, 8)
        n = len(chunks)
        has2, frame = np.zeros(n, np.uint8), np.zeros((n, 4), np.uint32)
        radio, latlon = np.zeros((n, 32), np.uint8), np.zeros((n, 2), np.float32)
        self._fn("ysf_data")(_p(chunks), _p(offsets), C.c_size_t(n), _p(has2), _p(frame), _p(radio), _p(latlon))
        return has2, frame, radio, latlon

    def pocsag_codeword(self, bits):
        b = np.ascontiguousarray(bits, np.uint8).reshape(-1, 32)
        out, words = np.zeros((len(b), 4), np.uint8), np.zeros((len(b), 3), np.uint32)
        self._fn("pocsag_codeword")(_p(b), C.c_size_t(len(b)), _p(out), _p(words))
        return out, words

    def dstar_header(self, raw):
        r = np.ascontiguousarray(raw, np.uint8).reshape(-1, 660)
        n = len(r)
        ok, data, text = np.zeros(n, np.uint8), np.zeros((n, 41), np.uint8), np.zeros((n, 160), np.uint8)
        self._fn("dstar_header")(_p(r), C.c_size_t(n), _p(ok), _p(data), _p(text))
        return ok, data, text


def all_cach_dibits(start, count):
    """CACH number i (24 bits, dibit 0 = the two most significant bits) for i in [start, start + count) -> [count][12]"""
    i = np.arange(start, start + count, dtype=np.uint32)
    return np.stack([((i >> (22 - 2 * k)) & 3).astype(np.uint8) for k in range(12)], axis=1)
