"""Python front-end over the C ABI (include/digiham_amd.h).

Device memory and streams come from PyTorch-ROCm (plumbing only): inputs are
``torch`` CUDA tensors, an engine enqueues its HIP kernels on the stream that was
torch's current stream WHEN THE ENGINE WAS CREATED (inputs produced on another stream
must be ordered against it by the caller), outputs are fetched into numpy arrays on request.  All compute happens
in ``libdigiham_amd.so``; nothing here computes or falls back.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import DhError

EVENT_DTYPE = np.dtype([("sym_index", "<u4"), ("type", "u1"), ("a", "u1"), ("b", "u1"), ("len", "u1"),
                        ("payload", "u1", (24,))])

_CODES = {"hamming_7_4": np.uint8, "hamming_13_9": np.uint16, "hamming_15_11": np.uint16,
          "hamming_16_11": np.uint16, "quadratic_residue": np.uint16, "golay_20_8": np.uint32, "golay_24_12": np.uint32, "bch_31_21": np.uint32}


class TorchCudaMemory:
    """Device arrays as torch CUDA tensors."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("digiham_amd needs an MI355X (gfx950) visible to PyTorch-ROCm; there is no CPU path")
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.index = device

    _NP2T = {"uint8": "uint8", "int16": "int16", "uint16": "int16", "uint32": "int32", "int32": "int32",
             "float32": "float32"}

    def from_numpy(self, a):
        a = np.ascontiguousarray(a)
        t = self.torch.from_numpy(a.view(getattr(np, self._NP2T[a.dtype.name])) if a.dtype.name in ("uint16", "uint32") else a)
        return t.to(self.device)

    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=getattr(self.torch, self._NP2T[np.dtype(dtype).name]), device=self.device)

    def to_numpy(self, t, dtype=None):
        a = t.detach().cpu().numpy()
        return a.view(dtype) if dtype is not None else a

    def ptr(self, t):
        return C.c_void_p(t.data_ptr())

    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def is_device_array(self, x):
        return self.torch.is_tensor(x) and x.is_cuda


def parse_lc(payload):
    """Fields of a 9-byte DMR link control word (DH_EV_DMR_LC payload): what Digiham::Dmr::Lc's getters return
    (src/dmr_decoder/lc.cpp:26-43)."""
    d = bytes(bytearray(payload[:9]))
    return {"opcode": d[0] & 0x3F, "feature_set_id": d[1], "target": d[3] << 16 | d[4] << 8 | d[5],
            "source": d[6] << 16 | d[7] << 8 | d[8], "data": d[2:9]}


def _check(rc, what, lib):
    if rc != 0:
        raise DhError(rc, what, lib.dh_last_error().decode(errors="replace"))


class Context:
    """A loaded library + a device-memory provider."""

    def __init__(self, lib=None, mem=None, device=0):
        self.lib = lib if lib is not None else _capi.load()
        self.mem = mem if mem is not None else TorchCudaMemory(device)

    # ------------------------------------------------------------------ batch FEC
    def block_decode(self, code, words):
        """words: numpy array -> (corrected words, ok flags) as numpy."""
        dt = _CODES[code]
        w = self.mem.from_numpy(np.ascontiguousarray(words, dt).ravel())
        ok = self.mem.zeros((w.shape[0],), np.uint8)
        _check(getattr(self.lib, "dh_" + code)(self.mem.ptr(w), self.mem.ptr(ok), w.shape[0], self.mem.stream()), "dh_" + code, self.lib)
        return self.mem.to_numpy(w, dt), self.mem.to_numpy(ok)

    def bptc_196_96(self, payloads):
        p = self.mem.from_numpy(np.ascontiguousarray(payloads, np.uint8).reshape(-1, 25))
        n = p.shape[0]
        out = self.mem.zeros((n, 12), np.uint8)
        ok = self.mem.zeros((n,), np.uint8)
        _check(self.lib.dh_bptc_196_96(self.mem.ptr(p), self.mem.ptr(out), self.mem.ptr(ok), n, self.mem.stream()), "dh_bptc_196_96", self.lib)
        return self.mem.to_numpy(out), self.mem.to_numpy(ok)

    def trellis(self, packed, n_dibits):
        p = np.ascontiguousarray(packed, np.uint8)
        n, stride = p.shape
        ob = (n_dibits + 7) // 8
        d = self.mem.from_numpy(p)
        out = self.mem.zeros((n, ob), np.uint8)
        metric = self.mem.zeros((n,), np.uint8)
        _check(self.lib.dh_trellis(self.mem.ptr(d), stride, n_dibits, self.mem.ptr(out), ob, self.mem.ptr(metric), n, self.mem.stream()),
               "dh_trellis", self.lib)
        return self.mem.to_numpy(out), self.mem.to_numpy(metric)

    def crc16(self, data, count):
        a = np.ascontiguousarray(data, np.uint8)
        n, stride = a.shape
        d = self.mem.from_numpy(a)
        out = self.mem.zeros((n,), np.uint16)
        _check(self.lib.dh_crc16(self.mem.ptr(d), stride, count, self.mem.ptr(out), n, self.mem.stream()), "dh_crc16", self.lib)
        return self.mem.to_numpy(out, np.uint16)

    def whitening(self, data, n_bits):
        a = np.ascontiguousarray(data, np.uint8)
        n, stride = a.shape
        d = self.mem.from_numpy(a)
        out = self.mem.zeros((n, stride), np.uint8)
        _check(self.lib.dh_whitening(self.mem.ptr(d), self.mem.ptr(out), stride, n_bits, n, self.mem.stream()), "dh_whitening", self.lib)
        return self.mem.to_numpy(out)

    def debug_div_gain(self, x, narrow=False):
        d = self.mem.from_numpy(np.ascontiguousarray(x, np.float32).ravel())
        out = self.mem.zeros((d.shape[0],), np.float32)
        _check(self.lib.dh_debug_div_gain(self.mem.ptr(d), self.mem.ptr(out), d.shape[0], int(narrow), self.mem.stream()),
               "dh_debug_div_gain", self.lib)
        return self.mem.to_numpy(out)

    def frontend(self, x, mode, dcblock=True, state=None):
        """The receiver front-end (dh_frontend_s16): x int16 [B][n] audio (mode "audio") or [B][2 n] interleaved I / Q ("iq").
        Returns (float32 device array [B][n] for Engine.push, state) -- pass the state back in to continue the streams."""
        a = x if self.mem.is_device_array(x) else self.mem.from_numpy(np.ascontiguousarray(x, np.int16))
        B, w = a.shape
        n = w if mode == "audio" else w // 2
        out = self.mem.zeros((B, n), np.float32)
        if state is None:
            state = self.mem.zeros((B, 4), np.float32)
        _check(self.lib.dh_frontend_s16(self.mem.ptr(a), w, self.mem.ptr(out), n, self.mem.ptr(state), B, n,
                                        1 if mode == "audio" else 2, int(bool(dcblock)), self.mem.stream()), "dh_frontend_s16", self.lib)
        return out, state

    def debug_div_const(self, x, divisor):
        """x (numpy float32, or a device array) / float(divisor) as the slicer kernels compute it; returns what it was given."""
        dev = self.mem.is_device_array(x)
        d = x if dev else self.mem.from_numpy(np.ascontiguousarray(x, np.float32).ravel())
        out = self.mem.zeros((d.shape[0],), np.float32)
        _check(self.lib.dh_debug_div_const(self.mem.ptr(d), self.mem.ptr(out), d.shape[0], int(divisor), self.mem.stream()),
               "dh_debug_div_const", self.lib)
        return out if dev else self.mem.to_numpy(out)

    def debug_mfma_f16(self, a, b, c):
        """d[t] = c[t] + a[t] @ b[t] on the matrix cores: a [T][16][32], b [T][32][16] as float16 (or uint16 bit patterns), c [T][16][16] float32."""
        a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
        a = a.view(np.uint16) if a.dtype == np.float16 else a.astype(np.uint16)
        b = b.view(np.uint16) if b.dtype == np.float16 else b.astype(np.uint16)
        T = a.shape[0]
        da, db = self.mem.from_numpy(a.reshape(T, 512)), self.mem.from_numpy(b.reshape(T, 512))
        dc = self.mem.from_numpy(np.ascontiguousarray(c, np.float32).reshape(T, 256))
        out = self.mem.zeros((T, 256), np.float32)
        _check(self.lib.dh_debug_mfma_f16(self.mem.ptr(da), self.mem.ptr(db), self.mem.ptr(dc), self.mem.ptr(out), T, self.mem.stream()),
               "dh_debug_mfma_f16", self.lib)
        return self.mem.to_numpy(out).reshape(T, 16, 16)

    def debug_f16_split(self, x, scale=1.0):
        """(h1, h2) as float16 arrays: the two halves the error-bounded FIR makes of x * scale."""
        d = self.mem.from_numpy(np.ascontiguousarray(x, np.float32).ravel())
        n = d.shape[0]
        h1, h2 = self.mem.zeros((n,), np.uint16), self.mem.zeros((n,), np.uint16)
        _check(self.lib.dh_debug_f16_split(self.mem.ptr(d), self.mem.ptr(h1), self.mem.ptr(h2), n, float(scale), self.mem.stream()),
               "dh_debug_f16_split", self.lib)
        return self.mem.to_numpy(h1).view(np.float16), self.mem.to_numpy(h2).view(np.float16)

    def dvfilter(self, x, state=None):
        """x: int16 [B][n] numpy; returns (y, state) with state a device array [B][22] to carry on."""
        a = np.ascontiguousarray(x, np.int16)
        if a.ndim == 1:
            a = a[None, :]
        B, n = a.shape
        d = self.mem.from_numpy(a)
        out = self.mem.zeros((B, n), np.int16)
        if state is None:
            state = self.mem.zeros((B, 22), np.float32)
        _check(self.lib.dh_dvfilter_s16(self.mem.ptr(d), self.mem.ptr(out), self.mem.ptr(state), B, n, n, self.mem.stream()),
               "dh_dvfilter_s16", self.lib)
        return self.mem.to_numpy(out), state


class Engine:
    """B independent `rrc_filter | gfsk_demodulator | dmr_decoder` pipes with state resident in HBM."""

    def __init__(self, n_channels, max_samples, rrc="wide", demod="gfsk", sps=10, proto="dmr", fast_fir=False,
                 keep_filtered=False, invert=False, events=True, slot_filter=3, ctx=None, device=0, ordered_timing=False, split_stages=False,
                 taps=None, gain=None, exact_symbols=False, exact_fir=False, overlap_pushes=False, one_launch=False):
        """rrc = "custom" takes the caller's coefficient table: `taps` (nZeros + 1 floats, any shape) and `gain`, as
        Digiham::RrcFilter::RrcFilter(nZeros, gain, coeffs[]) does (include/rrc_filter.hpp:12)."""
        self.ctx = ctx if ctx is not None else Context(device=device)
        lib, mem = self.ctx.lib, self.ctx.mem
        flags = (_capi.FLAG_FAST_FIR if fast_fir else 0) | (_capi.FLAG_KEEP_FILTERED if keep_filtered else 0) | \
                (_capi.FLAG_FSK_INVERT if invert else 0) | (0 if events else _capi.FLAG_NO_EVENTS) | \
                (_capi.FLAG_ORDERED_TIMING if ordered_timing else 0) | (_capi.FLAG_SPLIT_STAGES if split_stages else 0) | \
                (_capi.FLAG_EXACT_SYMBOLS if exact_symbols else 0) | (_capi.FLAG_EXACT_FIR if exact_fir else 0) | \
                (_capi.FLAG_OVERLAP_PUSHES if overlap_pushes else 0) | (_capi.FLAG_ONE_LAUNCH if one_launch else 0)
        cfg = _capi.EngineConfig(C.sizeof(_capi.EngineConfig), getattr(mem, "index", 0), n_channels, max_samples,
                                 _capi.RRC[rrc], _capi.DEMOD[demod], sps, _capi.PROTO[proto], flags, slot_filter,
                                 mem.stream())
        if rrc != "custom":
            cfg.struct_size = _capi.EngineConfig.rrc_taps.offset       # the layout before the custom-filter fields: every library version takes it
        if rrc == "custom":
            t = np.ascontiguousarray(taps, np.float32).ravel()
            cfg.rrc_taps = t.ctypes.data_as(C.POINTER(C.c_float))         # copied by dh_engine_create
            cfg.rrc_nzeros, cfg.rrc_gain = len(t) - 1, float(gain)
        h = C.c_void_p()
        _check(lib.dh_engine_create(C.byref(cfg), C.byref(h)), "dh_engine_create", lib)
        self._h = h
        self.B, self.max_samples = n_channels, max_samples
        self.has_demod, self.has_proto = _capi.DEMOD[demod] != 0, _capi.PROTO[proto] != 0
        self.keep_filtered = (keep_filtered or rrc == "custom") and _capi.RRC[rrc] != 0
        self._keep = None
        # DH_FLAG_OVERLAP_PUSHES: the kernels of a push run on the engine's own streams, which torch's caching allocator knows
        # nothing about -- EVERY pushed buffer has to stay alive (and untouched) until the streams are joined again
        # (sync, reset or any read), not just the latest one
        self._overlap = bool(overlap_pushes)
        self._inflight = []

    def close(self):
        if getattr(self, "_h", None):
            self.ctx.lib.dh_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        _check(self.ctx.lib.dh_engine_reset(self._h), "dh_engine_reset", self.ctx.lib)
        if self._inflight:
            self.sync()                          # the reset joined the streams; inputs are released once it has run

    def set_slot_filter(self, f):
        _check(self.ctx.lib.dh_engine_set_slot_filter(self._h, f), "dh_engine_set_slot_filter", self.ctx.lib)

    def push(self, x, n=None, counts=None):
        """x: device array float32 [B][stride] (torch CUDA tensor); processes the first n samples of every row -- or, with
        `counts` ([B] uint32), the first counts[b] <= n samples of row b (dh_engine_push_ragged)."""
        mem = self.ctx.mem
        if not mem.is_device_array(x):
            x = mem.from_numpy(np.ascontiguousarray(x, np.float32).reshape(self.B, -1))
        elif getattr(mem, "torch", None) is not None and mem.torch.is_tensor(x):
            # the library reads raw float32 rows: anything else would be silent garbage or an out-of-bounds read
            if x.dtype != mem.torch.float32 or x.dim() != 2 or x.shape[0] != self.B or (x.shape[1] > 1 and x.stride(1) != 1):
                raise ValueError("Engine.push: need a float32 [%d][n] device array with contiguous rows, got %s %s strides %s"
                                 % (self.B, x.dtype, tuple(x.shape), tuple(x.stride())))
            if x.device.index != mem.index:
                raise ValueError("Engine.push: the array lives on cuda:%s, the engine on cuda:%s" % (x.device.index, mem.index))
        stride = x.stride(0) if callable(getattr(x, "stride", None)) else x.strides[0] // x.itemsize
        if x.shape[0] == 1:
            stride = x.shape[1]                 # a single row: its "row stride" is arbitrary (numpy reports 0 for a new axis)
        n = x.shape[1] if n is None else n
        self._keep = x          # the launch is asynchronous: keep the input alive
        if self._overlap:
            self._inflight.append(x)
        if counts is not None:  # ragged push: channel b brings counts[b] <= n samples
            c = counts if mem.is_device_array(counts) else mem.from_numpy(np.ascontiguousarray(counts, np.uint32))
            self._keep = (x, c)
            if self._overlap:
                self._inflight.append(c)
            _check(self.ctx.lib.dh_engine_push_ragged(self._h, mem.ptr(x), stride, mem.ptr(c), n), "dh_engine_push_ragged", self.ctx.lib)
            return
        _check(self.ctx.lib.dh_engine_push(self._h, mem.ptr(x), stride, n), "dh_engine_push", self.ctx.lib)

    def push_host(self, x):
        a = np.ascontiguousarray(x, np.float32).reshape(self.B, -1)
        _check(self.ctx.lib.dh_engine_push_host(self._h, a.ctypes.data_as(C.c_void_p), a.shape[1], a.shape[1]),
               "dh_engine_push_host", self.ctx.lib)

    def push_symbols(self, syms, counts):
        mem = self.ctx.mem
        s = syms if mem.is_device_array(syms) else mem.from_numpy(np.ascontiguousarray(syms, np.uint8).reshape(self.B, -1))
        c = counts if mem.is_device_array(counts) else mem.from_numpy(np.ascontiguousarray(counts, np.uint32))
        self._keep = (s, c)
        _check(self.ctx.lib.dh_engine_push_symbols(self._h, mem.ptr(s), s.shape[1], mem.ptr(c)), "dh_engine_push_symbols", self.ctx.lib)

    def timing_enable(self, max_pushes):
        _check(self.ctx.lib.dh_engine_timing_enable(self._h, max_pushes), "dh_engine_timing_enable", self.ctx.lib)
        self._timing_cap = max_pushes

    def timing_read(self):
        """(rrc_ms, slicer_ms, decoder_ms) arrays, one entry per push since the last read (HIP events)."""
        cap = getattr(self, "_timing_cap", 0)
        a, b, c = (np.zeros(max(cap, 1), np.float32) for _ in range(3))
        n = C.c_uint32(cap)
        _check(self.ctx.lib.dh_engine_timing_read(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                  c.ctypes.data_as(C.c_void_p), C.byref(n)), "dh_engine_timing_read", self.ctx.lib)
        return a[:n.value], b[:n.value], c[:n.value]

    def timing_read_split(self):
        """(first_ms, first_channels) per push since the last timing_read: duration and channel count of the first of the
        two launches of a push of a large engine with overlap_pushes (zeros otherwise).  Call before timing_read()."""
        cap = getattr(self, "_timing_cap", 0)
        a, c = np.zeros(max(cap, 1), np.float32), np.zeros(max(cap, 1), np.uint32)
        n = C.c_uint32(cap)
        _check(self.ctx.lib.dh_engine_timing_read_split(self._h, a.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), C.byref(n)),
               "dh_engine_timing_read_split", self.ctx.lib)
        return a[:n.value], c[:n.value]

    def timing_stats(self):
        """(blocks, ordered) per channel: 100-symbol timing blocks evaluated, and those decided by the in-order chain."""
        blocks, ordered = np.zeros(self.B, np.uint32), np.zeros(self.B, np.uint32)
        _check(self.ctx.lib.dh_engine_timing_stats(self._h, blocks.ctypes.data_as(C.c_void_p), ordered.ctypes.data_as(C.c_void_p)),
               "dh_engine_timing_stats", self.ctx.lib)
        return blocks, ordered

    def debug_header(self, word):
        out = np.zeros(self.B, np.uint32)
        _check(self.ctx.lib.dh_engine_debug_header(self._h, word, out.ctypes.data_as(C.c_void_p)), "dh_engine_debug_header", self.ctx.lib)
        return out

    def sync(self):
        try:
            _check(self.ctx.lib.dh_engine_sync(self._h), "dh_engine_sync", self.ctx.lib)
        finally:
            del self._inflight[:-1]              # everything queued has run (the latest stays in _keep as before)

    def _fetch(self, getter, elem_dtype, with_counts=True):
        lib = self.ctx.lib
        p, stride, cnt = C.c_void_p(), C.c_size_t(), C.c_void_p()
        if with_counts:
            _check(getter(self._h, C.byref(p), C.byref(stride), C.byref(cnt)), getter.__name__, lib)
        else:
            _check(getter(self._h, C.byref(p), C.byref(stride)), getter.__name__, lib)
        self.sync()
        rows = np.empty((self.B, stride.value), elem_dtype)
        _check(lib.dh_copy_to_host(rows.ctypes.data_as(C.c_void_p), p, rows.nbytes), "dh_copy_to_host", lib)
        counts = None
        if with_counts:
            counts = np.empty(self.B, np.uint32)
            _check(lib.dh_copy_to_host(counts.ctypes.data_as(C.c_void_p), cnt, counts.nbytes), "dh_copy_to_host", lib)
        return rows, counts

    def symbols(self):
        """(dibits [B][stride] uint8, counts [B]) of the last push."""
        return self._fetch(self.ctx.lib.dh_engine_symbols, np.uint8)

    def frames(self):
        return self._fetch(self.ctx.lib.dh_engine_frames, np.uint8)

    def events(self):
        return self._fetch(self.ctx.lib.dh_engine_events, EVENT_DTYPE)

    def filtered(self):
        rows, _ = self._fetch(self.ctx.lib.dh_engine_filtered, np.float32, with_counts=False)
        return rows

    def read_rows(self, what, channels):
        """The current push's rows of the given channels only: what = "symbols" | "frames" | "events" -> (rows [len(channels)][stride],
        counts [len(channels)]); "filtered" -> (rows, None).  For looking at a few channels of a large engine (bench.py checks the
        engine it has just timed) without copying every row to the host."""
        lib = self.ctx.lib
        getter, dt = {"symbols": (lib.dh_engine_symbols, np.dtype(np.uint8)), "frames": (lib.dh_engine_frames, np.dtype(np.uint8)),
                      "events": (lib.dh_engine_events, EVENT_DTYPE), "filtered": (lib.dh_engine_filtered, np.dtype(np.float32))}[what]
        p, stride, cnt = C.c_void_p(), C.c_size_t(), C.c_void_p()
        if what == "filtered":
            _check(getter(self._h, C.byref(p), C.byref(stride)), getter.__name__, lib)
        else:
            _check(getter(self._h, C.byref(p), C.byref(stride), C.byref(cnt)), getter.__name__, lib)
        self.sync()
        channels = [int(c) for c in channels]
        rows = np.empty((len(channels), stride.value), dt)
        row_bytes = stride.value * dt.itemsize
        for j, b in enumerate(channels):
            if not 0 <= b < self.B:
                raise ValueError("Engine.read_rows: channel %d of %d" % (b, self.B))
            _check(lib.dh_copy_to_host(rows[j].ctypes.data_as(C.c_void_p), C.c_void_p(p.value + b * row_bytes), row_bytes), "dh_copy_to_host", lib)
        if what == "filtered":
            return rows, None
        counts = np.empty(self.B, np.uint32)
        _check(lib.dh_copy_to_host(counts.ctypes.data_as(C.c_void_p), cnt, counts.nbytes), "dh_copy_to_host", lib)
        return rows, counts[channels]

    def device_views(self):
        """Raw device pointers of the output buffers (for zero-copy consumers)."""
        lib = self.ctx.lib
        out = {}
        for name, getter in (("symbols", lib.dh_engine_symbols), ("frames", lib.dh_engine_frames), ("events", lib.dh_engine_events)):
            p, stride, cnt = C.c_void_p(), C.c_size_t(), C.c_void_p()
            if getter(self._h, C.byref(p), C.byref(stride), C.byref(cnt)) == 0:
                out[name] = (p.value, stride.value, cnt.value)
        return out
