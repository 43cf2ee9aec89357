"""ctypes declarations for include/digiham_amd.h and the loader of the gfx950 library.

The product library is ``digiham_amd/libdigiham_amd.so`` (built in-tree by
``__graft_entry__.build()``).  There is no CPU implementation behind this
module: if the library is missing, or no MI355X is visible, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdigiham_amd.so")

DH_OK, DH_EINVAL, DH_ENOMEM, DH_EDEVICE, DH_ENODEV, DH_ECAPACITY = 0, -1, -2, -3, -4, -5
RRC = {"none": 0, None: 0, "wide": 1, "narrow": 2, "custom": 3}
DEMOD = {"none": 0, None: 0, "fsk": 2, "fsk2": 2, "gfsk": 4, "gfsk4": 4}
PROTO = {"none": 0, None: 0, "dmr": 1, "ysf": 2, "nxdn": 3, "pocsag": 4, "dstar": 5}
FLAG_FAST_FIR, FLAG_KEEP_FILTERED, FLAG_FSK_INVERT, FLAG_NO_EVENTS, FLAG_ORDERED_TIMING, FLAG_SPLIT_STAGES = 1, 2, 4, 8, 16, 32
FLAG_EXACT_SYMBOLS, FLAG_EXACT_FIR, FLAG_OVERLAP_PUSHES, FLAG_ONE_LAUNCH = 64, 128, 256, 512


class EngineConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("n_channels", C.c_uint32),
                ("max_samples", C.c_uint32), ("rrc", C.c_int32), ("demod", C.c_int32), ("sps", C.c_uint32),
                ("proto", C.c_int32), ("flags", C.c_uint32), ("slot_filter", C.c_uint32), ("stream", C.c_void_p),
                ("rrc_taps", C.POINTER(C.c_float)), ("rrc_nzeros", C.c_uint32), ("rrc_gain", C.c_double)]


class DhError(RuntimeError):
    def __init__(self, code, what, detail=""):
        names = {-1: "DH_EINVAL", -2: "DH_ENOMEM", -3: "DH_EDEVICE", -4: "DH_ENODEV", -5: "DH_ECAPACITY"}
        super().__init__("%s failed: %s %s" % (what, names.get(code, code), detail))
        self.code = code


def declare(L, lenient=False):
    """Attach argtypes / restypes for every symbol of include/digiham_amd.h."""
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    L.dh_version.restype = C.c_char_p
    L.dh_last_error.restype = C.c_char_p
    L.dh_device_count.restype = C.c_int
    sig = {
        "dh_device_alloc": [C.c_int, sz, C.POINTER(vp)], "dh_device_free": [vp],
        "dh_copy_to_host": [vp, vp, sz], "dh_copy_to_device": [vp, vp, sz],
        "dh_hamming_7_4": [vp, vp, sz, vp], "dh_hamming_13_9": [vp, vp, sz, vp], "dh_hamming_15_11": [vp, vp, sz, vp],
        "dh_hamming_16_11": [vp, vp, sz, vp], "dh_quadratic_residue": [vp, vp, sz, vp],
        "dh_golay_20_8": [vp, vp, sz, vp], "dh_golay_24_12": [vp, vp, sz, vp], "dh_bch_31_21": [vp, vp, sz, vp],
        "dh_bptc_196_96": [vp, vp, vp, sz, vp],
        "dh_trellis": [vp, sz, C.c_int, vp, sz, vp, sz, vp],
        "dh_crc16": [vp, sz, C.c_int, vp, sz, vp],
        "dh_whitening": [vp, vp, sz, C.c_int, sz, vp],
        "dh_dvfilter_s16": [vp, vp, vp, sz, sz, sz, vp],
        "dh_frontend_s16": [vp, sz, vp, sz, vp, sz, sz, C.c_int, C.c_int, vp],
        "dh_debug_div_gain": [vp, vp, sz, C.c_int, vp],
        "dh_debug_div_const": [vp, vp, sz, C.c_uint, vp],
        "dh_debug_mfma_f16": [vp, vp, vp, vp, sz, vp],
        "dh_debug_f16_split": [vp, vp, vp, sz, C.c_float, vp],
        "dh_debug_copy": [vp, vp, sz, vp],
        "dh_engine_create": [C.POINTER(EngineConfig), C.POINTER(vp)],
        "dh_engine_reset": [vp], "dh_engine_set_slot_filter": [vp, u32],
        "dh_engine_reset_channel": [vp, u32], "dh_engine_set_slot_filter_channel": [vp, u32, u32],
        "dh_engine_push": [vp, vp, sz, sz], "dh_engine_push_host": [vp, vp, sz, sz],
        "dh_engine_push_ragged": [vp, vp, sz, vp, sz], "dh_engine_push_host_ragged": [vp, vp, sz, vp, sz],
        "dh_engine_push_symbols": [vp, vp, sz, vp],
        "dh_engine_filtered": [vp, C.POINTER(vp), C.POINTER(sz)],
        "dh_engine_symbols": [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp)],
        "dh_engine_frames": [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp)],
        "dh_engine_events": [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp)],
        "dh_engine_read_symbols": [vp, u32, vp, C.POINTER(sz)], "dh_engine_read_frames": [vp, u32, vp, C.POINTER(sz)],
        "dh_engine_read_events": [vp, u32, vp, C.POINTER(sz)], "dh_engine_read_filtered": [vp, u32, vp, C.POINTER(sz)],
        "dh_engine_sync": [vp],
        "dh_engine_timing_enable": [vp, u32],
        "dh_engine_timing_read": [vp, vp, vp, vp, C.POINTER(u32)],
        "dh_engine_timing_read_split": [vp, vp, vp, C.POINTER(u32)],
        "dh_engine_timing_stats": [vp, vp, vp],
        "dh_engine_debug_header": [vp, u32, vp],
    }
    for name, args in sig.items():
        if lenient and not hasattr(L, name):        # A/B build variants of older sources (tools/) may lack new entry points
            continue
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.dh_engine_destroy.argtypes = [vp]
    L.dh_engine_destroy.restype = None
    return L


EXPORTED_SYMBOLS = [
    "dh_version", "dh_last_error", "dh_device_count", "dh_device_alloc", "dh_device_free", "dh_copy_to_host",
    "dh_copy_to_device", "dh_hamming_7_4", "dh_hamming_13_9", "dh_hamming_15_11", "dh_hamming_16_11",
    "dh_quadratic_residue", "dh_golay_20_8", "dh_golay_24_12", "dh_bch_31_21", "dh_bptc_196_96", "dh_trellis", "dh_crc16",
    "dh_whitening", "dh_dvfilter_s16", "dh_frontend_s16", "dh_debug_div_gain", "dh_debug_div_const", "dh_debug_mfma_f16", "dh_debug_f16_split", "dh_debug_copy", "dh_engine_create", "dh_engine_destroy", "dh_engine_reset",
    "dh_engine_set_slot_filter", "dh_engine_reset_channel", "dh_engine_set_slot_filter_channel", "dh_engine_push", "dh_engine_push_host", "dh_engine_push_ragged", "dh_engine_push_host_ragged", "dh_engine_push_symbols",
    "dh_engine_filtered", "dh_engine_symbols", "dh_engine_frames", "dh_engine_events", "dh_engine_read_symbols",
    "dh_engine_read_frames", "dh_engine_read_events", "dh_engine_read_filtered", "dh_engine_sync",
    "dh_engine_timing_enable", "dh_engine_timing_read", "dh_engine_timing_read_split", "dh_engine_timing_stats", "dh_engine_debug_header",
]

_LIB = None


def load(path=None):
    """Load the gfx950 library.  Raises if it has not been built -- there is no fallback."""
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError("digiham_amd: %s is missing; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback." % p)
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the HIP runtime of the process, so it has to
    # be loaded before this library's NEEDED entry is resolved (two HIP runtimes in one process cannot see
    # each other's devices, streams or allocations).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = declare(C.CDLL(p), lenient=path is not None)
    if path is None:
        _LIB = L
    return L
