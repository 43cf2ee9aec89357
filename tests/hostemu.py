"""CPU wave-emulation context for the non-GPU test tier (TEST INFRASTRUCTURE ONLY).

Builds tests/host_harness/libdh_hostemu.so (the kernel bodies compiled with g++, lanes as
loops) and wraps it in a digiham_amd.api.Context whose "device memory" is numpy.  The
digiham_amd package itself never loads this library.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from digiham_amd import _capi, api

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_harness")
_SO = os.path.join(_DIR, "libdh_hostemu.so")
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(force=False):
    srcs = [os.path.join(_DIR, "harness.cpp")] + [os.path.join(_ROOT, "digiham_amd", "csrc", f)
                                                  for f in os.listdir(os.path.join(_ROOT, "digiham_amd", "csrc"))]
    srcs.append(os.path.join(_ROOT, "include", "digiham_amd.h"))
    def fresh():
        return os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)
    if not force and fresh():
        return _SO
    # one build at a time (pytest-xdist workers all come through here), into a temporary name: nobody loads half a library
    import fcntl
    with open(os.path.join(_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not fresh():
            tmp = _SO + ".%d.tmp" % os.getpid()
            subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                            "-Wno-subobject-linkage", os.path.join(_DIR, "harness.cpp"), "-o", tmp], check=True)
            os.replace(tmp, _SO)
    return _SO


class NumpyMemory:
    index = 0

    def from_numpy(self, a):
        return np.ascontiguousarray(a).copy()

    def zeros(self, shape, dtype):
        return np.zeros(shape, dtype)

    def to_numpy(self, t, dtype=None):
        return t.view(dtype) if dtype is not None else t

    def ptr(self, t):
        return C.c_void_p(t.ctypes.data)

    def stream(self):
        return C.c_void_p(0)

    def is_device_array(self, x):
        return isinstance(x, np.ndarray) and x.flags["C_CONTIGUOUS"]


_CTX = None


def context():
    global _CTX
    if _CTX is None:
        lib = _capi.declare(C.CDLL(build()))
        _CTX = api.Context(lib=lib, mem=NumpyMemory())
    return _CTX
