"""RRC coefficient tables, parsed from csrc/rrc_taps.h (the single source of the constants)."""
import os
import re

import numpy as np

_HDR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "rrc_taps.h")


def _half(name):
    src = open(_HDR).read()
    body = src[src.index(name):]
    body = body[body.index("{") + 1:body.index("}")]
    return np.array([float(t.rstrip("f")) for t in re.findall(r"[+-]\d\.\d+f", body)], np.float32)


def wide():
    h = _half("dh_rrc_wide_half")
    assert h.size == 41
    return np.concatenate([h, h[-2::-1]])


def narrow():
    h = _half("dh_rrc_narrow_half")
    assert h.size == 81
    return np.concatenate([h, h[-2::-1]])


WIDE_GAIN = 8.337797030e+00
NARROW_GAIN = 1.667711971e+01
