"""Pin of the RRC coefficient tables (run in the development container only: it reads /root/reference).

The product (digiham_amd/csrc/rrc_taps.h) and the oracle (oracle/dsp.c includes that header) share ONE copy of the two
mkshape tables -- a typo there would pass every parity test.  This script parses the numbers out of the reference's own
source text, src/rrc_filter/rrc_filter.cpp (NarrowRrcFilter: 161 taps and its gain, WideRrcFilter: 81 taps and its gain), and
commits their SHA-256 (float32 bit patterns, little endian, in table order; gains as float64) -- data, not source text.
tests/test_numerics.py::test_rrc_tables_equal_the_references hashes what dh_rrc_expand_taps() produces against it.

    python tests/golden/make_golden_taps.py
"""
import hashlib
import json
import os
import re

import numpy as np

REF = "/root/reference/src/rrc_filter/rrc_filter.cpp"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rrc_taps_ref_hashes.json")


def table(src, cls):
    body = src[src.index(cls + "::" + cls):]
    nz, gain = re.search(r"RrcFilter\(\s*(\d+)\s*,\s*([0-9.eE+-]+)\s*,", body).groups()
    nums = body[body.index("{", body.index("(const float[])")) + 1:]
    nums = nums[:nums.index("}")]
    taps = np.array([float(t) for t in re.findall(r"[+-]?\d+\.\d+(?:[eE][+-]?\d+)?", nums)], np.float32)
    assert taps.size == int(nz) + 1, (cls, taps.size, nz)
    return int(nz), float(gain), taps


def main():
    src = open(REF).read()
    out = {"source": "numbers parsed from the reference's src/rrc_filter/rrc_filter.cpp:36-115 by tests/golden/make_golden_taps.py"}
    for key, cls in (("narrow", "NarrowRrcFilter"), ("wide", "WideRrcFilter")):
        nz, gain, taps = table(src, cls)
        out[key] = {"nzeros": nz, "taps": int(taps.size), "gain_float64_hex": np.float64(gain).tobytes().hex(),
                    "taps_float32_sha256": hashlib.sha256(taps.astype("<f4").tobytes()).hexdigest(),
                    "sum_float64": float(taps.astype(np.float64).sum())}
        print(key, nz, gain, out[key]["taps_float32_sha256"][:16])
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
