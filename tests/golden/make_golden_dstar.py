"""Generate tests/golden/dstar_ref.npz (run in the development container only): inputs and the outputs of the REFERENCE's
own src/dstar_decoder/{scrambler,crc}.cpp, compiled in place into oracle/_ref/libdigiham_ref_dstar.so.

    python tests/golden/make_golden_dstar.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    assert O.ref_dstar() is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20261001)
    scr_in = rng.integers(0, 2, (16, 660)).astype(np.uint8)
    scr_in[0] = 0                                                       # row 0: the bare whitening sequence
    scr_out = np.stack([O.dstar_scramble(r, "ref") for r in scr_in])
    # CRC: byte strings (zero padded to 64) with their lengths and a candidate checksum; the reference says valid / not
    n = 600
    data = rng.integers(0, 256, (n, 64)).astype(np.uint8)
    lens = rng.integers(1, 65, n).astype(np.uint32)
    lens[:50] = 39                                                      # the radio header's span
    cand = np.zeros(n, np.uint16)
    for i in range(n):
        c = O.dstar_crc(data[i, :lens[i]])                              # the restatement proposes, the reference disposes
        cand[i] = c if i % 3 else c ^ (1 << int(rng.integers(0, 16)))
    valid = np.array([O.dstar_crc_valid(data[i, :lens[i]], int(cand[i]), "ref") for i in range(n)], np.uint8)
    assert valid[np.arange(n) % 3 != 0].all() and not valid[np.arange(n) % 3 == 0].any()
    np.savez_compressed(os.path.join(OUT, "dstar_ref.npz"), scr_in=scr_in, scr_out=scr_out, crc_data=data, crc_len=lens,
                        crc_cand=cand, crc_valid=valid)
    print(scr_in.shape, n, int(valid.sum()))


if __name__ == "__main__":
    main()
