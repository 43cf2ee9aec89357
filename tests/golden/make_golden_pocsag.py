"""Generate tests/golden/pocsag_ref.npz (run in the development container only): BCH(31,21) inputs and the outputs of the
REFERENCE's own src/pocsag_decoder/bch_31_21.c, compiled in place into oracle/_ref/libdigiham_ref_fec.so.

    python tests/golden/make_golden_pocsag.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from digiham_amd import synth           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    assert O.ref() is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20260930)
    cw = np.array([synth.bch_31_21_encode(int(x)) for x in rng.integers(0, 1 << 21, 6000)], np.uint32)
    err = np.zeros_like(cw)
    for i in range(len(cw)):
        for bp in rng.choice(31, i % 5, replace=False):          # 0..4 flipped bits
            err[i] |= np.uint32(1 << int(bp))
    words = np.concatenate([cw ^ err, rng.integers(0, 1 << 31, 6000, dtype=np.uint32),
                            np.array([1 << a for a in range(31)] + [(1 << a) | (1 << b) for a in range(31) for b in range(a)], np.uint32)])
    out, ok = O.block_decode("bch_31_21", words, "ref")
    np.savez_compressed(os.path.join(OUT, "pocsag_ref.npz"), bch_in=words, bch_out=np.where(ok == 1, out, 0).astype(np.uint32), bch_ok=ok)
    print(len(words), int(ok.sum()))


if __name__ == "__main__":
    main()
