"""Generate tests/golden/nxdn_ref.npz (run in the development container only).

Every expected value comes from the REFERENCE's own NXDN classes: oracle/_ref/libdigiham_ref_nxdn.so is
src/nxdn_decoder/{scrambler,lich,sacch,facch1,trellis}.cpp compiled where they lie (they do not need csdr),
called through oracle/ref_nxdn.cpp.  Inputs: random dibits, encoded SACCH / FACCH1 blocks (digiham_amd/synth.py)
with 0..n flipped bits, and the four SACCH patterns the reference quotes from the NXDN "Common Air Interface
Test" document (src/nxdn_decoder/nxdn_phase.cpp:73-98), which form one VCALL superframe.

    python tests/golden/make_golden_nxdn.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from digiham_amd import synth           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CAI_SACCH = np.array([   # nxdn_phase.cpp:73-98, as transmitted (scrambled), 30 dibits each
    [3, 0, 3, 3, 2, 2, 0, 0, 2, 2, 1, 2, 3, 2, 2, 0, 2, 2, 0, 2, 0, 3, 1, 1, 1, 2, 3, 2, 2, 0],
    [3, 0, 1, 2, 3, 1, 2, 3, 2, 3, 0, 0, 3, 2, 2, 3, 0, 3, 2, 2, 1, 0, 0, 2, 1, 2, 2, 2, 2, 0],
    [1, 2, 0, 3, 2, 2, 0, 1, 2, 3, 1, 0, 2, 2, 2, 0, 0, 1, 2, 2, 2, 0, 3, 2, 0, 2, 2, 0, 0, 0],
    [1, 0, 0, 2, 2, 0, 2, 0, 0, 3, 0, 0, 0, 2, 2, 3, 0, 0, 0, 2, 3, 1, 0, 0, 1, 3, 3, 2, 0, 2]], np.uint8)


def flip(dibits, rng, nflips):
    d = np.array(dibits, np.uint8)
    for bp in rng.choice(2 * len(d), nflips, replace=False):
        d[bp // 2] ^= 2 >> (bp % 2)
    return d


def main():
    assert O.ref_nxdn() is not None, "build oracle/_ref first (make -C oracle)"
    rng = np.random.default_rng(20260929)
    v = {}
    d = rng.integers(0, 4, (256, 182), dtype=np.uint8)
    v["scr_in"], v["scr_out"] = d, np.stack([O.nxdn_scramble(r, "ref") for r in d])
    l = np.concatenate([rng.integers(0, 4, (192, 8), dtype=np.uint8), np.array([synth.nxdn_lich_dibits(x) for x in range(128)], np.uint8)])
    v["lich_in"], v["lich_out"] = l, np.array([O.nxdn_lich(r, "ref") for r in l], np.int32)
    for nb in (72, 192):
        p = rng.integers(0, 256, (384, nb // 8), dtype=np.uint8)
        r = [O.nxdn_trellis(x, nb, "ref") for x in p]
        v["trellis%d_in" % nb], v["trellis%d_out" % nb] = p, np.stack([a for a, _ in r])
        v["trellis%d_metric" % nb] = np.array([m for _, m in r], np.uint32)
    sac = [flip(synth.nxdn_sacch_dibits(i & 3, int(rng.integers(0, 64)), list(rng.integers(0, 2, 18))), rng, i % 7) for i in range(512)]
    sac += list(rng.integers(0, 4, (256, 30), dtype=np.uint8))
    # the CAI patterns are descrambled the way the frame does it: scrambler positions 8..37 (after the LICH)
    for row in CAI_SACCH:
        sac.append(O.nxdn_scramble(np.concatenate([np.zeros(8, np.uint8), row]), "ref")[8:])
    sac = np.array(sac, np.uint8)
    r = [O.nxdn_sacch(x, "ref") for x in sac]
    v["sacch_in"], v["sacch_ok"] = sac, np.array([ok for ok, _ in r], np.uint8)
    v["sacch_out"] = np.stack([o if ok else np.zeros(5, np.uint8) for ok, o in r])
    fa = [flip(synth.nxdn_facch1_dibits(list(rng.integers(0, 2, 80))), rng, i % 13) for i in range(512)]
    fa += list(rng.integers(0, 4, (128, 72), dtype=np.uint8))
    fa = np.array(fa, np.uint8)
    r = [O.nxdn_facch1(x, "ref") for x in fa]
    v["facch1_in"], v["facch1_ok"] = fa, np.array([ok for ok, _ in r], np.uint8)
    v["facch1_out"] = np.stack([o if ok else np.zeros(12, np.uint8) for ok, o in r])
    v["cai_sacch_tx"] = CAI_SACCH
    np.savez_compressed(os.path.join(OUT, "nxdn_ref.npz"), **v)
    print("sacch ok %d / %d, facch1 ok %d / %d" % (v["sacch_ok"].sum(), len(sac), v["facch1_ok"].sum(), len(fa)))
    print("CAI superframe fragments:", [bytes(o).hex() for o in v["sacch_out"][-4:]])


if __name__ == "__main__":
    main()
