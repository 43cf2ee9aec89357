#!/usr/bin/env python3
"""How far from the ends of a block must a single wrong dibit sit for the reference's Viterbi decoder (src/ysf_decoder/trellis.c:32-109:
every start state at metric 0, no tail, best end state = smallest metric) to return the codeword one dibit away -- behind the
single-dibit repair of decoder_core.hpp (dh_ysf_clean100).

The code is linear (G1 = 1 + D^3 + D^4, G2 = 1 + D + D^2 + D^4), so the distance between two trellis paths is the output weight of
their DIFFERENCE path, which is "active" (non-zero state, or a non-zero input leaving state 0) exactly where the two paths have not
merged.  With r = c + e, e a single dibit of weight w <= 2 at position p, a competitor c' has d(r, c') = w + d(c, c') if its
difference path is not active at p, and d(r, c') >= d(c, c') - w otherwise; the decoder is certain to return c when d(c, c') >= 2 w + 1
= 5 for every difference path active at p:
  * a detour that leaves and rejoins inside the block weighs at least d_free;
  * one that is active from the block's start (the paths begin in different states) through step p weighs at least f_start(p + 1);
  * one that diverges at or before p and is still active at the block's end weighs at least f_end(N - p).
This script computes d_free, f_start(L) and f_end(L) by dynamic programming over the 16 difference states and prints the smallest
margin M with f_start(M + 1) >= 5 and f_end(M + 1) >= 5: positions M <= p <= N - 1 - M are safe."""

INF = 10 ** 9


def out_weight(state, bit):
    s0, s1, s2, s3 = state & 1, (state >> 1) & 1, (state >> 2) & 1, (state >> 3) & 1
    return (bit ^ s1 ^ s0) + (bit ^ s3 ^ s2 ^ s0)           # G1, G2 (dh_trellis_out)


def step(state, bit):
    return (bit << 3) | (state >> 1)


def f_start(L):
    """min weight over L steps of a difference path whose state before each of the L steps is non-zero"""
    best = {s: 0 for s in range(1, 16)}
    for _ in range(L):
        nxt = {}
        for s, w in best.items():
            if s == 0:
                continue
            for b in (0, 1):
                t, ww = step(s, b), w + out_weight(s, b)
                if ww < nxt.get(t, INF):
                    nxt[t] = ww
        best = nxt
    return min(best.values())


def f_end(L):
    """min weight over L steps of a difference path that leaves state 0 with input 1 at the first step and is active in all L"""
    best = {step(0, 1): out_weight(0, 1)}
    for _ in range(L - 1):
        nxt = {}
        for s, w in best.items():
            if s == 0:
                continue                                      # merged: no longer active
            for b in (0, 1):
                t, ww = step(s, b), w + out_weight(s, b)
                if ww < nxt.get(t, INF):
                    nxt[t] = ww
        best = nxt
    live = [w for s, w in best.items()]                       # (the state AFTER the last step may be zero: the path was active in it)
    return min(live) if live else INF


def d_free():
    best, done = {step(0, 1): out_weight(0, 1)}, INF
    for _ in range(64):
        nxt = {}
        for s, w in best.items():
            for b in (0, 1):
                t, ww = step(s, b), w + out_weight(s, b)
                if t == 0:
                    done = min(done, ww)
                elif ww < nxt.get(t, INF):
                    nxt[t] = ww
        best = nxt
    return done


if __name__ == "__main__":
    print("d_free =", d_free())
    for L in range(1, 24):
        print("L = %2d   f_start = %d   f_end = %d" % (L, f_start(L), f_end(L)))
    M = next(m for m in range(1, 64) if f_start(m + 1) >= 5 and f_end(m + 1) >= 5)
    print("margin M =", M, "(positions M .. N - 1 - M)")
