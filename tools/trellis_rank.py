#!/usr/bin/env python3
"""The map (start state, message) -> N dibits of the rate-1/2 K = 5 code (G1 = 1 + D^3 + D^4, G2 = 1 + D + D^2 + D^4) is injective
for N >= 4: rank over GF(2) of the 2 N x (N + 4) encoding matrix.  Behind the clean-codeword shortcut of decoder_core.hpp
(dh_viterbi_clean): a received word with zero syndrome has exactly one path of metric 0."""
import numpy as np


def rank_gf2(M):
    M = M.copy() % 2
    r = 0
    rows, cols = M.shape
    for c in range(cols):
        p = next((i for i in range(r, rows) if M[i, c]), None)
        if p is None:
            continue
        M[[r, p]] = M[[p, r]]
        for i in range(rows):
            if i != r and M[i, c]:
                M[i] ^= M[r]
        r += 1
    return r


def encoding_matrix(N):
    M = np.zeros((2 * N, N + 4), np.uint8)          # unknowns u_(-4) .. u_(N-1); rows h_0, l_0, h_1, l_1, ...
    for t in range(N):
        for d in (0, 3, 4):
            M[2 * t, t - d + 4] ^= 1
        for d in (0, 1, 2, 4):
            M[2 * t + 1, t - d + 4] ^= 1
    return M


if __name__ == "__main__":
    for N in list(range(1, 33)) + [100, 180, 192]:
        r = rank_gf2(encoding_matrix(N))
        print("N = %3d: rank %3d of %3d%s" % (N, r, N + 4, "" if r == N + 4 else "   (not injective)"))
