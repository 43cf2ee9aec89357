// fec_tables.hpp -- host-side construction of the parity-check rows and dense syndrome LUTs.
//
// Codes (generator matrices G = [I | P] from ETSI TS 102 361-1 Annex B.3 and the YSF spec,
// the same ones the reference quotes in the comment blocks of src/dmr_decoder/hamming_*.c,
// golay_20_8.c, quadratic_residue.c and src/ysf_decoder/golay_24_12.c):
//   Hamming(7,4), (13,9), (15,11), (16,11)  t = 1
//   QR(16,7,6)                               t = 2
//   Golay(20,8,7), Golay(24,12,8)            t = 3
// LUT[s] = the unique error pattern of weight <= t with syndrome s, or 0 if there is none.
#pragma once

#include "fec_core.hpp"

namespace dh {

inline void build_code(DhCode& c, int n, int k, const uint16_t* p) {
    c.n = n; c.k = k;
    const int r = n - k;
    for (int row = 0; row < 12; row++) c.h[row] = 0;
    for (int row = 0; row < r; row++) {
        uint32_t m = 1u << (r - 1 - row);                       // identity part
        for (int j = 0; j < k; j++)
            if ((p[j] >> (r - 1 - row)) & 1) m |= 1u << (n - 1 - j);   // P^T part
        c.h[row] = m;
    }
}

template <typename LutT>
inline void build_lut(const DhCode& c, int t, LutT* lut, int lut_size) {
    for (int i = 0; i < lut_size; i++) lut[i] = 0;
    // enumerate error patterns of weight 1..t by position tuples
    const int n = c.n;
    for (int a = 0; a < n; a++) {
        const uint32_t ea = 1u << a;
        lut[dh_syndrome(c, ea)] = (LutT) ea;
        if (t < 2) continue;
        for (int b = a + 1; b < n; b++) {
            const uint32_t eb = ea | (1u << b);
            lut[dh_syndrome(c, eb)] = (LutT) eb;
            if (t < 3) continue;
            for (int d = b + 1; d < n; d++) {
                const uint32_t ed = eb | (1u << d);
                lut[dh_syndrome(c, ed)] = (LutT) ed;
            }
        }
    }
}

// cyclic code from its generator polynomial: parity-check row `row`, bit j = coefficient (r-1-row) of x^j mod g
// (BCH(31,21): g = x^10+x^9+x^8+x^6+x^5+x^3+1, CCIR 584; gives the rows listed at src/pocsag_decoder/bch_31_21.c:3-14)
inline void build_cyclic_code(DhCode& c, int n, int k, uint32_t gen) {
    c.n = n; c.k = k;
    const int r = n - k;
    for (int row = 0; row < 12; row++) c.h[row] = 0;
    uint32_t rem = 1;
    for (int j = 0; j < n; j++) {
        for (int row = 0; row < r; row++) if ((rem >> (r - 1 - row)) & 1u) c.h[row] |= 1u << j;
        rem <<= 1;
        if (rem & (1u << r)) rem ^= gen;
    }
}

inline void build_fec_tables(DhFecTables& T) {
    static const uint16_t P_H74[4] = { 0x5, 0x7, 0x6, 0x3 };
    static const uint16_t P_H139[9] = { 0xF, 0xE, 0x7, 0xA, 0x5, 0xB, 0xC, 0x6, 0x3 };
    static const uint16_t P_H1511[11] = { 0x9, 0xD, 0xF, 0xE, 0x7, 0xA, 0x5, 0xB, 0xC, 0x6, 0x3 };
    static const uint16_t P_H1611[11] = { 0x13, 0x1A, 0x1F, 0x1C, 0x0E, 0x15, 0x0B, 0x16, 0x19, 0x0D, 0x07 };
    static const uint16_t P_G2412[12] = { 0xC75, 0x63B, 0xF68, 0x7B4, 0x3DA, 0xD99, 0x6CD, 0x367, 0xDC6, 0xA97, 0x93E, 0x8EB };
    static const uint16_t P_QR[7] = { 0x04F, 0x11E, 0x1B7, 0x1E2, 0x1C9, 0x0E5, 0x073 };
    build_code(T.h74, 7, 4, P_H74);
    build_code(T.h139, 13, 9, P_H139);
    build_code(T.h1511, 15, 11, P_H1511);
    build_code(T.h1611, 16, 11, P_H1611);
    build_code(T.qr, 16, 7, P_QR);
    build_code(T.g208, 20, 8, P_G2412 + 4);      // Golay(20,8) = rows 4..11 of the (24,12) P matrix
    build_code(T.g2412, 24, 12, P_G2412);
    build_lut(T.h74, 1, T.lut_h74, 8);
    build_lut(T.h139, 1, T.lut_h139, 16);
    build_lut(T.h1511, 1, T.lut_h1511, 16);
    build_lut(T.h1611, 1, T.lut_h1611, 32);
    build_lut(T.qr, 2, T.lut_qr, 512);
    build_lut(T.g208, 3, T.lut_g208, 4096);
    build_lut(T.g2412, 3, T.lut_g2412, 4096);
    build_cyclic_code(T.bch3121, 31, 21, 0x769u);
    build_lut(T.bch3121, 2, T.lut_bch3121, 1024);
}

}  // namespace dh
