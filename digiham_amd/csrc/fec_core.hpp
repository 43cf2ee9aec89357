// fec_core.hpp -- block codes, CRC-16, PN9 as lane-local functions.
//
// What the reference does with a bit-serial syndrome loop plus a linear scan of a generated
// {syndrome, pattern} list (e.g. src/dmr_decoder/golay_20_8.c:1403-1435) is done here with
//   syndrome = for each parity-check row: parity(popcount(word & row))      (v_bcnt / s_bcnt1)
//   pattern  = dense LUT[syndrome]                                          (LDS / L2 resident)
// The dense LUT is equivalent to the reference's first-match scan because no two correctable
// error patterns of these codes share a syndrome (d_min >= 2t+1); the tables are regenerated
// at library start-up from the generator matrices (fec_tables.cpp), not copied.
#pragma once

#include "dh_portable.hpp"

struct DhCode {
    // parity-check rows as masks over the n-bit word (bit n-1 = first transmitted column)
    uint32_t h[12];
    int n, k;
};

struct DhFecTables {
    DhCode h74, h139, h1511, h1611, qr, g208, g2412, bch3121;
    uint8_t  lut_h74[8];
    uint16_t lut_h139[16];
    uint16_t lut_h1511[16];
    uint16_t lut_h1611[32];
    uint16_t lut_qr[512];
    uint32_t lut_g208[4096];
    uint32_t lut_g2412[4096];
    uint32_t lut_bch3121[1024];
};

// syndrome in the reference's bit order: first parity-check row ends up in the MSB
// (hamming_13_9.c:52-68 and siblings)
DH_HD uint32_t dh_syndrome(const DhCode& c, uint32_t word) {
    uint32_t s = 0;
    const int r = c.n - c.k;
    for (int i = 0; i < r; i++) s = (s << 1) | (uint32_t) (dh_popc32(word & c.h[i]) & 1);
    return s;
}

template <typename LutT>
DH_HD bool dh_block_decode(const DhCode& c, const LutT* lut, uint32_t& word) {
    const uint32_t s = dh_syndrome(c, word);
    if (s == 0) return true;
    const uint32_t p = lut[s];
    if (p == 0) return false;          // syndrome of no pattern of weight <= t: uncorrectable
    word ^= p;
    return true;
}

// the same with the number of parity checks known at compile time: straight-line code (the run-time loop above compiles to an
// unrolled-by-eight loop plus a remainder loop, with the row count fetched from the table first)
template <int R, typename LutT>
DH_HD bool dh_block_decode_rows(const DhCode& c, const LutT* lut, uint32_t& word) {
    uint32_t h[R];
#pragma unroll
    for (int i = 0; i < R; i++) h[i] = c.h[i];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < R; i++) s = (s << 1) | (uint32_t) (dh_popc32(word & h[i]) & 1);
    if (s == 0) return true;
    const uint32_t p = lut[s];
    if (p == 0) return false;
    word ^= p;
    return true;
}

// Same decode for a wave-uniform word with the wavefront's help: lane i evaluates parity-check row i, a vote
// collects the syndrome (first row in the MSB), the pattern lookup is one load.  R = n - k.
template <int R, typename LutT>
DH_HD bool dh_block_decode_wave(const DhCode& c, const LutT* lut, uint32_t& word) {
    uint64_t votes = 0;
    DH_FOR_LANES(lane) {
        bool b = false;
        if (lane < R) b = (dh_popc32(word & c.h[lane]) & 1) != 0;
        DH_BALLOT_ACC(votes, b, lane);
    }
    const uint32_t s = dh_brev32((uint32_t) votes) >> (32 - R);
    if (s == 0) return true;
    const uint32_t p = dh_uniform((uint32_t) lut[s]);
    if (p == 0) return false;
    word ^= p;
    return true;
}

// (BPTC(196,96): decoder_core.hpp, dh_dmr_bptc_lane -- one block per lane, the column code bit-sliced; the batch entry dh_bptc_196_96 runs it too)

// ------------------------------------------------------------------ CRC-16 / PN9
// reference: src/ysf_decoder/crc16.c:3-18
DH_HD uint16_t dh_crc16(const uint8_t* data, int count) {
    uint32_t crc = 0;
    for (int k = 0; k < count; k++) {
        crc ^= (uint32_t) data[k] << 8;
        for (int i = 0; i < 8; i++) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x1021u) : (crc << 1);
    }
    return (uint16_t) (~crc & 0xFFFFu);
}

// reference: src/ysf_decoder/whitening.c:6-22; one PN9 step returns the whitening bit
DH_HD uint32_t dh_pn9_step(uint32_t& wsr) {
    const uint32_t wb = wsr & 1u;
    const uint32_t fb = ((wsr >> 4) & 1u) ^ wb;
    wsr = ((wsr & 0x1FEu) >> 1) | (fb << 8);
    return wb;
}

DH_HD void dh_whiten(const uint8_t* in, uint8_t* out, int n_bits) {
    uint32_t wsr = 0x1C9u;
    const int nbytes = (n_bits + 7) / 8;
    for (int b = 0; b < nbytes; b++) {
        uint32_t m = 0;
        for (int i = 0; i < 8; i++) {
            const int bit = b * 8 + i;
            uint32_t wb = 0;
            if (bit < n_bits) wb = dh_pn9_step(wsr);
            m = (m << 1) | wb;
        }
        // bits past n_bits in the last byte come out as 0, as in the reference (only the first n are copied)
        const uint32_t keep = (b * 8 + 8 <= n_bits) ? 0xFFu : (0xFFu << (8 - (n_bits - b * 8))) & 0xFFu;
        out[b] = (uint8_t) ((in[b] ^ m) & keep);
    }
}

// The same two functions for the frame decoders, where they run wave-uniformly on a path whose cost is its VALU
// instruction count: CRC byte-wise through a compile-time table (the bit loop above costs ~4 instructions per BIT),
// de-whitening as an XOR with the PN9 sequence packed at compile time (it is a constant: fixed seed, whitening.c:8).
struct DhCrc16Table {
    uint16_t t[256];
    constexpr DhCrc16Table(): t() {
        for (int b = 0; b < 256; b++) {
            uint32_t crc = (uint32_t) b << 8;
            for (int i = 0; i < 8; i++) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x1021u) : (crc << 1);
            t[b] = (uint16_t) crc;
        }
    }
};
struct DhPn9Bytes {
    uint8_t b[24];                                    // 192 whitening bits, first bit in the MSB of b[0]
    constexpr DhPn9Bytes(): b() {
        uint32_t wsr = 0x1C9u;
        for (int k = 0; k < 24; k++) {
            uint32_t m = 0;
            for (int i = 0; i < 8; i++) {
                const uint32_t wb = wsr & 1u, fb = ((wsr >> 4) & 1u) ^ wb;
                wsr = ((wsr & 0x1FEu) >> 1) | (fb << 8);
                m = (m << 1) | wb;
            }
            b[k] = (uint8_t) m;
        }
    }
};
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
__device__ static constexpr DhCrc16Table dh_crc16_table{};
__device__ static constexpr DhPn9Bytes dh_pn9_bytes{};
#else
static constexpr DhCrc16Table dh_crc16_table{};
static constexpr DhPn9Bytes dh_pn9_bytes{};
#endif

// four bytes of the whitening sequence as a little-endian word (byte 4 i in bits 0..7)
constexpr uint32_t dh_pn9_le_word(int i) {
    constexpr DhPn9Bytes pn{};
    return (uint32_t) pn.b[4 * i] | (uint32_t) pn.b[4 * i + 1] << 8 | (uint32_t) pn.b[4 * i + 2] << 16 | (uint32_t) pn.b[4 * i + 3] << 24;
}

DH_HD uint16_t dh_crc16_bytewise(const uint8_t* data, int count) {
    uint32_t crc = 0;
    for (int k = 0; k < count; k++) crc = ((crc << 8) & 0xFFFFu) ^ dh_crc16_table.t[((crc >> 8) ^ data[k]) & 0xFFu];
    return (uint16_t) (~crc & 0xFFFFu);
}

DH_HD void dh_whiten_packed(const uint8_t* in, uint8_t* out, int n_bits) {     // n_bits <= 192
    const int nbytes = (n_bits + 7) / 8;
    for (int b = 0; b < nbytes; b++) {
        const uint32_t keep = (b * 8 + 8 <= n_bits) ? 0xFFu : (0xFFu << (8 - (n_bits - b * 8))) & 0xFFu;
        out[b] = (uint8_t) ((in[b] ^ dh_pn9_bytes.b[b]) & keep);
    }
}

// rate-1/2 K=5 encoder output of the transition leaving `state` with input `bit`
// (G1 = 1+D^3+D^4, G2 = 1+D+D^2+D^4; equals the table at src/ysf_decoder/trellis.c:8-25)
DH_HD uint32_t dh_trellis_out(uint32_t state, uint32_t bit) {
    const uint32_t s0 = state & 1u, s1 = (state >> 1) & 1u, s2 = (state >> 2) & 1u, s3 = (state >> 3) & 1u;
    return ((bit ^ s1 ^ s0) << 1) | (bit ^ s3 ^ s2 ^ s0);
}
