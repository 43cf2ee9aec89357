/*
 * nxdn.c -- oracle restatement of the NXDN48 frame elements (TEST INFRASTRUCTURE ONLY):
 * scrambler, LICH, SACCH, FACCH1 and the NXDN flavour of the rate-1/2 K=5 Viterbi decoder.
 *
 * PINNED: these functions are checked bit-for-bit against the reference's own
 * src/nxdn_decoder/{scrambler,lich,sacch,facch1,trellis}.cpp, compiled in place into
 * oracle/_ref/libdigiham_ref_nxdn.so (they do not depend on csdr), on the vectors of
 * tests/golden/nxdn_ref.npz (tests/test_oracle.py).  The frame state machine that calls them
 * (nxdn_phase.cpp, needs csdr) is restated in decoders.c and is PARITY UNPINNED.
 */
#include "dh_oracle.h"
#include <string.h>

/* scrambler.cpp:9-25: a 9-bit LFSR flips the high bit of a dibit when its output is 1 */
void orc_nxdn_scramble(uint16_t* shift_register, const uint8_t* input, uint8_t* output, size_t len) {
    memset(output, 0, len);
    uint16_t sr = *shift_register;
    for (size_t i = 0; i < len; i++) {
        unsigned wb = sr & 1u;
        output[i] = (uint8_t) ((input[i] & 3u) ^ (wb << 1));
        wb = ((sr >> 4) & 1u) ^ wb;
        sr = (uint16_t) (((sr & 0x1FEu) >> 1) | (wb << 8));
    }
    *shift_register = sr;
}

/* lich.cpp:5-31: 8 dibits -> 7-bit LICH + parity over the 4 MSBs; returns -1 on parity error */
int orc_nxdn_lich_parse(const uint8_t* raw) {
    uint8_t bits[8];
    for (int i = 0; i < 8; i++) bits[i] = (raw[i] >> 1) & 1u;
    uint8_t check = 0;
    for (int i = 0; i < 4; i++) check ^= bits[i];
    if (bits[7] != check) return -1;
    int lich = 0;
    for (int i = 0; i < 7; i++) lich |= bits[i] << (6 - i);
    return lich;
}

/* trellis.cpp:8-25 */
static const uint8_t nxdn_transitions[16][2] = {
    {0, 3}, {3, 0}, {2, 1}, {1, 2}, {1, 2}, {2, 1}, {3, 0}, {0, 3},
    {1, 2}, {2, 1}, {3, 0}, {0, 3}, {0, 3}, {3, 0}, {2, 1}, {1, 2},
};

/* trellis.cpp:27-101: len = number of input BITS (two per transition), packed MSB first; uint16 metrics
 * all starting at 0; during the first four steps a state whose bits overlap `blocked` only considers
 * its k = 0 predecessor; k = 0 wins ties; the best end state is the lowest index among the minimum.
 * The survivors are kept by register exchange here as in the reference (bit strings per state). */
unsigned orc_nxdn_trellis_decode(const uint8_t* input, uint8_t* output, size_t len) {
    const size_t data_size = (len + 15) / 16;
    uint16_t metric[16], next_metric[16];
    uint8_t data[16][16], next_data[16][16];
    uint8_t blocked = 0xF;
    memset(metric, 0, sizeof(metric));
    memset(data, 0, sizeof(data));
    for (size_t pos = 0; pos < len / 2; pos++) {
        const uint8_t in_transition = (uint8_t) ((input[pos / 4] >> (2 * (3 - pos % 4))) & 3u);
        const size_t outpos = pos / 8; const unsigned outshift = 7 - pos % 8;
        for (int i = 0; i < 16; i++) {
            uint16_t best_metric = 0xFFFF; uint8_t selected = 0xFF;
            const uint8_t outbit = (uint8_t) ((i & 8) >> 3);
            const int limit = 1 + ((i & blocked) == 0);
            for (int k = 0; k < limit; k++) {
                const uint8_t previous_state = (uint8_t) (((i << 1) & 0xE) | k);
                const uint8_t transition = nxdn_transitions[previous_state][outbit];
                const uint16_t m = (uint16_t) (metric[previous_state] + orc_hamming_distance(&in_transition, &transition, 1));
                if (k == 0 || m < best_metric) { best_metric = m; selected = previous_state; }
            }
            next_metric[i] = best_metric;
            memcpy(next_data[i], data[selected], data_size);
            next_data[i][outpos] |= (uint8_t) (outbit << outshift);
        }
        memcpy(metric, next_metric, sizeof(metric));
        memcpy(data, next_data, sizeof(data));
        blocked = (uint8_t) ((blocked << 1) & 0xF);
    }
    int best = 0;
    for (int i = 1; i < 16; i++) if (metric[i] < metric[best]) best = i;
    memcpy(output, data[best], data_size);
    return metric[best];
}

/* sacch.cpp:45-68: 12 x 5 bit de-interleave, then 60 -> 72 bits with a 0 in every 6th position */
static void sacch_deinterleave(const uint8_t* in, uint8_t* out) {
    memset(out, 0, 30);
    for (int i = 0; i < 12; i++) for (int k = 0; k < 5; k++) {
        const int inpos = i * 5 + k, outpos = k * 12 + i;
        out[outpos / 2] |= (uint8_t) (((in[inpos / 2] >> (1 - inpos % 2)) & 1u) << (1 - outpos % 2));
    }
}
static void sacch_inflate(const uint8_t* input, uint8_t* output) {
    memset(output, 0, 9);
    int pos = 0;
    for (int i = 0; i < 72; i++) {
        unsigned x = 0;
        if ((i + 1) % 6 != 0) { x = (input[pos / 2] >> (1 - pos % 2)) & 1u; pos++; }
        output[i / 8] |= (uint8_t) (x << (7 - i % 8));
    }
}
/* sacch.cpp:70-84: CRC-6 over the first 26 bits, compared with in[3] & 0x3F */
static int sacch_check_crc(const uint8_t* in) {
    uint8_t crc = 0x3F;
    for (int i = 0; i < 26; i++) {
        const unsigned cb = ((crc >> 5) & 1u) ^ ((in[i / 8] >> (7 - i % 8)) & 1u);
        if (cb) crc ^= 0x13;
        crc = (uint8_t) (((crc << 1) & 0x3E) | cb);
    }
    return (in[3] & 0x3F) == crc;
}
/* Sacch::parse (sacch.cpp:24-43): 30 dibits -> 5 bytes; returns 1 when the CRC holds */
int orc_nxdn_sacch_parse(const uint8_t* dibits30, uint8_t* out5) {
    uint8_t deinterleaved[30], inflated[9];
    sacch_deinterleave(dibits30, deinterleaved);
    sacch_inflate(deinterleaved, inflated);
    memset(out5, 0, 5);
    orc_nxdn_trellis_decode(inflated, out5, 72);
    return sacch_check_crc(out5);
}

/* facch1.cpp:39-61, :63-75 */
static void facch1_deinterleave(const uint8_t* in, uint8_t* out) {
    memset(out, 0, 72);
    for (int i = 0; i < 16; i++) for (int k = 0; k < 9; k++) {
        const int inpos = i * 9 + k, outpos = k * 16 + i;
        out[outpos / 2] |= (uint8_t) (((in[inpos / 2] >> (1 - inpos % 2)) & 1u) << (1 - outpos % 2));
    }
}
static void facch1_inflate(const uint8_t* input, uint8_t* output) {
    memset(output, 0, 24);
    int pos = 0;
    for (int i = 0; i < 192; i++) {
        unsigned x = 0;
        if ((i - 1) % 4 != 0) { x = (input[pos / 2] >> (1 - pos % 2)) & 1u; pos++; }
        output[i / 8] |= (uint8_t) (x << (7 - i % 8));
    }
}
static int facch1_check_crc(const uint8_t* in) {
    uint16_t crc = 0xFFF;
    for (int i = 0; i < 80; i++) {
        const unsigned cb = ((crc >> 11) & 1u) ^ ((in[i / 8] >> (7 - i % 8)) & 1u);
        if (cb) crc ^= 0x407;
        crc = (uint16_t) (((crc << 1) & 0xFFE) | cb);
    }
    const uint16_t to_check = (uint16_t) (((uint16_t) in[10] << 4) | (in[11] >> 4));
    return to_check == crc;
}
/* Facch1::parse (facch1.cpp:8-27): 72 dibits -> 12 bytes; returns 1 when the CRC holds */
int orc_nxdn_facch1_parse(const uint8_t* dibits72, uint8_t* out12) {
    uint8_t deinterleaved[72], inflated[24];
    facch1_deinterleave(dibits72, deinterleaved);
    facch1_inflate(deinterleaved, inflated);
    memset(out12, 0, 12);
    orc_nxdn_trellis_decode(inflated, out12, 192);
    return facch1_check_crc(out12);
}
