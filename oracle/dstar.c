/*
 * dstar.c -- oracle restatement of the D-Star frame elements (TEST INFRASTRUCTURE ONLY): scrambler, CRC, and the
 * radio header (de-interleave, rate-1/2 K=3 Viterbi, CRC).
 *
 * PINNED: orc_dstar_scramble and orc_dstar_crc against the reference's own src/dstar_decoder/{scrambler,crc}.cpp
 * compiled in place (oracle/_ref/libdigiham_ref_dstar.so, tests/golden/dstar_ref.npz).
 * PARITY UNPINNED: the header decoder (header.cpp pulls in charset.hpp -> ICU, an external library: not built here)
 * and the phase state machine (dstar_phase.cpp, csdr), restated in decoders.c.
 */
#include "dh_oracle.h"
#include <string.h>

/* scrambler.cpp:7-21: 7-bit LFSR, whitening bit = s0 ^ s3, fed back into bit 6; bits are 0/1 */
void orc_dstar_scramble(uint8_t* shift_register, const uint8_t* input, uint8_t* output, size_t len) {
    memset(output, 0, len);
    uint8_t sr = *shift_register;
    for (size_t i = 0; i < len; i++) {
        const unsigned wb = (sr & 1u) ^ ((sr >> 3) & 1u);
        output[i] = (uint8_t) ((input[i] & 1u) ^ wb);
        sr = (uint8_t) (((sr & 0x7Eu) >> 1) | (wb << 6));
    }
    *shift_register = sr;
}

/* crc.cpp:6-23: reflected CCITT (0x8408), init 0xFFFF, inverted; returns the checksum (the reference compares its
 * two bytes, little endian in memory, with the two bytes that follow the data) */
uint16_t orc_dstar_crc(const uint8_t* data, size_t len) {
    uint16_t checksum = 0xFFFF;
    for (size_t k = 0; k < len; k++) {
        for (int i = 0; i < 8; i++) {
            checksum ^= (uint16_t) ((data[k] >> i) & 1u);
            if (checksum & 1u) checksum = (uint16_t) ((checksum >> 1) ^ 0x8408u);
            else checksum >>= 1;
        }
    }
    return (uint16_t) (checksum ^ 0xFFFFu);
}

/* Header::parseFromFrameData (header.cpp:53-58): CRC over 39 bytes against bytes 39, 40 */
int orc_dstar_header_crc_ok(const uint8_t* decoded41) {
    const uint16_t c = orc_dstar_crc(decoded41, 39);
    return decoded41[39] == (uint8_t) (c & 0xFF) && decoded41[40] == (uint8_t) (c >> 8);
}

/* header.cpp:60-72 */
static void header_deinterleave(const uint8_t* in, uint8_t* out) {
    memset(out, 0, 660);
    for (int i = 0; i < 12; i++) for (int k = 0; k < 28; k++) out[k * 24 + i] = in[i * 28 + k];
    for (int i = 12; i < 24; i++) for (int k = 0; k < 27; k++) out[k * 24 + i] = in[12 + i * 27 + k];
}

/* header.cpp:79-84 */
static const uint8_t dstar_transitions[4][2] = { {0, 3}, {3, 0}, {2, 1}, {1, 2} };

/* header.cpp:86-150: 4-state Viterbi over 330 steps, uint16 metrics from 0, k = 0 wins ties, lowest best end state;
 * output bit `pos` goes to bit (pos % 8) of byte pos / 8 (bit order reversed on the fly); 42 bytes */
static unsigned header_viterbi(const uint8_t* input, uint8_t* output) {
    uint16_t metric[4] = { 0, 0, 0, 0 }, next_metric[4];
    uint8_t data[4][42], next_data[4][42];
    memset(data, 0, sizeof(data));
    for (int pos = 0; pos < 330; pos++) {
        const uint8_t in_transition = (uint8_t) (((input[pos * 2] & 1u) << 1) | (input[pos * 2 + 1] & 1u));
        const int outpos = pos / 8, outshift = pos % 8;
        for (int i = 0; i < 4; i++) {
            uint16_t best_metric = 0xFFFF; int selected = -1;
            const uint8_t outbit = (uint8_t) ((i & 2) >> 1);
            for (int k = 0; k < 2; k++) {
                const int previous_state = ((i << 1) & 2) | k;
                const uint8_t transition = dstar_transitions[previous_state][outbit];
                const uint16_t m = (uint16_t) (metric[previous_state] + orc_hamming_distance(&in_transition, &transition, 1));
                if (k == 0 || m < best_metric) { best_metric = m; selected = previous_state; }
            }
            next_metric[i] = best_metric;
            memcpy(next_data[i], data[selected], 42);
            next_data[i][outpos] |= (uint8_t) (outbit << outshift);
        }
        memcpy(metric, next_metric, sizeof(metric));
        memcpy(data, next_data, sizeof(data));
    }
    int best = 0;
    for (int i = 1; i < 4; i++) if (metric[i] < metric[best]) best = i;
    memcpy(output, data[best], 42);
    return metric[best];
}

/* Header::parseFromHeader (header.cpp:24-51): 660 received bits -> 41 header bytes; returns 1 when the path metric is
 * <= 10 and the CRC holds */
int orc_dstar_header_parse(const uint8_t* raw660, uint8_t* out41) {
    uint8_t descrambled[660], deinterleaved[660], decoded[42];
    uint8_t sr = 0x7F;                                      /* a fresh Scrambler (scrambler.hpp:13) */
    orc_dstar_scramble(&sr, raw660, descrambled, 660);
    header_deinterleave(descrambled, deinterleaved);
    const unsigned errors = header_viterbi(deinterleaved, decoded);
    if (errors > 10) return 0;
    if (!orc_dstar_header_crc_ok(decoded)) return 0;
    memcpy(out41, decoded, 41);
    return 1;
}
