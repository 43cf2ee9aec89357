/*
 * fec.c -- oracle restatement of the reference's FEC primitives (TEST INFRASTRUCTURE ONLY).
 *
 * Follows the reference's algorithms step by step: bit-serial syndrome from
 * the parity-check rows, then a first-match linear scan of a
 * {syndrome, error_pattern} list.  The reference ships those lists as
 * generated literals; here they are regenerated at start-up from the same
 * definition ("syndromes of every error pattern of weight <= t, enumerated in
 * ascending numeric order"), using parity-check rows derived from the ETSI
 * TS 102 361-1 Annex B / YSF spec generator matrices G = [I | P].
 *
 * PINNED: every decoder here is compared exhaustively against the reference's
 * own compiled C (oracle/_ref/libdigiham_ref_fec.so) in tests/test_oracle_vs_ref.py.
 */
#include "dh_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- helpers */

/* reference: src/lib/hamming_distance.c:3-10, LUT src/lib/hamming_distance.h:4-10 */
unsigned int orc_hamming_distance(const uint8_t* a, const uint8_t* b, size_t size) {
    unsigned int distance = 0;
    for (size_t i = 0; i < size; i++) {
        uint8_t x = a[i] ^ b[i];
        unsigned int c = 0;
        while (x) { c += x & 1; x >>= 1; }   /* == lookuptable[x] */
        distance += c;
    }
    return distance;
}

typedef struct { uint32_t syndrome; uint32_t pattern; } correction;

typedef struct {
    int n, k, t;              /* code length, dimension, corrected weight */
    const uint16_t* p;        /* P part of G = [I_k | P], one (n-k)-bit row per data bit */
    uint32_t h[12];           /* parity-check rows, bit l <-> data bit l (n-1 = first column) */
    correction* table;
    int table_len;
    int ready;
} blockcode;

static int popcount32(uint32_t x) { int c = 0; while (x) { c += x & 1; x >>= 1; } return c; }

/* the bit-serial syndrome of e.g. golay_20_8.c:1403-1419 / hamming_13_9.c:52-68 */
static uint32_t bc_parity(const blockcode* c, uint32_t data) {
    uint32_t parity = 0;
    for (int k = 0; k < c->n - c->k; k++) {
        uint8_t bit = 0;
        for (int l = 0; l < c->n; l++) {
            if ((c->h[k] >> l) & 1) bit ^= (data >> l) & 1;
        }
        parity = (parity << 1) | (bit & 1);
    }
    return parity;
}

static void bc_init(blockcode* c) {
    if (c->ready) return;
    int r = c->n - c->k;
    /* H = [P^T | I_r] (the comment blocks in hamming_13_9.c:15-20, golay_20_8.c:14-27 ...) */
    for (int row = 0; row < r; row++) {
        uint32_t m = 0;
        for (int j = 0; j < c->k; j++) {
            if ((c->p[j] >> (r - 1 - row)) & 1) m |= 1u << (c->n - 1 - j);
        }
        m |= 1u << (r - 1 - row);
        c->h[row] = m;
    }
    /* correction list: all patterns of weight 1..t, ascending (golay_20_8.c:50-1400 etc.) */
    int cap = 0;
    for (uint32_t e = 1; e < (1u << c->n); e++) if (popcount32(e) <= c->t) cap++;
    c->table = (correction*) malloc(sizeof(correction) * (size_t) cap);
    c->table_len = 0;
    for (uint32_t e = 1; e < (1u << c->n); e++) {
        if (popcount32(e) > c->t) continue;
        c->table[c->table_len].syndrome = bc_parity(c, e);
        c->table[c->table_len].pattern = e;
        c->table_len++;
    }
    c->ready = 1;
}

/* e.g. golay_20_8.c:1421-1435: zero syndrome -> ok; first match -> xor; else fail */
static bool bc_decode(blockcode* c, uint32_t* data) {
    bc_init(c);
    uint32_t parity = bc_parity(c, *data);
    if (parity == 0) return true;
    for (int i = 0; i < c->table_len; i++) {
        if (c->table[i].syndrome == parity) {
            *data ^= c->table[i].pattern;
            return true;
        }
    }
    return false;
}

static uint32_t bc_encode(const blockcode* c, uint32_t info) {
    int r = c->n - c->k;
    uint32_t par = 0;
    for (int j = 0; j < c->k; j++) {
        if ((info >> (c->k - 1 - j)) & 1) par ^= c->p[j];
    }
    return (info << r) | par;
}

/* ------------------------------------------------ generator matrices (P) */
/* ETSI TS 102 361-1 B.3.5 (hamming_7_4.c:4-9) */
static const uint16_t P_H74[4] = { 0x5, 0x7, 0x6, 0x3 };
/* ETSI B.3.4 (hamming_13_9.c:4-13) */
static const uint16_t P_H139[9] = { 0xF, 0xE, 0x7, 0xA, 0x5, 0xB, 0xC, 0x6, 0x3 };
/* ETSI B.3.4 (hamming_15_11.c:4-15) */
static const uint16_t P_H1511[11] = { 0x9, 0xD, 0xF, 0xE, 0x7, 0xA, 0x5, 0xB, 0xC, 0x6, 0x3 };
/* ETSI B.3.4 (hamming_16_11.c:5-16) */
static const uint16_t P_H1611[11] = { 0x13, 0x1A, 0x1F, 0x1C, 0x0E, 0x15, 0x0B, 0x16, 0x19, 0x0D, 0x07 };
/* YSF spec appendix A (golay_24_12.c:5-16); rows 4..11 are also the Golay(20,8) rows of ETSI B.3.1 (golay_20_8.c:4-11) */
static const uint16_t P_G2412[12] = { 0xC75, 0x63B, 0xF68, 0x7B4, 0x3DA, 0xD99, 0x6CD, 0x367, 0xDC6, 0xA97, 0x93E, 0x8EB };
/* ETSI B.3.2 (quadratic_residue.c:4-10) */
static const uint16_t P_QR[7] = { 0x04F, 0x11E, 0x1B7, 0x1E2, 0x1C9, 0x0E5, 0x073 };

static blockcode C_H74   = { 7, 4, 1, P_H74 };
static blockcode C_H139  = { 13, 9, 1, P_H139 };
static blockcode C_H1511 = { 15, 11, 1, P_H1511 };
static blockcode C_H1611 = { 16, 11, 1, P_H1611 };
static blockcode C_G208  = { 20, 8, 3, P_G2412 + 4 };
static blockcode C_G2412 = { 24, 12, 3, P_G2412 };
static blockcode C_QR    = { 16, 7, 2, P_QR };

/* The correction lists are built once when the library is loaded, so the decoders are safe to call
 * from the worker threads of pipe.c (bc_init itself is not re-entrant). */
__attribute__((constructor)) static void orc_fec_init(void) {
    bc_init(&C_H74); bc_init(&C_H139); bc_init(&C_H1511); bc_init(&C_H1611);
    bc_init(&C_G208); bc_init(&C_G2412); bc_init(&C_QR);
}

/* hamming_7_4.c:56-72 */
bool orc_hamming_7_4(uint8_t* data) { uint32_t d = *data; bool r = bc_decode(&C_H74, &d); *data = (uint8_t) d; return r; }
/* hamming_13_9.c:70-84 */
bool orc_hamming_13_9(uint16_t* data) { uint32_t d = *data; bool r = bc_decode(&C_H139, &d); *data = (uint16_t) d; return r; }
/* hamming_15_11.c:74-88 */
bool orc_hamming_15_11(uint16_t* data) { uint32_t d = *data; bool r = bc_decode(&C_H1511, &d); *data = (uint16_t) d; return r; }
/* hamming_16_11.c:79-93 */
bool orc_hamming_16_11(uint16_t* data) { uint32_t d = *data; bool r = bc_decode(&C_H1611, &d); *data = (uint16_t) d; return r; }
/* golay_20_8.c:1421-1435 */
bool orc_golay_20_8(uint32_t* data) { return bc_decode(&C_G208, data); }
/* golay_24_12.c:2401-2415 */
bool orc_golay_24_12(uint32_t* data) { return bc_decode(&C_G2412, data); }
/* quadratic_residue.c:321-335 */
bool orc_quadratic_residue(uint16_t* data) { uint32_t d = *data; bool r = bc_decode(&C_QR, &d); *data = (uint16_t) d; return r; }

uint8_t  orc_hamming_7_4_encode(uint8_t d)     { return (uint8_t)  bc_encode(&C_H74, d & 0xF); }
uint16_t orc_hamming_13_9_encode(uint16_t d)   { return (uint16_t) bc_encode(&C_H139, d & 0x1FF); }
uint16_t orc_hamming_15_11_encode(uint16_t d)  { return (uint16_t) bc_encode(&C_H1511, d & 0x7FF); }
uint16_t orc_hamming_16_11_encode(uint16_t d)  { return (uint16_t) bc_encode(&C_H1611, d & 0x7FF); }
uint32_t orc_golay_20_8_encode(uint8_t d)      { return bc_encode(&C_G208, d); }
uint32_t orc_golay_24_12_encode(uint16_t d)    { return bc_encode(&C_G2412, d & 0xFFF); }
uint16_t orc_quadratic_residue_encode(uint8_t d) { return (uint16_t) bc_encode(&C_QR, d & 0x7F); }

/* ------------------------------------------------------------ BPTC(196,96) */
/* reference: src/dmr_decoder/bptc_196_96.c:5-59 */
bool orc_bptc_196_96(const uint8_t* payload, uint8_t* output) {
    /* :11-15 deinterleave, index i <- (i * 181) mod 196 */
    uint8_t deint[25];
    memset(deint, 0, sizeof(deint));
    for (unsigned i = 0; i < 196; i++) {
        unsigned src = (i * 181u) % 196u;
        deint[i / 8] |= ((payload[src / 8] >> (7 - (src % 8))) & 1) << (7 - (i % 8));
    }
    /* :18-28 pivot into 15 columns of 13 bits (skipping R(3): +1), column Hamming(13,9);
     * the result is and-ed, all 15 columns are always decoded (H6) */
    uint16_t cols[15];
    bool ok = true;
    for (unsigned i = 0; i < 15; i++) {
        cols[i] = 0;
        for (unsigned k = 0; k < 13; k++) {
            unsigned src = k * 15 + i + 1;
            cols[i] |= ((deint[src / 8] >> (7 - (src % 8))) & 1) << (12 - k);
        }
        ok &= orc_hamming_13_9(&cols[i]);
    }
    if (!ok) return false;
    /* :33-40 pivot back, first 9 rows, row Hamming(15,11) */
    uint16_t rows[9];
    for (unsigned i = 0; i < 9; i++) {
        rows[i] = 0;
        for (unsigned k = 0; k < 15; k++) {
            rows[i] |= ((cols[k] >> (12 - i)) & 1) << (14 - k);
        }
        ok &= orc_hamming_15_11(&rows[i]);
    }
    if (!ok) return false;
    /* :45-56 extract 96 info bits: row 0 carries 3 reserved + 8 info, rows 1..8 carry 11 info bits each
     * (expressed as a bit stream instead of the reference's twelve mask expressions; same bits) */
    uint8_t bits[96];
    int nb = 0;
    for (int r = 0; r < 9; r++) {
        for (int c = (r == 0 ? 3 : 0); c < 11; c++) {
            bits[nb++] = (rows[r] >> (14 - c)) & 1;
        }
    }
    for (int i = 0; i < 12; i++) {
        uint8_t v = 0;
        for (int b = 0; b < 8; b++) v = (uint8_t) ((v << 1) | bits[i * 8 + b]);
        output[i] = v;
    }
    return true;
}

void orc_bptc_196_96_encode(const uint8_t* info, uint8_t* payload) {
    uint16_t rows[13];
    int nb = 0;
    memset(rows, 0, sizeof(rows));
    for (int r = 0; r < 9; r++) {
        uint16_t d = 0;
        for (int c = (r == 0 ? 3 : 0); c < 11; c++) {
            int bit = (info[nb / 8] >> (7 - nb % 8)) & 1;
            nb++;
            d |= (uint16_t) (bit << (10 - c));
        }
        rows[r] = orc_hamming_15_11_encode(d);
    }
    for (int c = 0; c < 15; c++) {
        uint16_t d = 0;
        for (int r = 0; r < 9; r++) d |= (uint16_t) (((rows[r] >> (14 - c)) & 1) << (8 - r));
        uint16_t cw = orc_hamming_13_9_encode(d);
        for (int r = 9; r < 13; r++) rows[r] |= (uint16_t) (((cw >> (12 - r)) & 1) << (14 - c));
    }
    uint8_t deint[196];
    deint[0] = 0; /* R(3) */
    for (int r = 0; r < 13; r++)
        for (int c = 0; c < 15; c++) deint[1 + r * 15 + c] = (rows[r] >> (14 - c)) & 1;
    memset(payload, 0, 25);
    for (unsigned i = 0; i < 196; i++) {
        unsigned dst = (i * 181u) % 196u;
        payload[dst / 8] |= (uint8_t) (deint[i] << (7 - dst % 8));
    }
}

/* ---------------------------------------------------- rate-1/2 K=5 Viterbi */
/* transition outputs of the YSF convolutional code, G1 = 1+D^3+D^4, G2 = 1+D+D^2+D^4,
 * state = last four input bits, newest in bit 3 (equals the table at trellis.c:8-25) */
static uint8_t trellis_out(uint8_t state, uint8_t bit) {
    uint8_t s0 = state & 1, s1 = (state >> 1) & 1, s2 = (state >> 2) & 1, s3 = (state >> 3) & 1;
    uint8_t hi = bit ^ s1 ^ s0;
    uint8_t lo = bit ^ s3 ^ s2 ^ s0;
    return (uint8_t) ((hi << 1) | lo);
}

/* reference: src/ysf_decoder/trellis.c:32-109.  Register-exchange Viterbi with
 * uint8 metrics (wrap on overflow), all 16 states start at metric 0, ties keep
 * k = 0, final pick = lowest index among the minimum metric. */
uint8_t orc_decode_trellis(const uint8_t* input, uint8_t size, uint8_t* output) {
    uint8_t data_size = (uint8_t) ((size + 7) / 8);
    uint8_t metric[16], nmetric[16];
    uint8_t data[16][32], ndata[16][32];
    memset(metric, 0, sizeof(metric));
    memset(data, 0, sizeof(data));
    for (uint8_t pos = 0; pos < size; pos++) {
        uint8_t in_transition = (input[pos / 4] >> (2 * (3 - pos % 4))) & 3;
        uint8_t outpos = pos / 8, outshift = (uint8_t) (7 - pos % 8);
        for (uint8_t i = 0; i < 16; i++) {
            uint8_t best_metric = 0xFF, selected = 0xFF;
            uint8_t outbit = (i & 8) >> 3;
            for (uint8_t k = 0; k < 2; k++) {
                uint8_t previous_state = (uint8_t) (((i << 1) & 0xE) | k);
                uint8_t transition = trellis_out(previous_state, outbit);
                uint8_t m = (uint8_t) (metric[previous_state] + orc_hamming_distance(&in_transition, &transition, 1));
                if (k == 0 || m < best_metric) { best_metric = m; selected = previous_state; }
            }
            nmetric[i] = best_metric;
            memcpy(ndata[i], data[selected], data_size);
            ndata[i][outpos] |= (uint8_t) (outbit << outshift);
        }
        memcpy(metric, nmetric, sizeof(metric));
        memcpy(data, ndata, sizeof(data));
    }
    int best = 0;
    for (int i = 1; i < 16; i++) if (metric[i] < metric[best]) best = i;
    memcpy(output, data[best], data_size);
    return metric[best];
}

void orc_trellis_encode(const uint8_t* bits, int nbits, uint8_t* out) {
    uint8_t state = 0;
    memset(out, 0, (size_t) (nbits + 3) / 4);
    for (int i = 0; i < nbits; i++) {
        uint8_t b = (bits[i / 8] >> (7 - i % 8)) & 1;
        uint8_t o = trellis_out(state, b);
        state = (uint8_t) ((b << 3) | (state >> 1));
        out[i / 4] |= (uint8_t) (o << (6 - 2 * (i % 4)));
    }
}

/* ------------------------------------------------------------- CRC16 / PN9 */
/* reference: src/ysf_decoder/crc16.c:3-18 (CCITT 0x1021, init 0, final inversion) */
uint16_t orc_crc16_checksum(const uint8_t* data, int count) {
    uint16_t checksum = 0;
    for (int k = 0; k < count; k++) {
        for (int i = 0; i < 8; i++) {
            uint16_t input = (data[k] >> (7 - i)) & 1;
            uint16_t next_input = input ^ ((checksum >> 15) & 1);
            checksum = (uint16_t) (checksum << 1);
            checksum ^= (uint16_t) ((next_input << 12) | (next_input << 5) | next_input);
        }
    }
    return checksum ^ 0xFFFF;
}

/* reference: src/ysf_decoder/whitening.c:6-22 */
void orc_decode_whitening(const uint8_t* input, uint8_t* output, uint8_t num) {
    uint16_t wsr = 0x1C9;
    memset(output, 0, (size_t) (num + 7) / 8);
    for (int i = 0; i < num; i++) {
        int pos = i / 8, shift = 7 - i % 8;
        uint16_t wb = wsr & 1;
        output[pos] |= (uint8_t) ((input[pos] & (1 << shift)) ^ (wb << shift));
        wb = ((wsr >> 4) & 1) ^ wb;
        wsr = (uint16_t) (((wsr & 0x1FE) >> 1) | (wb << 8));
    }
}
