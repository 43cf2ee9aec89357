/*
 * pocsag.c -- oracle restatement of the POCSAG paging decoder's pieces (TEST INFRASTRUCTURE ONLY):
 * BCH(31,21) (src/pocsag_decoder/bch_31_21.c), Codeword (codeword.cpp), Message (message.cpp).
 *
 * PINNED: orc_bch_31_21 is checked against the reference's own bch_31_21.c compiled in place
 * (oracle/_ref/libdigiham_ref_fec.so, tests/golden/pocsag_ref.npz).  The codeword / message classes and the
 * phase state machine (pocsag_phase.cpp) include csdr headers and are PARITY UNPINNED; the state machine is in
 * decoders.c.
 */
#include "dh_oracle.h"
#include <stdlib.h>
#include <string.h>

/* BCH(31,21), generator x^10+x^9+x^8+x^6+x^5+x^3+1 (CCIR 584 / ITU-R M.584 section 2): the reference lists the
 * parity-check rows (bch_31_21.c:3-14) and a {syndrome, pattern} table for all single and double errors
 * (:21-518); the same rows follow from the generator (row k, bit j = coefficient 9-k of x^j mod g) and, since
 * the code has minimum distance 5, every pattern of weight <= 2 has its own syndrome: a dense table is equivalent
 * to the reference's first-match scan. */
static uint32_t bch_h[10];
static uint32_t bch_lut[1024];
static int bch_ready = 0;

static uint16_t bch_syndrome(uint32_t data) {               /* bch_31_21.c:520-543: row 0 ends up in the MSB */
    uint16_t parity = 0;
    for (int k = 0; k < 10; k++) parity = (uint16_t) ((parity << 1) | (__builtin_popcount(data & bch_h[k]) & 1));
    return parity;
}

__attribute__((constructor)) static void bch_init(void) {
    if (bch_ready) return;
    uint32_t rem[31];
    uint32_t r = 1;                                          /* x^0 mod g */
    for (int j = 0; j < 31; j++) {
        rem[j] = r;
        r <<= 1;
        if (r & 0x400u) r ^= 0x769u;                         /* g = 111 0110 1001 */
    }
    for (int k = 0; k < 10; k++) {
        bch_h[k] = 0;
        for (int j = 0; j < 31; j++) if ((rem[j] >> (9 - k)) & 1u) bch_h[k] |= 1u << j;
    }
    memset(bch_lut, 0, sizeof(bch_lut));
    for (int a = 0; a < 31; a++) {
        bch_lut[bch_syndrome(1u << a)] = 1u << a;
        for (int b = a + 1; b < 31; b++) bch_lut[bch_syndrome((1u << a) | (1u << b))] = (1u << a) | (1u << b);
    }
    bch_ready = 1;
}

/* bch_31_21.c:545-561 */
bool orc_bch_31_21(uint32_t* data) {
    const uint16_t parity = bch_syndrome(*data);
    if (parity == 0) return true;
    const uint32_t p = bch_lut[parity];
    if (p == 0) return false;
    *data ^= p;
    return true;
}

uint32_t orc_bch_31_21_row(int k) { return bch_h[k]; }

/* systematic encoder for the tests: 21 data bits -> 31-bit word (data in bits 30..10) */
uint32_t orc_bch_31_21_encode(uint32_t data21) {
    uint32_t w = (data21 & 0x1FFFFFu) << 10, r = w;
    for (int j = 30; j >= 10; j--) if (r & (1u << j)) r ^= 0x769u << (j - 10);
    return w | (r & 0x3FFu);
}

/* Codeword::parse (codeword.cpp:9-32): 32 received bits (one per byte, `input[i] && 1`: any non-zero byte is a 1),
 * BCH over the upper 31, then even parity over all 32.  Returns 0 and leaves *out untouched on failure. */
int orc_pocsag_codeword_parse(const uint8_t* input, uint32_t* out) {
    uint32_t codeword = 0;
    for (int i = 0; i < 32; i++) codeword |= (uint32_t) (input[i] && 1) << (31 - i);
    uint32_t payload = codeword >> 1;
    if (!orc_bch_31_21(&payload)) return 0;
    codeword = (codeword & 1u) | (payload << 1);
    if (__builtin_popcount(codeword) & 1) return 0;
    *out = codeword;
    return 1;
}
