/*
 * elements.c -- oracle restatement of the burst / frame ELEMENT parsers of DMR and YSF (TEST INFRASTRUCTURE ONLY):
 * Cach + Tact, Emb, SlotType, EmbeddedCollector::getLc, Lc getters (src/dmr_decoder/{cach,tact,emb,slottype,
 * embedded,lc}.cpp) and Fich (src/ysf_decoder/fich.cpp).  The frame state machines in decoders.c call exactly these
 * functions.
 *
 * PINNED: those reference classes do not include csdr, so oracle/Makefile compiles them where they lie into
 * oracle/_ref/libdigiham_ref_{dmr,ysf,pocsag,dstar}.so (glue: ref_dmr.cpp, ref_ysf.cpp, ref_pocsag.cpp,
 * ref_dstar.cpp); tests/golden/make_golden_elements.py ran them (Cach on all 2^24 CACHs, Emb on all 2^16 words,
 * SlotType on all 2^20, encoded + noisy embedded LCs and FICHs, ...) and committed hashes / vectors
 * (tests/golden/elements_ref.{npz,json}).  tests/test_elements.py checks the functions below against those.
 *
 * The orc_el_* batch entry points at the end have the signatures of the ref_el_* functions of the glue files.
 */
#include "dh_oracle.h"
#include <string.h>

/* ------------------------------------------------------------------ DMR */

/* Cach::parse (cach.cpp:11-31) + Tact::parse (tact.cpp:9-12).  raw: 12 dibits.  Returns 1 when the TACT passes
 * Hamming(7,4) (then *tact = the corrected 7-bit word); payload3 = the 17 CACH payload bits, LSB first per byte. */
int orc_dmr_cach_parse(const uint8_t* raw, uint8_t* tact_out, uint8_t* payload3) {
    static const uint8_t tact_positions[7] = { 0, 4, 8, 12, 14, 18, 22 };
    static const uint8_t payload_positions[17] = { 1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 15, 16, 17, 19, 20, 21, 23 };
    uint8_t tact = 0;
    for (int i = 0; i < 7; i++) {
        uint8_t bit = tact_positions[i];
        int pos = bit / 2, shift = 1 - (bit % 2);
        tact = (uint8_t) ((tact << 1) | ((raw[pos] >> shift) & 1));
    }
    if (payload3 != NULL) {
        payload3[0] = payload3[1] = payload3[2] = 0;
        for (int i = 0; i < 17; i++) {
            uint8_t bit = payload_positions[i];
            int pos = bit / 2, shift = 1 - (bit % 2);
            payload3[i / 8] |= (uint8_t) (((raw[pos] >> shift) & 1) << (i % 8));
        }
    }
    if (!orc_hamming_7_4(&tact)) return 0;
    *tact_out = tact;
    return 1;
}

/* Emb::parse (emb.cpp:9-14): QR(16,7) in place; getters emb.cpp:18-24 */
int orc_dmr_emb_parse(uint16_t* data) { return orc_quadratic_residue(data) ? 1 : 0; }
uint8_t orc_dmr_emb_color_code(uint16_t data) { return (data >> 12) & 15; }
uint8_t orc_dmr_emb_lcss(uint16_t data) { return (data >> 9) & 3; }

/* SlotType::parse (slottype.cpp:9-12): Golay(20,8) in place; getters slottype.cpp:16-22 */
int orc_dmr_slottype_parse(uint32_t* data) { return orc_golay_20_8(data) ? 1 : 0; }
uint8_t orc_dmr_slottype_color_code(uint32_t data) { return (data >> 16) & 15; }
uint8_t orc_dmr_slottype_data_type(uint32_t data) { return (data >> 12) & 15; }

/* EmbeddedCollector::getLc (embedded.cpp:32-94) over the collector's 16-byte buffer and fragment count.
 * Note :33 `offset < 3`: three fragments are enough, the fourth quarter is then whatever the buffer held. */
int orc_dmr_embedded_get_lc(const uint8_t* data16, int offset, uint8_t* lc) {
    if (offset < 3) return 0;
    uint16_t m[8] = { 0 };
    for (int i = 0; i < 16; i++) {
        uint8_t byte = data16[i];
        for (int k = 0; k < 8; k++) m[k] = (uint16_t) ((m[k] << 1) | ((byte >> (7 - k)) & 1));
    }
    for (int i = 0; i < 7; i++) if (!orc_hamming_16_11(&m[i])) return 0;
    uint16_t parity = 0;
    for (int i = 0; i < 8; i++) parity ^= m[i];
    if (parity != 0) return 0;
    lc[0] = (uint8_t) ((m[0] & 0xFF00) >> 8);
    lc[1] = (uint8_t) ((m[0] & 0x00E0) | ((m[1] & 0xF800) >> 11));
    lc[2] = (uint8_t) (((m[1] & 0x07E0) >> 3) | ((m[2] & 0xC000) >> 14));
    lc[3] = (uint8_t) ((m[2] & 0x3FC0) >> 6);
    lc[4] = (uint8_t) ((m[3] & 0xFF00) >> 8);
    lc[5] = (uint8_t) ((m[3] & 0x00C0) | ((m[4] & 0xFC00) >> 10));
    lc[6] = (uint8_t) (((m[4] & 0x03C0) >> 2) | ((m[5] & 0xF000) >> 12));
    lc[7] = (uint8_t) (((m[5] & 0x0FC0) >> 4) | ((m[6] & 0xC000) >> 14));
    lc[8] = (uint8_t) ((m[6] & 0x3FC0) >> 6);
    uint16_t checksum = 0;
    for (int i = 0; i < 9; i++) checksum = (uint16_t) (checksum + lc[i]);
    uint8_t checksum_mod = (uint8_t) (checksum % 31);
    uint8_t received = 0;
    for (int i = 0; i < 5; i++) received |= (uint8_t) ((m[i + 2] & 0x0020) >> (i + 1));
    return checksum_mod == received;
}

/* Lc getters (lc.cpp:23-43) */
void orc_dmr_lc_fields(const uint8_t* lc9, uint32_t* fields4, uint8_t* data7) {
    fields4[0] = lc9[0] & 0x3F;
    fields4[1] = lc9[1];
    fields4[2] = (uint32_t) lc9[6] << 16 | (uint32_t) lc9[7] << 8 | lc9[8];
    fields4[3] = (uint32_t) lc9[3] << 16 | (uint32_t) lc9[4] << 8 | lc9[5];
    memcpy(data7, lc9 + 2, 7);
}

/* ------------------------------------------------------------------ YSF */

/* Fich::parse (fich.cpp:12-52): 5 x 20 de-interleave, Viterbi, 4 x Golay(24,12), CRC-16 over the big-endian word */
int orc_ysf_fich_parse(const uint8_t* data, uint32_t* fich) {
    uint8_t raw[25] = { 0 };
    for (int i = 0; i < 100; i++) {
        int offset = ((i * 20) % 100 + i * 20 / 100);
        raw[i / 4] |= (uint8_t) ((data[offset] & 3) << (6 - 2 * (i % 4)));
    }
    uint8_t tr[13];
    orc_decode_trellis(raw, 100, tr);
    uint32_t g[4];
    bool ok = true;
    for (int i = 0; i < 4; i++) {
        g[i] = (uint32_t) tr[i * 3] << 16 | (uint32_t) tr[i * 3 + 1] << 8 | tr[i * 3 + 2];
        ok &= orc_golay_24_12(&g[i]);
    }
    if (!ok) return 0;
    uint32_t fich_data = (g[0] & 0x00FFF000) << 8 | (g[1] & 0x00FFF000) >> 4 | (g[2] & 0x00FF0000) >> 16;
    uint16_t fich_checksum = (uint16_t) ((g[2] & 0x0000F000) | (g[3] & 0x00FFF000) >> 12);
    uint8_t be[4] = { (uint8_t) (fich_data >> 24), (uint8_t) (fich_data >> 16), (uint8_t) (fich_data >> 8), (uint8_t) fich_data };
    if (orc_crc16_checksum(be, 4) != fich_checksum) return 0;
    *fich = fich_data;
    return 1;
}

/* ------------------------------------------------------------------ batch entry points (= ref_el_* of the glue) */

void orc_el_dmr_cach(const uint8_t* raw, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; i++, raw += 12, out += 8) {
        uint8_t tact = 0;
        memset(out, 0, 8);
        if (orc_dmr_cach_parse(raw, &tact, out + 5)) {
            out[0] = 1; out[1] = tact;
            out[2] = (tact >> 6) & 1; out[3] = (tact >> 5) & 1; out[4] = (tact >> 3) & 3;      /* tact.cpp:16-26 */
        }
    }
}

void orc_el_dmr_emb(const uint16_t* in, size_t n, uint8_t* out, uint16_t* corrected) {
    for (size_t i = 0; i < n; i++, out += 4) {
        uint16_t w = in[i];
        memset(out, 0, 4);
        corrected[i] = in[i];
        if (orc_dmr_emb_parse(&w)) { out[0] = 1; out[1] = orc_dmr_emb_color_code(w); out[2] = orc_dmr_emb_lcss(w); corrected[i] = w; }
    }
}

void orc_el_dmr_slottype(const uint32_t* in, size_t n, uint8_t* out, uint32_t* corrected) {
    for (size_t i = 0; i < n; i++, out += 4) {
        uint32_t w = in[i];
        memset(out, 0, 4);
        corrected[i] = in[i];
        if (orc_dmr_slottype_parse(&w)) { out[0] = 1; out[1] = orc_dmr_slottype_color_code(w); out[2] = orc_dmr_slottype_data_type(w); corrected[i] = w; }
    }
}

void orc_el_dmr_embedded_lc(const uint8_t* prev, const uint8_t* frags, const uint8_t* nfrags, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; i++, prev += 16, frags += 20, out += 10) {
        uint8_t data[16];
        int offset = 0;
        memcpy(data, prev, 16);                                          /* four collects, then reset() */
        for (int k = 0; k < nfrags[i]; k++) {                            /* embedded.cpp:20-26 */
            if (offset > 3) continue;
            memcpy(data + offset * 4, frags + 4 * k, 4);
            offset++;
        }
        memset(out, 0, 10);
        uint8_t lc[9];
        if (orc_dmr_embedded_get_lc(data, offset, lc)) { out[0] = 1; memcpy(out + 1, lc, 9); }
    }
}

void orc_el_dmr_lc(const uint8_t* lc, size_t n, uint32_t* fields, uint8_t* data7) {
    for (size_t i = 0; i < n; i++, lc += 9, fields += 4, data7 += 7) orc_dmr_lc_fields(lc, fields, data7);
}

void orc_el_ysf_fich(const uint8_t* dibits, size_t n, uint8_t* out, uint32_t* data) {
    for (size_t i = 0; i < n; i++, dibits += 100, out += 4) {
        uint32_t f = 0;
        memset(out, 0, 4);
        data[i] = 0;
        if (orc_ysf_fich_parse(dibits, &f)) {
            out[0] = 1; out[1] = (f >> 30) & 3; out[2] = (f >> 8) & 3; out[3] = (f >> 19) & 7;     /* fich.cpp:56-66 */
            data[i] = f;
        }
    }
}

void orc_el_pocsag_codeword(const uint8_t* bits, size_t n, uint8_t* out, uint32_t* words) {
    for (size_t i = 0; i < n; i++, bits += 32, out += 4, words += 3) {
        uint32_t w = 0;
        memset(out, 0, 4);
        words[0] = words[1] = words[2] = 0;
        if (orc_pocsag_codeword_parse(bits, &w)) {                       /* getters: codeword.cpp:35-55 */
            out[0] = 1; out[1] = w == 0x7A89C197u; out[2] = (w >> 31) == 0; out[3] = (w >> 11) & 3;
            words[0] = w; words[1] = (w >> 11) & 0xFFFFF; words[2] = (w >> 13) & 0x3FFFF;
        }
    }
}

void orc_el_dstar_header(const uint8_t* raw, size_t n, uint8_t* ok, uint8_t* data, uint8_t* text) {
    (void) text;                                                         /* strings are the host side's job */
    for (size_t i = 0; i < n; i++, raw += 660, data += 41) {
        memset(data, 0, 41);
        ok[i] = (uint8_t) orc_dstar_header_parse(raw, data);
        if (!ok[i]) memset(data, 0, 41);
    }
}
