/*
 * batch_impl.h -- batch loops over the scalar FEC primitives (TEST INFRASTRUCTURE ONLY).
 * Included twice: by batch.c with the oracle's orc_* functions and by
 * ref_batch.c with the reference's own symbols, so tests can push millions of
 * words through either implementation with one ctypes call.
 * Before inclusion define BATCH(name) and FN_* macros.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>
#include <string.h>

void BATCH(hamming_7_4)(uint8_t* d, uint8_t* ok, size_t n) { for (size_t i = 0; i < n; i++) ok[i] = FN_HAMMING_7_4(&d[i]); }
void BATCH(hamming_13_9)(uint16_t* d, uint8_t* ok, size_t n) { for (size_t i = 0; i < n; i++) ok[i] = FN_HAMMING_13_9(&d[i]); }
void BATCH(hamming_15_11)(uint16_t* d, uint8_t* ok, size_t n) { for (size_t i = 0; i < n; i++) ok[i] = FN_HAMMING_15_11(&d[i]); }
void BATCH(hamming_16_11)(uint16_t* d, uint8_t* ok, size_t n) { for (size_t i = 0; i < n; i++) ok[i] = FN_HAMMING_16_11(&d[i]); }
void BATCH(quadratic_residue)(uint16_t* d, uint8_t* ok, size_t n) { for (size_t i = 0; i < n; i++) ok[i] = FN_QR(&d[i]); }
void BATCH(golay_20_8)(uint32_t* d, uint8_t* ok, size_t n) { for (size_t i = 0; i < n; i++) ok[i] = FN_GOLAY_20_8(&d[i]); }
void BATCH(golay_24_12)(uint32_t* d, uint8_t* ok, size_t n) { for (size_t i = 0; i < n; i++) ok[i] = FN_GOLAY_24_12(&d[i]); }
void BATCH(bch_31_21)(uint32_t* d, uint8_t* ok, size_t n) { for (size_t i = 0; i < n; i++) ok[i] = FN_BCH_31_21(&d[i]); }

/* in [n][25] -> out [n][12] (left untouched = caller-zeroed on failure), ok[n] */
void BATCH(bptc_196_96)(const uint8_t* in, uint8_t* out, uint8_t* ok, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint8_t tmp[25];
        memcpy(tmp, in + i * 25, 25);
        ok[i] = FN_BPTC(tmp, out + i * 12);
    }
}

/* in [n][in_stride] packed dibits -> out [n][out_stride], metric[n] */
void BATCH(trellis)(const uint8_t* in, size_t in_stride, uint8_t size, uint8_t* out, size_t out_stride, uint8_t* metric, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint8_t tmp[64], o[32];
        memcpy(tmp, in + i * in_stride, (size_t) (size + 3) / 4);
        memset(o, 0, sizeof(o));
        metric[i] = FN_TRELLIS(tmp, size, o);
        memcpy(out + i * out_stride, o, (size_t) (size + 7) / 8);
    }
}

void BATCH(crc16)(const uint8_t* in, size_t stride, int count, uint16_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint8_t tmp[256];
        memcpy(tmp, in + i * stride, (size_t) count);
        out[i] = FN_CRC16(tmp, count);
    }
}

void BATCH(whitening)(const uint8_t* in, uint8_t* out, size_t stride, uint8_t num, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint8_t tmp[32], o[32];
        memcpy(tmp, in + i * stride, (size_t) (num + 7) / 8);
        FN_WHITENING(tmp, o, num);
        memcpy(out + i * stride, o, (size_t) (num + 7) / 8);
    }
}

void BATCH(hamming_distance)(const uint8_t* a, const uint8_t* b, size_t size, uint32_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint8_t ta[64], tb[64];
        memcpy(ta, a + i * size, size); memcpy(tb, b + i * size, size);
        out[i] = FN_HAMMING_DISTANCE(ta, tb, size);
    }
}
