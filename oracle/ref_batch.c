/*
 * ref_batch.c -- batch entry points over the REFERENCE's own FEC functions
 * (TEST INFRASTRUCTURE ONLY).  Linked into oracle/_ref/libdigiham_ref_fec.so
 * together with the reference's unmodified C sources compiled in place; the
 * prototypes below restate the reference headers
 * (src/dmr_decoder/{bptc_196_96,hamming_*,golay_20_8,quadratic_residue}.h,
 * src/ysf_decoder/{golay_24_12,trellis,crc16,whitening}.h, src/lib/hamming_distance.h).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>
bool hamming_7_4(uint8_t* data);
bool hamming_13_9(uint16_t* data);
bool hamming_15_11(uint16_t* data);
bool hamming_16_11(uint16_t* data);
bool quadratic_residue(uint16_t* data);
bool golay_20_8(uint32_t* data);
bool golay_24_12(uint32_t* data);
bool bch_31_21(uint32_t* data);                     /* src/pocsag_decoder/bch_31_21.h */
bool bptc_196_96(uint8_t* payload, uint8_t* output);
uint8_t decode_trellis(uint8_t* input, uint8_t size, uint8_t* output);
uint16_t crc16_checksum(uint8_t* data, int count);
void decode_whitening(uint8_t* input, uint8_t* output, uint8_t num);
unsigned int hamming_distance(uint8_t* a, uint8_t* b, size_t size);
#define BATCH(name) ref_batch_##name
#define FN_HAMMING_7_4 hamming_7_4
#define FN_HAMMING_13_9 hamming_13_9
#define FN_HAMMING_15_11 hamming_15_11
#define FN_HAMMING_16_11 hamming_16_11
#define FN_QR quadratic_residue
#define FN_GOLAY_20_8 golay_20_8
#define FN_GOLAY_24_12 golay_24_12
#define FN_BCH_31_21 bch_31_21
#define FN_BPTC bptc_196_96
#define FN_TRELLIS decode_trellis
#define FN_CRC16 crc16_checksum
#define FN_WHITENING decode_whitening
#define FN_HAMMING_DISTANCE hamming_distance
#include "batch_impl.h"
