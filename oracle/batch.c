/* batch.c -- batch entry points over the oracle's own FEC restatement (TEST INFRASTRUCTURE ONLY). */
#include "dh_oracle.h"
#define BATCH(name) orc_batch_##name
#define FN_HAMMING_7_4 orc_hamming_7_4
#define FN_HAMMING_13_9 orc_hamming_13_9
#define FN_HAMMING_15_11 orc_hamming_15_11
#define FN_HAMMING_16_11 orc_hamming_16_11
#define FN_QR orc_quadratic_residue
#define FN_GOLAY_20_8 orc_golay_20_8
#define FN_GOLAY_24_12 orc_golay_24_12
#define FN_BCH_31_21 orc_bch_31_21
#define FN_BPTC orc_bptc_196_96
#define FN_TRELLIS orc_decode_trellis
#define FN_CRC16 orc_crc16_checksum
#define FN_WHITENING orc_decode_whitening
#define FN_HAMMING_DISTANCE orc_hamming_distance
#include "batch_impl.h"
