/*
 * dh_oracle.h -- CPU oracle for the digiham hot path (TEST INFRASTRUCTURE ONLY).
 *
 * A plain-C restatement of the reference algorithms (jketterl/digiham v0.7.0-dev)
 * for rrc_filter -> gfsk/fsk_demodulator -> dmr/ysf_decoder FEC and the
 * digitalvoice_filter.  Every function cites the reference file:line it follows.
 *
 * This code is the *checker*: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product (digiham_amd/) never
 * links, imports or falls back to anything in this directory.
 *
 * Pinning status (see oracle/README.md):
 *   - FEC primitives (fec.c): PINNED against the reference's own C sources,
 *     compiled unmodified into oracle/_ref/libdigiham_ref_fec.so by
 *     oracle/Makefile and compared exhaustively (tests/test_oracle_vs_ref.py).
 *   - Burst / frame element parsers (elements.c: CACH/TACT, EMB, slot type,
 *     embedded LC, LC getters, FICH; pocsag.c: Codeword; dstar.c: header): PINNED
 *     against the reference's own csdr-free classes compiled in place
 *     (oracle/_ref/libdigiham_ref_{dmr,ysf,pocsag,dstar}.so, tests/test_elements.py).
 *   - DSP (dsp.c) and frame state machines (decoders.c): PARITY UNPINNED --
 *     those reference TUs include <csdr/module.hpp> (csdr 0.18, third-party,
 *     absent from this image), so they are unbuildable here without writing
 *     stand-in headers, which the rules forbid.  They are restated line by
 *     line from the cited sources instead.
 */
#ifndef DH_ORACLE_H
#define DH_ORACLE_H

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ fec.c */
unsigned int orc_hamming_distance(const uint8_t* a, const uint8_t* b, size_t size);
bool orc_hamming_7_4(uint8_t* data);
bool orc_hamming_13_9(uint16_t* data);
bool orc_hamming_15_11(uint16_t* data);
bool orc_hamming_16_11(uint16_t* data);
bool orc_golay_20_8(uint32_t* data);
bool orc_golay_24_12(uint32_t* data);
bool orc_quadratic_residue(uint16_t* data);
bool orc_bptc_196_96(const uint8_t* payload /*25*/, uint8_t* output /*12*/);
uint8_t orc_decode_trellis(const uint8_t* input, uint8_t size, uint8_t* output);
uint16_t orc_crc16_checksum(const uint8_t* data, int count);
void orc_decode_whitening(const uint8_t* input, uint8_t* output, uint8_t num);

/* encoders (not in the reference; used by the synthetic signal generator and
 * by round-trip tests).  Systematic encoders from the same generator matrices. */
uint8_t  orc_hamming_7_4_encode(uint8_t data4);
uint16_t orc_hamming_13_9_encode(uint16_t data9);
uint16_t orc_hamming_15_11_encode(uint16_t data11);
uint16_t orc_hamming_16_11_encode(uint16_t data11);
uint32_t orc_golay_20_8_encode(uint8_t data8);
uint32_t orc_golay_24_12_encode(uint16_t data12);
uint16_t orc_quadratic_residue_encode(uint8_t data7);
void orc_bptc_196_96_encode(const uint8_t* info /*12*/, uint8_t* payload /*25*/);
/* rate-1/2 K=5 convolutional encoder: nbits input bits (MSB first) -> nbits dibits packed 4/byte */
void orc_trellis_encode(const uint8_t* bits, int nbits, uint8_t* out);

/* ------------------------------------------------------------------ dsp.c */
typedef struct orc_rrc orc_rrc;
orc_rrc* orc_rrc_new(int narrow);
orc_rrc* orc_rrc_new_custom(unsigned n_zeros, double gain, const float* coeffs);
void orc_rrc_free(orc_rrc*);
void orc_rrc_process(orc_rrc*, const float* in, float* out, size_t n);
const float* orc_rrc_taps(int narrow, unsigned* n_zeros, double* gain);

/* ------------------------------------------------------------- frontend.c */
/* the receiver front-end of examples/dmr-decoder.sh:13-17 (own specification, parity unpinned) */
float orc_fe_atan2_over_pi(int32_t im, int32_t re);
void orc_frontend_process(float* state4, const int16_t* in, size_t n, float* out, int mode, int dcblock);

typedef struct orc_demod orc_demod;
/* levels = 4 -> GfskDemodulator(sps); levels = 2 -> FskDemodulator(sps, invert) */
orc_demod* orc_demod_new(unsigned sps, int levels, int invert);
void orc_demod_free(orc_demod*);
/* streaming: consumes from in[0..n), returns number of samples consumed (the
 * caller keeps the unread tail, exactly like a csdr ringbuffer reader) and
 * appends symbols to out (capacity cap); *n_out receives the count. */
size_t orc_demod_process(orc_demod*, const float* in, size_t n, uint8_t* out, size_t cap, size_t* n_out);

typedef struct orc_dvfilter orc_dvfilter;
orc_dvfilter* orc_dvfilter_new(void);
void orc_dvfilter_free(orc_dvfilter*);
void orc_dvfilter_process(orc_dvfilter*, const int16_t* in, int16_t* out, size_t n);

/* --------------------------------------------------------- dmr.c / ysf.c */
/* Decoder event record: what the reference hands to its MetaCollector (and a
 * few FEC intermediates), as plain data so it can be compared bit-exactly.
 * Layout is shared with the product's C-ABI (include/digiham_amd.h: dh_event). */
typedef struct {
    uint32_t sym_index;   /* absolute symbol index of the frame start (mod 2^32) */
    uint8_t  type;        /* ORC_EV_* */
    uint8_t  a;           /* slot (DMR) / frame number (YSF) */
    uint8_t  b;           /* sub-type: sync type, data type, ... */
    uint8_t  len;         /* valid payload bytes */
    uint8_t  payload[24];
} orc_event;

enum {
    ORC_EV_DMR_SYNC        = 1,  /* a=slot b=syncType payload[0]=softReset      dmr_phase.cpp:111-114 */
    ORC_EV_DMR_SLOT_RESET  = 2,  /* a=slot                                       dmr_phase.cpp:81,178,197,285 */
    ORC_EV_DMR_META_RESET  = 3,  /* MetaCollector::reset()                       dmr_phase.cpp:184,202 */
    ORC_EV_DMR_LC          = 4,  /* a=slot b=0 voice header / 1 embedded; payload=9 LC bytes  dmr_phase.cpp:159,276 */
    ORC_EV_DMR_SOFT_RESET  = 5,  /* a=slot b=data_type (terminator / idle)       dmr_phase.cpp:278-281 */
    ORC_EV_DMR_BPTC        = 6,  /* a=slot b=data_type payload=12 bytes          dmr_phase.cpp:272 */
    ORC_EV_DMR_SLOTTYPE    = 7,  /* a=slot b=data_type payload[0]=color code     dmr_phase.cpp:247-249 */
    ORC_EV_DMR_EMB         = 8,  /* a=slot b=lcss payload[0]=color code          dmr_phase.cpp:134 */
    ORC_EV_YSF_FICH        = 16, /* payload=fich u32 big endian                   fich.cpp:12-52 */
    ORC_EV_YSF_MODE        = 17, /* b=data type (0 V1, 2 DN, 3 VW, 1 FR data)      ysf_phase.cpp:75,87,113,134 */
    ORC_EV_YSF_DCH         = 18, /* a=frame number payload=10 bytes                ysf_phase.cpp:258-269 */
    ORC_EV_YSF_HEADER_DCH  = 19, /* a=0 (CSD1) / 1 (CSD2) payload=20 bytes         ysf_phase.cpp:145,152 */
    ORC_EV_YSF_META_RESET  = 20, /* b=0 sync loss, 1 header, 2 terminator          ysf_phase.cpp:50,141,163 */
    ORC_EV_NXDN_LICH       = 32, /* payload[0]=7-bit LICH (parity ok)                nxdn_phase.cpp:66-71 */
    ORC_EV_NXDN_SACCH      = 33, /* a=structure index payload=5 bytes (CRC ok)       nxdn_phase.cpp:112-114 */
    ORC_EV_NXDN_SACCH_SF   = 34, /* payload=9 bytes: a complete SACCH superframe     nxdn_phase.cpp:115-121 */
    ORC_EV_NXDN_SYNC_VOICE = 35, /* MetaCollector::setSync("voice")                  nxdn_phase.cpp:141 */
    ORC_EV_NXDN_FACCH1     = 36, /* a=block (0/1) payload=12 bytes (CRC ok)          nxdn_phase.cpp:152-154 */
    ORC_EV_NXDN_META_RESET = 37, /* b=0 sync loss, 1 TX_RELEASE                      nxdn_phase.cpp:51,157 */
    ORC_EV_DSTAR_HEADER      = 64, /* a=part (0: bytes 0-23, 1: bytes 24-40) b=0 radio header / 1 slow-data header; valid
                                      voice headers only: setFromHeader            dstar_phase.cpp:50, 214 */
    ORC_EV_DSTAR_VOICE_START = 65, /* b=1 after a header, 0 after a voice sync: a new VoicePhase   dstar_phase.cpp:27,52 */
    ORC_EV_DSTAR_SYNC_VOICE  = 66, /* setSync("voice")                               dstar_phase.cpp:110 */
    ORC_EV_DSTAR_MESSAGE     = 67, /* payload=20 bytes                               dstar_phase.cpp:207 */
    ORC_EV_DSTAR_SIMPLE      = 68, /* payload=len simple-data bytes appended         dstar_phase.cpp:178 */
    ORC_EV_DSTAR_FRAME_SYNC  = 69, /* parseFrameData(): the consumer parses its simple-data lines  dstar_phase.cpp:113,220 */
    ORC_EV_DSTAR_META_RESET  = 70, /* b=0 terminator, 1 sync lost                    dstar_phase.cpp:97,105 */
    ORC_EV_POCSAG_CODEWORD = 48, /* a=position in the batch payload=corrected word, big endian  pocsag_phase.cpp:56-57 */
};

/* ------------------------------------------------------------- elements.c */
/* burst / frame element parsers (pinned against oracle/_ref/libdigiham_ref_{dmr,ysf}.so) */
int orc_dmr_cach_parse(const uint8_t* raw12, uint8_t* tact, uint8_t* payload3 /* or NULL */);
int orc_dmr_emb_parse(uint16_t* data);
uint8_t orc_dmr_emb_color_code(uint16_t data);
uint8_t orc_dmr_emb_lcss(uint16_t data);
int orc_dmr_slottype_parse(uint32_t* data);
uint8_t orc_dmr_slottype_color_code(uint32_t data);
uint8_t orc_dmr_slottype_data_type(uint32_t data);
int orc_dmr_embedded_get_lc(const uint8_t* data16, int offset, uint8_t* lc9);
void orc_dmr_lc_fields(const uint8_t* lc9, uint32_t* fields4 /* opcode, fid, source, target */, uint8_t* data7);
int orc_ysf_fich_parse(const uint8_t* dibits100, uint32_t* fich);

/* --------------------------------------------------------------- pocsag.c */
bool orc_bch_31_21(uint32_t* data);
uint32_t orc_bch_31_21_encode(uint32_t data21);
uint32_t orc_bch_31_21_row(int k);
int orc_pocsag_codeword_parse(const uint8_t* input32, uint32_t* out);

/* ---------------------------------------------------------------- dstar.c */
void orc_dstar_scramble(uint8_t* shift_register, const uint8_t* input, uint8_t* output, size_t len);
uint16_t orc_dstar_crc(const uint8_t* data, size_t len);
int orc_dstar_header_crc_ok(const uint8_t* decoded41);
int orc_dstar_header_parse(const uint8_t* raw660, uint8_t* out41);

/* ----------------------------------------------------------------- nxdn.c */
void orc_nxdn_scramble(uint16_t* shift_register, const uint8_t* input, uint8_t* output, size_t len);
int orc_nxdn_lich_parse(const uint8_t* raw8);                       /* -1 or the 7-bit LICH */
unsigned orc_nxdn_trellis_decode(const uint8_t* input, uint8_t* output, size_t len_bits);
int orc_nxdn_sacch_parse(const uint8_t* dibits30, uint8_t* out5);
int orc_nxdn_facch1_parse(const uint8_t* dibits72, uint8_t* out12);

typedef struct orc_decoder orc_decoder;
orc_decoder* orc_dmr_new(void);
orc_decoder* orc_ysf_new(void);
orc_decoder* orc_nxdn_new(void);
orc_decoder* orc_pocsag_new(void);
orc_decoder* orc_dstar_new(void);
void orc_decoder_free(orc_decoder*);
void orc_dmr_set_slot_filter(orc_decoder*, uint8_t filter);
/* streaming: consumes symbols from in[0..n); returns symbols consumed.  Output
 * bytes appended to out (cap), events to ev (ev_cap). */
size_t orc_decoder_process(orc_decoder*, const uint8_t* in, size_t n,
                           uint8_t* out, size_t cap, size_t* n_out,
                           orc_event* ev, size_t ev_cap, size_t* n_ev);

/* ----------------------------------------------------------------- pipe.c */
/* whole-chain helper used by bench.py's cpu_baseline leg and by tests:
 * rrc(wide|narrow|none) -> demod(levels,sps) -> decoder(proto) for n_channels
 * independent channels laid out [n_channels][stride] floats, on n_threads
 * pthreads (one channel per thread at a time). Returns 0 on success. */
typedef struct {
    int rrc;        /* 0 none, 1 wide, 2 narrow */
    int levels;     /* 4 gfsk, 2 fsk, 0 = no demod */
    int invert;
    unsigned sps;
    int proto;      /* 0 none, 1 DMR, 2 YSF, 3 NXDN, 4 POCSAG, 5 D-Star */
    int slot_filter;
} orc_chain_cfg;

int orc_chain_run(const orc_chain_cfg* cfg, const float* in, size_t n_channels, size_t stride, size_t n,
                  float* filtered /* [n_channels][stride] or NULL */,
                  uint8_t* syms, size_t sym_stride, uint32_t* sym_count,
                  uint8_t* out, size_t out_stride, uint32_t* out_count,
                  orc_event* ev, size_t ev_stride, uint32_t* ev_count,
                  int n_threads);

#ifdef __cplusplus
}
#endif
#endif
