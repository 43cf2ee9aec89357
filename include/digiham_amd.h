/*
 * digiham_amd.h -- C ABI of the MI355X-native many-channel digital-voice
 * demodulation / FEC engine (libdigiham_amd.so).
 *
 * This is the drop-in boundary for digiham's hot path
 *     rrc_filter -> gfsk_demodulator / fsk_demodulator -> dmr_decoder / ysf_decoder
 * (+ digitalvoice_filter).  Plain C types, caller-owned buffers, `int` return
 * (0 = ok, negative = DH_E*), no exceptions across the ABI.  The host-side
 * C++ classes in include/digiham/ (same names and constructor signatures as
 * the reference's include/ headers) and the Python binding in digiham_amd/ sit on
 * top of exactly these entry points.
 *
 * Pointer conventions
 *   d_*   device pointers (HBM; hipMalloc / torch CUDA tensors)
 *   h_*   host pointers
 *   stream: a hipStream_t passed as void* (NULL = the HIP default stream)
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the reference root, jketterl/digiham v0.7.0-dev).
 */
#ifndef DIGIHAM_AMD_H
#define DIGIHAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH_OK        0
#define DH_EINVAL   -1   /* bad argument */
#define DH_ENOMEM   -2   /* device allocation failed */
#define DH_EDEVICE  -3   /* HIP runtime error (see dh_last_error) */
#define DH_ENODEV   -4   /* no gfx950 device visible */
#define DH_ECAPACITY -5  /* an output buffer overflowed; results truncated */

const char* dh_version(void);
/* last HIP error string recorded on this thread ("" if none) */
const char* dh_last_error(void);
/* number of visible HIP devices, or a negative DH_E* code */
int dh_device_count(void);

/* small memory helpers so that C / C++ hosts need no HIP headers (synchronous) */
int dh_device_alloc(int device, size_t bytes, void** d_out);
int dh_device_free(void* d_ptr);
int dh_copy_to_host(void* h_dst, const void* d_src, size_t bytes);
int dh_copy_to_device(void* d_dst, const void* h_src, size_t bytes);

/* ------------------------------------------------------------------------
 * Stateless batch FEC kernels (one codeword / block per lane).
 * Replace the C functions under src/dmr_decoder and src/ysf_decoder.
 * All arrays are device pointers; `ok` receives 1/0 per item.
 * ---------------------------------------------------------------------- */
/* bool hamming_7_4(uint8_t*)      src/dmr_decoder/hamming_7_4.c:56-72  */
int dh_hamming_7_4(uint8_t* d_words, uint8_t* d_ok, size_t n, void* stream);
/* bool hamming_13_9(uint16_t*)    src/dmr_decoder/hamming_13_9.c:70-84 */
int dh_hamming_13_9(uint16_t* d_words, uint8_t* d_ok, size_t n, void* stream);
/* bool hamming_15_11(uint16_t*)   src/dmr_decoder/hamming_15_11.c:74-88 */
int dh_hamming_15_11(uint16_t* d_words, uint8_t* d_ok, size_t n, void* stream);
/* bool hamming_16_11(uint16_t*)   src/dmr_decoder/hamming_16_11.c:79-93 */
int dh_hamming_16_11(uint16_t* d_words, uint8_t* d_ok, size_t n, void* stream);
/* bool quadratic_residue(uint16_t*) src/dmr_decoder/quadratic_residue.c:321-335 */
int dh_quadratic_residue(uint16_t* d_words, uint8_t* d_ok, size_t n, void* stream);
/* bool golay_20_8(uint32_t*)      src/dmr_decoder/golay_20_8.c:1421-1435 */
int dh_golay_20_8(uint32_t* d_words, uint8_t* d_ok, size_t n, void* stream);
/* bool golay_24_12(uint32_t*)     src/ysf_decoder/golay_24_12.c:2401-2415 */
int dh_golay_24_12(uint32_t* d_words, uint8_t* d_ok, size_t n, void* stream);
/* bool bch_31_21(uint32_t*)       src/pocsag_decoder/bch_31_21.c:545-561 (31-bit words, up to 2 errors) */
int dh_bch_31_21(uint32_t* d_words, uint8_t* d_ok, size_t n, void* stream);
/* bool bptc_196_96(uint8_t payload[25], uint8_t output[12])  src/dmr_decoder/bptc_196_96.c:5-59
 * d_in [n][25] -> d_out [n][12] (zero-filled when ok == 0) */
int dh_bptc_196_96(const uint8_t* d_in, uint8_t* d_out, uint8_t* d_ok, size_t n, void* stream);
/* uint8_t decode_trellis(uint8_t* input, uint8_t size, uint8_t* output)  src/ysf_decoder/trellis.c:32-109
 * d_in [n][in_stride] dibits packed 4 per byte MSB first; n_dibits <= 192;
 * d_out [n][out_stride] decoded bits (ceil(n_dibits/8) bytes used); d_metric [n] */
int dh_trellis(const uint8_t* d_in, size_t in_stride, int n_dibits,
               uint8_t* d_out, size_t out_stride, uint8_t* d_metric, size_t n, void* stream);
/* uint16_t crc16_checksum(uint8_t* data, int count)  src/ysf_decoder/crc16.c:3-18 */
int dh_crc16(const uint8_t* d_in, size_t stride, int count, uint16_t* d_out, size_t n, void* stream);
/* void decode_whitening(uint8_t* in, uint8_t* out, uint8_t num)  src/ysf_decoder/whitening.c:6-22
 * in/out [n][stride]; ceil(n_bits/8) bytes per row are written */
int dh_whitening(const uint8_t* d_in, uint8_t* d_out, size_t stride, int n_bits, size_t n, void* stream);

/* ------------------------------------------------------------------------
 * Streaming engine: B independent channels, per-channel state resident in HBM.
 * Replaces one `rrc_filter | gfsk_demodulator | dmr_decoder` process chain
 * per channel (examples/dmr-decoder.sh:19-23, examples/ysf-decoder.sh:19-23):
 *   Digiham::RrcFilter::{Wide,Narrow}RrcFilter::process   src/rrc_filter/rrc_filter.cpp:16-34
 *   Digiham::Fsk::GfskDemodulator::process                src/gfsk_demodulator/gfsk_demodulator.cpp:24-122
 *   Digiham::Fsk::FskDemodulator::process                 src/fsk_demodulator/fsk_demodulator.cpp:25-112
 *   Digiham::Decoder::process + Dmr/Ysf phases            src/lib/decoder.cpp:21-47,
 *                                                         src/dmr_decoder/dmr_phase.cpp:18-302,
 *                                                         src/ysf_decoder/ysf_phase.cpp:16-349
 * ---------------------------------------------------------------------- */
typedef struct dh_engine dh_engine;

enum { DH_RRC_NONE = 0, DH_RRC_WIDE = 1, DH_RRC_NARROW = 2,
       DH_RRC_CUSTOM = 3 };   /* the caller's coefficient table: RrcFilter(nZeros, gain, coeffs[]), include/rrc_filter.hpp:12 */
enum { DH_DEMOD_NONE = 0, DH_DEMOD_FSK2 = 2, DH_DEMOD_GFSK4 = 4 };
enum { DH_PROTO_NONE = 0, DH_PROTO_DMR = 1, DH_PROTO_YSF = 2, DH_PROTO_NXDN = 3, DH_PROTO_POCSAG = 4, DH_PROTO_DSTAR = 5 };

/* flags */
#define DH_FLAG_FAST_FIR        0x1   /* FMA FIR: float outputs within 1e-6 of the reference, dibits NOT guaranteed bit-exact */
#define DH_FLAG_KEEP_FILTERED   0x2   /* also materialise the RRC output [B][n] (unfused path; BASELINE config 2) */
#define DH_FLAG_FSK_INVERT      0x4   /* FskDemodulator(sps, invert = true) */
#define DH_FLAG_NO_EVENTS       0x8   /* do not record decoder events */
#define DH_FLAG_SPLIT_STAGES    0x20  /* launch slicer and decoder as two kernels even where the one-wavefront chain kernel exists */
#define DH_FLAG_ORDERED_TIMING  0x10  /* always run the in-order variance chain of the timing recovery (diagnostic; results are identical) */
#define DH_FLAG_EXACT_SYMBOLS   0x40  /* error-bounded kernels: decide EVERY symbol with the reference's arithmetic (diagnostic; results are identical) */
#define DH_FLAG_EXACT_FIR       0x80  /* error-bounded kernels: run the rounded-product FIR in every run (diagnostic / A-B; results are identical) */
#define DH_FLAG_ONE_LAUNCH      0x200 /* with DH_FLAG_KEEP_FILTERED on the wide filter at 10 samples per symbol: `rrc_filter | gfsk_demodulator` as ONE kernel per push -- the
                                         error-bounded slicer kernel (dibits bit-exact, as without the flag) also stores the filtered samples it holds in LDS.  They come from
                                         its split-f16 matrix-core FIR: within 2.5e-6 of the reference's floats relative to max(|ref|, rms(ref)) (measured 1.0e-6), which is
                                         NOT the 1e-6 of DH_FLAG_FAST_FIR and not bit-exact.  With DH_FLAG_FAST_FIR set as well the one kernel filters with the f32 FMA chain
                                         instead (on the matrix cores, v_mfma_f32_16x16x4_f32: the floats of DH_FLAG_FAST_FIR, within 1e-6, measured 5e-7) and -- unlike
                                         DH_FLAG_FAST_FIR alone -- still delivers the reference's dibits bit for bit (error radius + exact re-evaluation of what it leaves in
                                         doubt): BASELINE configs[1] in one launch.  (The reference's own 81-term float chain is 8e-7 away from the exact
                                         convolution and the split-f16 FIR 1.3e-7: most of the distance is the reference's rounding, which only its own order of operations reproduces.) */
#define DH_FLAG_OVERLAP_PUSHES  0x100 /* engines of >= 8192 channels on the one-launch chains (DMR, YSF, NXDN, D-Star): a push goes out as two launches on two streams of the engine's
                                         own (three quarters of the channels at high priority, the rest at normal priority) which are ordered
                                         after the caller's stream at the moment of the push and joined with it again only when results are
                                         read, the engine is reset / synchronised or another kind of work is queued -- so the drain of one
                                         launch is filled by the next, across pushes.  CONTRACT: the input buffer of a push must stay
                                         untouched until dh_engine_sync() (or any read) returns, and the raw device
                                         views of the outputs (dh_engine_symbols / _frames / _events) are only valid after dh_engine_sync().
                                         Results are identical. */

typedef struct {
    uint32_t struct_size;     /* = sizeof(dh_engine_config) */
    int32_t  device;          /* HIP device ordinal */
    uint32_t n_channels;      /* B */
    uint32_t max_samples;     /* largest n a single push may carry (per channel) */
    int32_t  rrc;             /* DH_RRC_* */
    int32_t  demod;           /* DH_DEMOD_* */
    uint32_t sps;             /* samples per symbol (GfskDemodulator / FskDemodulator ctor argument) */
    int32_t  proto;           /* DH_PROTO_* */
    uint32_t flags;
    uint32_t slot_filter;     /* Dmr::Decoder::setSlotFilter initial value (3 = both slots) */
    void*    stream;          /* hipStream_t all work is enqueued on (NULL: default stream) */
    /* rrc == DH_RRC_CUSTOM only (struct_size must cover these fields): y[n] = (float)((double) sum_i rrc_taps[i] x[n - nZeros + i] / rrc_gain),
     * products and sums rounded to float one by one in tap order (src/rrc_filter/rrc_filter.cpp:22-34).  The table
     * (host memory, rrc_nzeros + 1 floats, 1 <= rrc_nzeros <= 160, any shape) is copied at create time.  A custom filter
     * is never fused into the slicer: the filtered signal is materialised and the demodulator (if any) reads it. */
    const float* rrc_taps;
    uint32_t rrc_nzeros;
    double   rrc_gain;
} dh_engine_config;
/* sizeof(dh_engine_config) before the custom-filter fields were added: still accepted as struct_size */
#define DH_ENGINE_CONFIG_V1_SIZE offsetof(dh_engine_config, rrc_taps)

/* Decoder event: one record per call the reference makes into its MetaCollector,
 * plus the FEC-corrected words feeding it (BPTC LC, slot type, EMB, FICH, DCH). */
typedef struct {
    uint32_t sym_index;       /* absolute symbol index of the frame start (mod 2^32) */
    uint8_t  type;            /* DH_EV_* */
    uint8_t  a;               /* DMR slot / YSF frame number or CSD index */
    uint8_t  b;               /* sub-type (sync type, data type, lcss, reset cause) */
    uint8_t  len;             /* valid payload bytes */
    uint8_t  payload[24];
} dh_event;

enum {
    /* DMR (src/dmr_decoder/dmr_phase.cpp): SLOT_RESET b = 1 when it is the OTHER slot that is reset after a TACT slot switch
     * (:80 -- the reference leaves that slot's talker alias collector alone until its next burst), 0 otherwise */
    DH_EV_DMR_SYNC = 1, DH_EV_DMR_SLOT_RESET = 2, DH_EV_DMR_META_RESET = 3, DH_EV_DMR_LC = 4,
    DH_EV_DMR_SOFT_RESET = 5, DH_EV_DMR_BPTC = 6, DH_EV_DMR_SLOTTYPE = 7, DH_EV_DMR_EMB = 8,
    DH_EV_YSF_FICH = 16, DH_EV_YSF_MODE = 17, DH_EV_YSF_DCH = 18, DH_EV_YSF_HEADER_DCH = 19,
    DH_EV_YSF_META_RESET = 20,
    /* NXDN48 (src/nxdn_decoder/nxdn_phase.cpp): LICH byte; SACCH fragment (a = structure index, 5 bytes); complete SACCH
     * superframe (9 bytes); setSync("voice"); FACCH1 (a = block, 12 bytes); MetaCollector::reset (b = 0 sync loss, 1 TX_RELEASE) */
    DH_EV_NXDN_LICH = 32, DH_EV_NXDN_SACCH = 33, DH_EV_NXDN_SACCH_SF = 34, DH_EV_NXDN_SYNC_VOICE = 35,
    DH_EV_NXDN_FACCH1 = 36, DH_EV_NXDN_META_RESET = 37,
    /* POCSAG (src/pocsag_decoder/pocsag_phase.cpp:56-57): a = position in the batch, payload = BCH-corrected codeword (big endian) */
    DH_EV_POCSAG_CODEWORD = 48,
    /* D-Star (src/dstar_decoder/dstar_phase.cpp): HEADER = setFromHeader of a CRC-valid voice header, 41 bytes over two
     * events (a = 0: bytes 0-23, a = 1: bytes 24-40; b = 0 radio header (:50), 1 slow-data header (:214)); VOICE_START =
     * a new VoicePhase (b = 1 after a header, 0 after a voice sync); SYNC_VOICE = setSync("voice") (:110); MESSAGE = the
     * 20-character slow-data message (:207); SIMPLE = len simple-data bytes appended (:178); FRAME_SYNC = parseFrameData
     * (:113): the consumer now parses its simple-data lines (DPRS / NMEA, :218-245); META_RESET = MetaCollector::reset
     * (b = 0 terminator, 1 sync lost) */
    DH_EV_DSTAR_HEADER = 64, DH_EV_DSTAR_VOICE_START = 65, DH_EV_DSTAR_SYNC_VOICE = 66, DH_EV_DSTAR_MESSAGE = 67,
    DH_EV_DSTAR_SIMPLE = 68, DH_EV_DSTAR_FRAME_SYNC = 69, DH_EV_DSTAR_META_RESET = 70
};

int  dh_engine_create(const dh_engine_config* cfg, dh_engine** out);
void dh_engine_destroy(dh_engine* e);
/* back to the freshly-constructed state of every module (zeroed delay lines, SyncPhase) */
int  dh_engine_reset(dh_engine* e);
/* Dmr::Decoder::setSlotFilter (src/dmr_decoder/dmr_decoder.cpp:9-15) for every channel */
int  dh_engine_set_slot_filter(dh_engine* e, uint32_t filter);

/* Feed n new samples per channel (n <= max_samples).  d_samples is [B][stride]
 * float32, channel-major, resident in HBM.  Asynchronous on the engine stream.
 * Outputs of this push replace those of the previous push.  n = 0 is a valid (empty) push; d_samples may then be NULL. */
int  dh_engine_push(dh_engine* e, const float* d_samples, size_t stride, size_t n);
/* same, from host memory (staged through an engine-owned device buffer) */
int  dh_engine_push_host(dh_engine* e, const float* h_samples, size_t stride, size_t n);
/* One channel of a many-channel engine: back to the freshly-constructed state / Dmr::Decoder::setSlotFilter for it alone
 * (the other channels keep streaming).  What a module instance attached to a shared engine needs. */
int  dh_engine_reset_channel(dh_engine* e, uint32_t channel);
int  dh_engine_set_slot_filter_channel(dh_engine* e, uint32_t channel, uint32_t filter);
/* Ragged pushes: channel b brings d_counts[b] (<= max_n) new samples of its row; the others stay where they are.  One launch
 * for N module instances whose ring buffers hold different amounts (include/digiham/shared_engine.hpp); a real-time receiver
 * whose channels arrive in blocks of unequal length.  Engines with a foreign tap table (DH_RRC_CUSTOM) take whole pushes only. */
int  dh_engine_push_ragged(dh_engine* e, const float* d_in, size_t stride, const uint32_t* d_counts, size_t max_n);
int  dh_engine_push_host_ragged(dh_engine* e, const float* h_in, size_t stride, const uint32_t* h_counts, size_t max_n);

/* Device views of the current push's outputs (valid until the next push).
 * Any out-pointer may be NULL. counts are uint32 [B]. */
int  dh_engine_filtered(dh_engine* e, const float** d_filtered, size_t* stride);           /* needs DH_FLAG_KEEP_FILTERED */
int  dh_engine_symbols(dh_engine* e, const uint8_t** d_syms, size_t* stride, const uint32_t** d_count);
int  dh_engine_frames(dh_engine* e, const uint8_t** d_bytes, size_t* stride, const uint32_t** d_count);
int  dh_engine_events(dh_engine* e, const dh_event** d_events, size_t* stride, const uint32_t** d_count);
/* Host copies for one channel (synchronises the engine stream). *n in: capacity, out: count */
int  dh_engine_read_symbols(dh_engine* e, uint32_t channel, uint8_t* h_out, size_t* n);
int  dh_engine_read_frames(dh_engine* e, uint32_t channel, uint8_t* h_out, size_t* n);
int  dh_engine_read_events(dh_engine* e, uint32_t channel, dh_event* h_out, size_t* n);
int  dh_engine_read_filtered(dh_engine* e, uint32_t channel, float* h_out, size_t* n);
/* Per-stage device timing (HIP events recorded on the engine stream around each stage of a push):
 * enable once with the number of pushes to keep; read returns, for each recorded push, the
 * milliseconds spent in the stand-alone RRC stage (0 when fused), the slicer (fused RRC+GFSK)
 * kernel and the decoder kernel.  *n in: capacity of the arrays, out: pushes recorded. */
int  dh_engine_timing_enable(dh_engine* e, uint32_t max_pushes);
int  dh_engine_timing_read(dh_engine* e, float* rrc_ms, float* slicer_ms, float* decoder_ms, uint32_t* n);
/* For pushes that went out as two launches (DH_FLAG_OVERLAP_PUSHES): duration of the FIRST launch alone (events on the
 * engine's high-priority stream) and the number of channels it covered (0 / 0 for a push with one launch).  Call BEFORE
 * dh_engine_timing_read, which starts a new series; the stage times of such a push on the caller's stream are ~0. */
int  dh_engine_timing_read_split(dh_engine* e, float* first_ms, uint32_t* first_channels, uint32_t* n);
/* Timing-recovery statistics since create/reset, per channel (host arrays of n_channels, either may be NULL):
 * blocks = 100-symbol variance blocks evaluated (gfsk_demodulator.cpp:41-80), ordered = those in which the
 * error-bounded estimate could not separate the phases and the reference's in-order sums decided. Synchronises. */
int  dh_engine_timing_stats(dh_engine* e, uint32_t* h_blocks, uint32_t* h_ordered);
/* Diagnostic: word `word` (0..31) of every channel's slicer state header, or word `word - 100` (0..63) of its
 * decoder state, or word `word - 200` (0..15) of the engine's own flag block (the same value for every channel; 202 =
 * hand-overs inside split launches that did not come and were finished by the fix-up launch), into h_out[n_channels].
 * Synchronises. */
int  dh_engine_debug_header(dh_engine* e, uint32_t word, uint32_t* h_out);
/* wait for all enqueued work; returns DH_ECAPACITY if any channel overflowed an output buffer */
int  dh_engine_sync(dh_engine* e);

/* Standalone decoder stage over symbols (for pipes that already have dibits):
 * d_syms [B][stride] uint8 dibits (values 0..3), d_count[B] valid symbols per channel. Requires
 * proto != NONE and an engine created with rrc = NONE, demod = DH_DEMOD_NONE (max_samples then
 * bounds the symbols per push). */
int  dh_engine_push_symbols(dh_engine* e, const uint8_t* d_syms, size_t stride, const uint32_t* d_count);

/* ------------------------------------------------------------------------
 * Digital voice post-filter, B independent int16 streams, state carried.
 * Replaces Digiham::DigitalVoice::DigitalVoiceFilter::process
 * (src/digitalvoice_filter/digitalvoice_filter.cpp:6-10,34-45).
 * d_state: [B][22] floats (xv[11], yv[11]) zero-initialised by the caller.
 * ---------------------------------------------------------------------- */
int dh_dvfilter_s16(const int16_t* d_in, int16_t* d_out, float* d_state, size_t n_channels, size_t stride, size_t n, void* stream);

/* ------------------------------------------------------------------------
 * Receiver front-end, B independent streams, state carried: what examples/dmr-decoder.sh:13-17 runs in front of
 * rrc_filter -- `rtl_fm -M fm -s 48000 | csdr convert -i s16 -o float | csdr dcblock` (third-party tools, no source in the
 * reference: the arithmetic is this library's own specification, digiham_amd/csrc/frontend_core.hpp).
 *   mode DH_FE_AUDIO_S16  d_in [B][in_stride] int16 FM-discriminator audio            -> x = s16 / 32768
 *   mode DH_FE_IQ_S16     d_in [B][in_stride] int16, interleaved I / Q (2 n values)    -> x = arg(z[n] conj(z[n-1])) / pi
 *   dcblock != 0          y[n] = (x[n] - x[n-1]) + 0.995 y[n-1]
 * d_out [B][out_stride] float32 is what dh_engine_push takes.  d_state: [B][4] floats, zero-initialised by the caller.
 * ---------------------------------------------------------------------- */
enum { DH_FE_AUDIO_S16 = 1, DH_FE_IQ_S16 = 2 };
int dh_frontend_s16(const int16_t* d_in, size_t in_stride, float* d_out, size_t out_stride, float* d_state,
                    size_t n_channels, size_t n, int mode, int dcblock, void* stream);

/* ------------------------------------------------------------------------
 * Diagnostics: the RRC output scaling `(float)((double)sum / gain)` of
 * src/rrc_filter/rrc_filter.cpp:33 exactly as the FIR kernels evaluate it
 * (reciprocal multiply + exact-division fallback near float rounding ties).
 * d_out[i] must equal (float)((double)d_in[i] / gain) for every float.
 * ---------------------------------------------------------------------- */
int dh_debug_div_gain(const float* d_in, float* d_out, size_t n, int narrow, void* stream);
/* The slicer's `volume_sum / samplesPerSymbol` (src/gfsk_demodulator/gfsk_demodulator.cpp:83) exactly as the kernels
 * evaluate it (reciprocal multiply + exact FMA residual + correction; IEEE division for 0, tiny, huge and non-finite
 * operands).  d_out[i] must equal d_in[i] / (float) divisor for every float. */
int dh_debug_div_const(const float* d_in, float* d_out, size_t n, unsigned divisor, void* stream);
/* The matrix-core instruction behind the error-bounded wide-filter FIR (v_mfma_f32_16x16x32_f16), one tile per item:
 * d_d[t][16][16] = d_c[t][16][16] + d_a[t][16][32] x d_b[t][32][16], A and B as IEEE binary16 bit patterns.  The bound of
 * that FIR rests on a stated assumption about how the hardware adds the products (dsp_core.hpp, "(H1)"): the parity tests
 * check it through this entry. */
int dh_debug_mfma_f16(const uint16_t* d_a, const uint16_t* d_b, const float* d_c, float* d_d, size_t tiles, void* stream);
/* The split of a scaled sample into two halves as the same kernels do it: d_h1[i] = f16(x scale), d_h2[i] =
 * f16((x scale - h1) 2^11), both rounded to nearest even, subnormal halves kept. */
int dh_debug_f16_split(const float* d_in, uint16_t* d_h1, uint16_t* d_h2, size_t n, float scale, void* stream);
/* A plain streaming copy of n_bytes (a multiple of 16, both pointers 16-byte aligned): 16 bytes per lane, non-temporal
 * loads and stores, grid-stride over 2 048 workgroups.  Not part of the path -- bench.py times it on the lease it runs on
 * as the achievable HBM ceiling (read + write bytes over its duration) that the path's kernels are priced against beside
 * the 8 TB/s of the data sheet.  d_dst == NULL: the bytes are only read (eight loads in flight per lane, 8 192 workgroups) -- the read-only
 * streaming rate, which is what a kernel that mostly reads (the chain kernels) can be priced against. */
int dh_debug_copy(const void* d_src, void* d_dst, size_t n_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
