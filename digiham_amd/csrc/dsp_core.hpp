// dsp_core.hpp -- RRC FIR + GFSK/FSK symbol slicer, one channel per wavefront.
//
// Arithmetic contract (bit-exact dibits need bit-exact floats):
//   FIR      y[n] = (float)((double)(((0 + c0*x[n-N]) + c1*x[n-N+1]) + ... + cN*x[n]) / gain)
//            every product and every sum separately rounded to float, taps in order
//            (src/rrc_filter/rrc_filter.cpp:22-34).  Built with -ffp-contract=off; the only
//            fused variant is the explicitly requested DH_FLAG_FAST_FIR path (__builtin_fmaf).
//   slicer   window sums in sample order, float; AGC thresholds in double, stored to float;
//            timing variance in double, 100-term sums in order
//            (src/gfsk_demodulator/gfsk_demodulator.cpp:24-122).
//
// Block-parallel formulation of the symbol loop: the +-1 sample timing step is
// decided only at the 100th symbol of a variance block and applied after the first symbol of
// the next block, so inside one block every symbol window position is known up front:
//   start[k] = p + (k - k0)*sps + (k0 == 0 && k > 0 ? pending_offset : 0)
// Lanes take symbols; the 100-entry sliding AGC min/max becomes
//   min_k = min( prefix-min over {old[0..k0), new[k0..k]}, suffix-min over old(k..99] )
// (two wave scans: min/max are exact in any order), and the per-phase variance runs on `sps`
// lanes at the end of the block.
#pragma once

#include "dh_portable.hpp"

#define DH_VARIANCE_SYMBOLS 100      // include/gfsk_demodulator.hpp:5
#define DH_VOLUME_RB_SIZE 100        // include/gfsk_demodulator.hpp:6
#define DH_FTILE 1024                // filtered samples produced per FIR pass (64 lanes x 16)
#define DH_FIR_L 16                  // consecutive outputs per lane
#define DH_PF_N 5                    // 16-byte prefetch loads per lane covering 1024 + 160 samples
#define DH_PF_SINK 512               // word offset inside the window block where the L2-touch loads drop their dwords (64 words)
#define DH_MAX_NZ 160
#define DH_MAX_SPS 40
// raw samples kept BEHIND the read position by the error-bounded kernels (see DH_BOUNDED_FIR): the 100 symbols of the
// rings + slack; and the most the tail carries between pushes: history + the last nz inputs + not yet consumed ones
DH_HD uint32_t dh_history(uint32_t sps) { return DH_VARIANCE_SYMBOLS * (sps > 10u ? sps : 10u) + 152u; }
DH_HD uint32_t dh_tail_max(uint32_t sps) { return 256u + dh_history(sps); }
#define DH_STATE_HDR 32              // u32 words of per-channel header

// per-channel state block in HBM (floats / u32 words, AoS, `state_stride` words apart):
//   [0..15]                      header: k, pending offset, tail count, symbols produced (lo),
//                                timing blocks decided by the ordered chain, timing blocks total, ...
//   [16 .. 16+100)               volume ring  (volume_rb)
//   [116 .. 116+100*sps)         variance ring (variance_rb), phase-major: [sample i][symbol k]
//   [.. + dh_tail_max(sps))      raw-sample tail: the last nz inputs + not yet consumed samples
enum { DH_ST_K = 0, DH_ST_OFF = 1, DH_ST_TAIL = 2, DH_ST_NSYM = 3, DH_ST_ORDERED = 4, DH_ST_BLOCKS = 5,
       DH_ST_P0 = 6,                 // read position inside the tail at the start of the next push (= history samples in front of it)
       DH_ST_CUR_START = 7, DH_ST_CUR_OFF = 8, DH_ST_PREV_START = 9, DH_ST_PREV_OFF = 10,   // filtered positions of symbol 0 of the current / previous variance block, their offsets
       DH_ST_BLOCK_FLAGS = 11,       // bit 0: the current block's start is known, bit 1: the previous block's
       DH_ST_E_CUR = 12, DH_ST_E_PREV = 13, DH_ST_E_COUNT = 14, DH_ST_E_BLOCK = 15,          // error radii of the ring entries (floats), symbols in the current bucket
       DH_ST_UNCERTAIN = 16, DH_ST_EXACT_RUNS = 17, DH_ST_EXACT_BLOCKS = 18,                    // statistics: symbols / runs / timing blocks decided by exact arithmetic
       DH_ST_PART = 19,              // tail split (k_chain): epoch of the last split push (24 bits) | parts written back << 24 | a later part gave up << 26 | XCC id << 28
       DH_ST_DIAG = 20 };            // 20..31: diagnostic builds (phase clocks 20..27, wave timeline 28..31)
#define DH_ST_VOL DH_STATE_HDR
#define DH_ST_VAR (DH_STATE_HDR + DH_VOLUME_RB_SIZE)

struct DhDspParams {
    const float* in; size_t in_stride; uint32_t n;     // n new samples per channel (ragged pushes: the most any channel brings)
    const uint32_t* n_per;                             // ragged pushes (dh_engine_push_ragged): [B] samples of each channel, <= n; or null
    float* state; size_t state_stride;                 // per-channel state (see above), in 4-byte words
    uint8_t* syms; size_t sym_stride;                  // symbol output [B][sym_stride]
    uint32_t* sym_count;                               // [B] symbols produced by this push
    uint32_t sym_cap;                                  // max symbols a push may append per channel
    uint32_t* overflow;                                // set to 1 if sym_cap was hit
    uint32_t n_channels;
    uint32_t ch_base;                                  // first channel of this launch (a push may go out as two launches, engine.hip)
    uint32_t sps, lo, hi;                              // samples/symbol, [lo,hi) = mid-symbol evaluation window
    int32_t levels, invert;                            // 4 = GFSK, 2 = FSK
    uint32_t nz;                                       // FIR order (0 = no RRC stage)
    int32_t fast;                                      // 1 = FMA FIR
    int32_t ordered_timing;                            // 1 = always run the ordered variance chain (DH_FLAG_ORDERED_TIMING)
    int32_t exact_mode;                                // error-bounded kernels: 0 normal, 1 every symbol decided by exact arithmetic, 2 exact FIR in every run
    float err_coef;                                    // error radius of a filtered sample per unit of max |x| (dh_fir_error_coefficient)
    double gain, rgain; float inv_gain;                // rgain = 1/gain rounded to double
    float taps[DH_MAX_NZ / 2 + 1];                     // first half + centre of the symmetric response
    const uint32_t* tapfrag;                           // split-f16 FIR: the per-lane tap fragments (DhF16Taps::frag), or null
    float err_coef_f16;                                // its error radius per unit of max |x| (dh_f16_error_coefficient)
    // tail split (engine.hip, HipBackend::go_chain): the launch has two or three workgroups per channel; workgroup
    // k * split_pad + channel takes the samples [0, split_n0), [split_n0, split_n1 or the end), [split_n1, end) of the
    // channel's row for k = 0, 1, 2.  split_n0 = 0: one workgroup per channel; split_n1 = 0: two.
    uint32_t split_n0, split_n1, split_pad, part_epoch;
    // split_fixup = 1: the launch BEHIND a split launch, one workgroup per channel -- a channel whose later parts did not get
    // their hand-over (flag never came, or came from another XCD) is finished here, unsplit, from where its last completed part
    // stopped; every other workgroup leaves at once.  split_force_fail = k > 0 (tests: DH_TAIL_SPLIT_FORCE_FAIL): the SECOND parts
    // of the channels with ch % k == 1 give up without looking.
    uint32_t split_fixup, split_force_fail;
    // DH_FLAG_KEEP_FILTERED | DH_FLAG_ONE_LAUNCH on the wide filter at sps 10 (without DH_FLAG_FAST_FIR): the error-bounded slicer also delivers
    // the filtered samples of the push it has in LDS anyway -- filt_out[ch][t] for every new sample t -- in ONE launch.  They are the split-f16
    // FIR's: within 2.5e-6 of the reference's (measured 1.0e-6), NOT the 1e-6 of DH_FLAG_FAST_FIR / BASELINE configs[1]
    float* filt_out; size_t filt_stride;
};

// (behind the tail: two words per phase that round 4's ring-less experiment kept there; the layout is unchanged)
#define DH_PART_WORDS(sps) ((2u * (sps) + 4u + 3u) & ~3u)
DH_HD uint32_t dh_state_part_offset(uint32_t sps) { return DH_ST_VAR + DH_VARIANCE_SYMBOLS * sps + dh_tail_max(sps); }
DH_HD uint32_t dh_state_words(uint32_t sps) { return dh_state_part_offset(sps) + DH_PART_WORDS(sps); }

// LDS block of one wavefront.  `xf` holds the raw-sample window during the FIR (padded one word
// per 16 so that the per-lane sliding windows, 16 words apart, hit 32 different banks) and is then
// overwritten in place by the filtered samples of the run.
#define DH_XPAD(i) ((i) + ((i) >> 4))
#define DH_SCAN_N 128
// The LDS block of one wavefront, carved by dh_dsp_carve():
//   vol_old[128] vol_new[128]                                fixed part, then
//   wide filter / no filter:  tap table[112] bound[12] stats[2] pad[2] (clk[8] in DH_PHASE_CLOCKS builds)
//   narrow filter:            tapsf[84] stats[2] clk[8] pad[2] bound[16]
//   var_rb[100 * sps]     variance ring, phase-major: row i holds sample i of the last 100 symbols
//   xf[...]               raw-sample window during the FIR (padded one word per 16 so that the per-lane sliding
//                         windows, 16 words apart, hit 32 different banks), then overwritten by the filtered samples
// The wide filter's tap table is the WHOLE response between two runs of 15 zeros (tap k at word 15 + k): the matrix-pipe
// FIR (dh_fir_mfma) reads its Toeplitz operand c[p - n], p = 0..95, n = 0..15, straight out of it; `tapsf` points at tap 0,
// so everything that indexes the first half + centre still works.
// Everything that only lives between the window phase (P3) and the next staging (P7) sits INSIDE the window block,
// which is idle then: the AGC extremes mn / mx (also the scratch of the timing estimate) in its first 256 words
// (the filtered samples there are dead once P3 has copied them into the ring), the ordered-chain variances behind
// them, and the mid-symbol sums behind the last filtered sample.  10 240 bytes per wavefront for the wide filter at
// sps 10: exactly 16 wavefronts per CU.
struct DhDspShared {
    float* xf;
    float* vol_old;                                    // ring content before the current run (+ identity padding)
    float* vol_new;                                    // entries written by the current run
    float* mn; float* mx;                              // AGC window min / max per symbol of the block
    float* sum;                                        // mid-symbol window sums of the current run
    double* variance;                                  // [DH_MAX_SPS] per-phase variances of the ordered chain
    uint32_t* stats;                                   // timing blocks of this push: all / decided by the ordered chain
    uint32_t* clk;                                     // DH_PHASE_CLOCKS builds only
    float* tapsf;                                      // FIR taps (first half + centre; the wide filter: all of them, see above)
    float* var_rb;
    float* bound;                                      // bookkeeping of the error-bounded kernels (DhBoundState, 16 words)
};

#define DH_TAP_LEAD 15                                 // zeros in front of (and behind) the wide filter's tap table
#define DH_LDS_CLK_WORDS 0u
// words in front of the variance ring (a multiple of 4: keeps var_rb 16-byte aligned)
DH_HD uint32_t dh_lds_fixed_words(uint32_t nz) { return nz > 80u ? 352u + 16u : 384u + DH_LDS_CLK_WORDS; }
DH_HD uint32_t dh_dsp_xf_words(uint32_t nz) {
    const uint32_t padded = DH_XPAD(DH_FTILE + nz) + 1u, with_sums = DH_FTILE + 4u + DH_SCAN_N;
    return ((padded > with_sums ? padded : with_sums) + 3u) & ~3u;
}
#define DH_RING_WORDS(sps) (DH_VARIANCE_SYMBOLS * (sps))
// (Round 4 measured kernels WITHOUT the variance ring in LDS -- per-phase sums carried instead, 6.3 KB per wavefront: the sums cost the window
// phase more than the ring costs, +4 % on the DMR chain, and a fifth wavefront per SIMD at 96 registers spills across the FIR.  Removed in
// round 5; profiles/r04_b_ab_logs.txt.)
DH_HD size_t dh_dsp_shared_bytes(uint32_t sps, uint32_t nz) {
    return sizeof(float) * (size_t) (dh_lds_fixed_words(nz) + DH_RING_WORDS(sps) + dh_dsp_xf_words(nz));
}
DH_HD DhDspShared dh_dsp_carve(void* base, uint32_t sps, uint32_t nz = 0) {          // base: 16-byte aligned
    float* f = reinterpret_cast<float*>(base);
    DhDspShared S;
    S.vol_old = f; S.vol_new = f + 128;
    if (nz > 80u) {
        S.tapsf = f + 256;
        S.stats = reinterpret_cast<uint32_t*>(f + 340); S.clk = S.stats + 2;
        S.bound = f + 352;
    } else {
        S.tapsf = f + 256 + DH_TAP_LEAD;
        S.bound = f + 256 + 112;
        S.stats = reinterpret_cast<uint32_t*>(f + 256 + 124); S.clk = reinterpret_cast<uint32_t*>(f + 384);
    }
    S.var_rb = f + dh_lds_fixed_words(nz);
    S.xf = S.var_rb + DH_RING_WORDS(sps);
    S.mn = S.xf; S.mx = S.xf + DH_SCAN_N;
    S.variance = reinterpret_cast<double*>(S.xf + 2 * DH_SCAN_N);
    S.sum = S.xf + DH_FTILE + 4;
    return S;
}

// Phase markers: with -DDH_ASM_MARKERS (tools/asm_census.py) every phase boundary of the slicer leaves a comment in the assembly; nothing otherwise.
// (Round 2-4's per-phase shader-clock builds used the same places; they went with round 5's prune.)
#if defined(DH_ASM_MARKERS) && DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
// (static census of the phases in the assembly: tools/asm_census.py counts the instructions between these comments)
#define DH_PHASE_MARK_BEGIN() asm volatile("; DH_PHASE begin" ::: "memory")
#define DH_PHASE_MARK(i) asm volatile("; DH_PHASE " #i ::: "memory")
#else
#define DH_PHASE_MARK_BEGIN() ((void) 0)
#define DH_PHASE_MARK(i) ((void) 0)
#endif

// (Wave priorities -- s_setprio around the FIR and in the decoder half -- were worth 2 % in round 2 and nothing since the FIR moved to
// the matrix cores: removed in round 5; the policies and their A/B logs are in git history and profiles/r03_b_ab_logs.txt.)

// per-lane values that must survive a barrier: registers on the GPU, [lane] arrays in the harness
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
#define DH_LANE_ARRAY(type, name, n) type name[n]
#define DH_LA(name, lane) name
#else
#define DH_LANE_ARRAY(type, name, n) type name[DH_WAVE][n]
#define DH_LA(name, lane) name[lane]
#endif

// ---------------------------------------------------------------------------------------------
// (float)((double)acc / gain), the reference's `sum / gain` with a double gain, without paying an
// IEEE double division per sample: q' = fl64(acc * fl64(1/gain)) is within 3 ulp of fl64(acc / gain);
// both round to the same float unless a float rounding boundary (mantissa bits 28..0 = 0x10000000)
// lies within that distance, or the result is a float subnormal.  Only then (p ~ 2e-8) the real
// division runs.  dh_div_gain_exact() is what the fast path must always equal (tested over 2^31
// operands in tests/test_numerics.py).
DH_HD float dh_div_gain_exact(float acc, double gain) { return (float) ((double) acc / gain); }

// fast part: returns (float) q' and sets `suspect` when the exact division has to decide.
// Both tests are "an unsigned word is small":
//   tie    t = (low word of q' << 3) + 0x80000020 <= 64: the low 29 mantissa bits of q' are within 4 ulp(double) of a
//          float midpoint (0x0FFFFFFC .. 0x10000004; the shift drops the three bits above them);
//   range  r = (bits(acc) << 1) - 1 < 2 * bits(2^-118) - 1: acc is non-zero and |acc| < 2^-118.  Results that may be
//          float subnormals round at other bit positions: with 1 < gain < 128 they need |acc| < 2^-118; the quotient of
//          a finite acc cannot overflow.  0, inf and nan are fine: acc * rgain is then acc / gain bit for bit.
// One v_lshl_add_u32 each; a lane's sixteen outputs share ONE decision, so the words are min-reduced (v_min3_u32,
// half an instruction per word) and compared once (dh_fir_finish).
#define DH_DIV_TIE_MAX 64u
#define DH_DIV_RANGE_MAX (2u * 0x04800000u - 1u)             /* bits(2^-118) = (127 - 118) << 23 */
DH_HD float dh_div_gain_words(float acc, double rgain, uint32_t& tie, uint32_t& range) {
    const double q = (double) acc * rgain;
    union { double d; uint64_t u; } b; b.d = q;
    union { float f; uint32_t u; } a; a.f = acc;
    tie = ((uint32_t) b.u << 3) + 0x80000020u;
    range = (a.u << 1) + 0xFFFFFFFFu;
    return (float) q;
}
DH_HD float dh_div_gain_fast(float acc, double rgain, bool& suspect) {
    uint32_t tie, range;
    const float y = dh_div_gain_words(acc, rgain, tie, range);
    suspect = tie <= DH_DIV_TIE_MAX || range < DH_DIV_RANGE_MAX;
    return y;
}

DH_HD float dh_div_gain(float acc, double gain, double rgain) {
    bool suspect;
    const float y = dh_div_gain_fast(acc, rgain, suspect);
    return suspect ? dh_div_gain_exact(acc, gain) : y;
}

// x / d for a float constant d (the samples per symbol: `volume_sum / samplesPerSymbol`, gfsk_demodulator.cpp:83), r =
// RN(1 / d): q0 = RN(x r) is within 1 ulp of x / d, the FMA delivers the residual x - d q0 exactly, and
// RN(q0 + residual * r) is then the correctly rounded quotient (Markstein's theorem; it needs r correctly rounded and
// no underflow / overflow on the way).  Three instructions instead of the eleven of an IEEE division; 0, tiny, huge
// and non-finite x (anything outside 2^-100 <= |x| <= 2^100) take the real division.  tests/test_numerics.py runs all
// 2^32 floats through it for d = 10 (and the other sps values on a sample).
DH_HD float dh_div_const(float x, float d, float r) {
    union { float f; uint32_t u; } a; a.f = x;
    const uint32_t t = (a.u << 1) - 2u * 0x0D800000u;                   // bits(2^-100) = (127 - 100) << 23
    if (t > 2u * (0x71800000u - 0x0D800000u)) return x / d;             // bits(2^100) = (127 + 100) << 23
    const float q0 = x * r;
    const float rem = __builtin_fmaf(-q0, d, x);
    return __builtin_fmaf(rem, r, q0);
}

// the same for a pair of operands (device: v_pk_mul_f32 + two v_pk_fma_f32 for both; one range test on the larger of the two words)
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
typedef float dh_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dh_f2v dh_div_const2(dh_f2v x, float d, float r) {
    const uint32_t ta = (__float_as_uint(x.x) << 1) - 2u * 0x0D800000u, tb = (__float_as_uint(x.y) << 1) - 2u * 0x0D800000u;
    if (__builtin_expect((ta > tb ? ta : tb) > 2u * (0x71800000u - 0x0D800000u), 0)) { dh_f2v q; q.x = dh_div_const(x.x, d, r); q.y = dh_div_const(x.y, d, r); return q; }
    const dh_f2v rr = { r, r }, dd = { -d, -d };
    const dh_f2v q0 = x * rr;
    const dh_f2v rem = __builtin_elementwise_fma(q0, dd, x);
    return __builtin_elementwise_fma(rem, rr, q0);
}
#endif

// loop-invariant wave-uniform values that should live in VGPRs rather than compete for the 102 SGPRs
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
#define DH_TO_VGPR(x) asm volatile("" : "+v"(x))
#else
#define DH_TO_VGPR(x) ((void) 0)
#endif
// compiler-only memory fence: values cached from memory before it must be re-read after it
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
#define DH_COMPILER_FENCE() asm volatile("" ::: "memory")
#else
#define DH_COMPILER_FENCE() ((void) 0)
#endif

// ---------------------------------------------------------------------------------------------
// FIR pass: lane computes outputs [base, base+16) of the tile from the padded LDS window.
//   x[e] = xs[DH_XPAD(e)] holds input sample (tile_start - NZ + e);  out[j] = sum_i c[i]*x[base+j+i]
//
// Register layout chosen for packed fp32 (v_pk_mul_f32 / v_pk_add_f32 work on aligned VGPR pairs):
// the 16 outputs are two groups of 8, and output j of group A shares a register pair with output j
// of group B.  The sliding window then holds pairs (x[t], x[t+8]); sliding by one tap renames
// pairs instead of re-aligning them, and one ds_read2_b32 fills a pair.  Straight-line after
// unrolling: 8 window pairs + 8 accumulator pairs + the taps (one VGPR each).
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
typedef float dh_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dh_f2 dh_f2_make(float a, float b) { dh_f2 v; v.x = a; v.y = b; return v; }
template <bool FAST> __device__ __forceinline__ dh_f2 dh_f2_mac(float c, dh_f2 w, dh_f2 acc) {
    if (FAST) { const dh_f2 cc = dh_f2_make(c, c); return __builtin_elementwise_fma(cc, w, acc); }   // v_pk_fma_f32
    const dh_f2 p = c * w;                                 // v_pk_mul_f32, rounded
    return acc + p;                                        // v_pk_add_f32, rounded (-ffp-contract=off)
}
// The same with the tap as ONE HALF of a 64-bit scalar register pair (two taps per pair, chosen with op_sel): the
// packed multiply takes the pair as its scalar operand, so the taps occupy 41 / 81 SGPRs instead of as many VGPRs.
// Left to the compiler a scalar tap becomes a (c, c) pair -- two SGPRs per tap, which spills SGPRs into VGPR lanes
// in the slicer kernel -- hence the explicit instruction.
template <bool FAST, int HALF> __device__ __forceinline__ dh_f2 dh_f2_mac_pair(dh_f2 cc, dh_f2 w, dh_f2 acc) {
    dh_f2 p;
    if (FAST) {
        if (HALF == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(p) : "s"(cc), "v"(w), "v"(acc));
        else           asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(p) : "s"(cc), "v"(w), "v"(acc));
        return p;
    }
    if (HALF == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(p) : "s"(cc), "v"(w));
    else           asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(p) : "s"(cc), "v"(w));
    return acc + p;                                        // v_pk_add_f32, rounded (-ffp-contract=off)
}
#else
struct dh_f2 { float x, y; };
inline dh_f2 dh_f2_make(float a, float b) { dh_f2 v; v.x = a; v.y = b; return v; }
template <bool FAST> inline dh_f2 dh_f2_mac(float c, dh_f2 w, dh_f2 acc) {
    if (FAST) return dh_f2_make(__builtin_fmaf(c, w.x, acc.x), __builtin_fmaf(c, w.y, acc.y));
    const float px = c * w.x, py = c * w.y;
    return dh_f2_make(acc.x + px, acc.y + py);
}
#endif
// element-wise pair arithmetic for both builds (device: v_pk_add_f32 / v_pk_fma_f32)
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ dh_f2 dh_f2_add(dh_f2 a, dh_f2 b) { return a + b; }
__device__ __forceinline__ dh_f2 dh_f2_sub(dh_f2 a, dh_f2 b) { return a - b; }
__device__ __forceinline__ dh_f2 dh_f2_fma(dh_f2 a, dh_f2 b, dh_f2 c) { return __builtin_elementwise_fma(a, b, c); }
#else
inline dh_f2 dh_f2_add(dh_f2 a, dh_f2 b) { return dh_f2_make(a.x + b.x, a.y + b.y); }
inline dh_f2 dh_f2_sub(dh_f2 a, dh_f2 b) { return dh_f2_make(a.x - b.x, a.y - b.y); }
inline dh_f2 dh_f2_fma(dh_f2 a, dh_f2 b, dh_f2 c) { return dh_f2_make(__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)); }
#endif
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ dh_f2 dh_f2_scale(dh_f2 a, float s) { const dh_f2 ss = { s, s }; return a * ss; }
// min(|a|, |b|, |c|) in one instruction; a NaN operand is skipped (v_min3_f32), so callers that must notice a NaN distance make
// sure ALL THREE are NaN then (they are: every distance of a symbol hangs on the same average and extremes) -- see P5
__device__ __forceinline__ float dh_min3_abs(float a, float b, float c) { float r; asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#else
inline dh_f2 dh_f2_scale(dh_f2 a, float s) { return dh_f2_make(a.x * s, a.y * s); }
inline float dh_min3_abs(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(__builtin_fabsf(a), __builtin_fabsf(b)), __builtin_fabsf(c)); }
#endif
#if !(DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__))
inline dh_f2 dh_div_const2(dh_f2 x, float d, float r) { return dh_f2_make(dh_div_const(x.x, d, r), dh_div_const(x.y, d, r)); }
#endif
#define DH_FIR_H (DH_FIR_L / 2)
#define DH_XLOFF(e) ((e) + ((e) >> 4))                 // dword offset of window element e from the lane's base

// (float)((double)acc / gain) for the 16 accumulators of a lane, see dh_div_gain_fast
template <bool FAST>
DH_HD void dh_fir_finish(const float* acc, double gain, double rgain, float inv_gain, float* out16, bool* nonfinite = nullptr) {
    if (FAST) {
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
        // one v_mul_f32 per output, from where the accumulator half lies to where the output is wanted: left to itself the
        // compiler packs the multiplies and pays for it with ~29 register moves (halves gathered into pairs before, results
        // copied into the outputs' registers after)
#pragma unroll
        for (int j = 0; j < DH_FIR_L; j++) asm("v_mul_f32 %0, %1, %2" : "=v"(out16[j]) : "s"(inv_gain), "v"(acc[j]));
#else
#pragma unroll
        for (int j = 0; j < DH_FIR_L; j++) out16[j] = acc[j] * inv_gain;
#endif
        if (nonfinite) {                                  // NaN or infinity in any accumulator: 0 * acc is then NaN
            float t = 0.0f;
#pragma unroll
            for (int j = 0; j < DH_FIR_L; j++) t = __builtin_fmaf(acc[j], 0.0f, t);
            *nonfinite = !(t == 0.0f);
        }
    } else {
        uint32_t tmin = 0xFFFFFFFFu, rmin = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < DH_FIR_L; j++) {
            uint32_t t, r;
            out16[j] = dh_div_gain_words(acc[j], rgain, t, r);
            tmin = dh_min<uint32_t>(tmin, t); rmin = dh_min<uint32_t>(rmin, r);
        }
        const bool any = tmin <= DH_DIV_TIE_MAX || rmin < DH_DIV_RANGE_MAX;
        if (any) {                                        // ~2e-8 per sample: one copy of the IEEE division, not sixteen
#pragma unroll 1
            for (int j = 0; j < DH_FIR_L; j++) {
                bool s;
                (void) dh_div_gain_fast(acc[j], rgain, s);
                if (s) out16[j] = dh_div_gain_exact(acc[j], gain);
            }
        }
    }
}

#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
// Window pairs come from LDS through inline asm: every sample is needed twice, 8 taps apart -- first as the high
// half of a pair, then as the low half of another.  Given ordinary loads the compiler keeps the first copy, pairs
// up adjacent addresses instead, and shuffles halves with ~2 v_mov per tap on the VALU, which is the bottleneck
// of this kernel; the LDS pipe is idle.  One ds_read2_b32 per tap delivers (x[t], x[t+8]) straight into an aligned
// register pair.  The loads run one batch of DH_FIR_G taps ahead of the arithmetic; the compiler does not track
// them, so each batch passes through an explicit `s_waitcnt lgkmcnt(0)` (robust against any other LDS / scalar
// memory operation the compiler may have in flight) before its first use.
#define DH_FIR_G 4
template <int O0, int O1> __device__ __forceinline__ dh_f2 dh_lds_read2(uint32_t addr) {
    static_assert(O0 < 256 && O1 < 256, "ds_read2_b32 offsets are 8-bit dword counts");
    dh_f2 v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1) : "memory");
    return v;
}
// (eight bytes from a four-byte aligned address: the queue's LDS alignment mode is "unaligned" on this target)
template <int OFF_BYTES> __device__ __forceinline__ dh_f2 dh_lds_read_b64(uint32_t addr) {
    dh_f2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF_BYTES) : "memory");
    return v;
}
typedef float dh_v4f __attribute__((ext_vector_type(4)));
template <int OFF_BYTES> __device__ __forceinline__ dh_v4f dh_lds_read_b128(uint32_t addr) {
    dh_v4f v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF_BYTES) : "memory");
    return v;
}
// pair i of the slide sequence: (x[8 + i], x[16 + i]), consumed when tap i has been accumulated
template <int NZ, int I> __device__ __forceinline__ void dh_fir_issue_one(uint32_t addr, dh_f2& d) {
    if constexpr (I < NZ) d = dh_lds_read2<DH_XLOFF(DH_FIR_H + I), DH_XLOFF(2 * DH_FIR_H + I)>(addr);
    else d = dh_f2_make(0.0f, 0.0f);
}
template <int NZ, int B> __device__ __forceinline__ void dh_fir_issue(uint32_t addr, dh_f2 (&d)[DH_FIR_G]) {
    dh_fir_issue_one<NZ, B * DH_FIR_G + 0>(addr, d[0]); dh_fir_issue_one<NZ, B * DH_FIR_G + 1>(addr, d[1]);
    dh_fir_issue_one<NZ, B * DH_FIR_G + 2>(addr, d[2]); dh_fir_issue_one<NZ, B * DH_FIR_G + 3>(addr, d[3]);
}
// The same batch, issued AND waited for inside one asm statement: nothing of the compiler's (a spill of a destination
// pair, say) can come between a load and its wait.  For the FIR bodies that are rare paths of a kernel at its register
// limit -- the reference-order FIR of the error-bounded chain kernels, which the compiler is told is cold and which it
// therefore spills around freely (tests/test_isa_contract.py caught a destination pair stored to scratch ahead of its wait).
template <int NZ, int B> __device__ __forceinline__ void dh_fir_issue_wait(uint32_t addr, dh_f2 (&d)[DH_FIR_G]) {
    if constexpr (B * DH_FIR_G < NZ) {
        static_assert(NZ % DH_FIR_G == 0, "a batch is all taps or all padding");
        asm volatile("ds_read2_b32 %0, %4 offset0:%5 offset1:%6\n\tds_read2_b32 %1, %4 offset0:%7 offset1:%8\n\t"
                     "ds_read2_b32 %2, %4 offset0:%9 offset1:%10\n\tds_read2_b32 %3, %4 offset0:%11 offset1:%12\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(addr),
                       "n"(DH_XLOFF(DH_FIR_H + B * DH_FIR_G + 0)), "n"(DH_XLOFF(2 * DH_FIR_H + B * DH_FIR_G + 0)),
                       "n"(DH_XLOFF(DH_FIR_H + B * DH_FIR_G + 1)), "n"(DH_XLOFF(2 * DH_FIR_H + B * DH_FIR_G + 1)),
                       "n"(DH_XLOFF(DH_FIR_H + B * DH_FIR_G + 2)), "n"(DH_XLOFF(2 * DH_FIR_H + B * DH_FIR_G + 2)),
                       "n"(DH_XLOFF(DH_FIR_H + B * DH_FIR_G + 3)), "n"(DH_XLOFF(2 * DH_FIR_H + B * DH_FIR_G + 3)) : "memory");
    } else {
        d[0] = dh_f2_make(0.0f, 0.0f); d[1] = d[0]; d[2] = d[0]; d[3] = d[0];
    }
}
__device__ __forceinline__ void dh_fir_arrived(dh_f2 (&d)[DH_FIR_G]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) :: "memory");
}
// SG: the taps are read as 64-bit scalar pairs from the kernel arguments (dh_f2_mac_pair) instead of from a register
// array.  A packed multiply with a scalar operand issues a little slower (slicer kernels: +5 %), so this only pays
// where the 41 / 81 tap registers cost a wavefront per SIMD -- the stand-alone RRC kernel (168 -> 105 VGPRs).
// ZINIT = false: the accumulators start as the first product instead of 0 + product -- the same float except that a
// product of -0 stays -0 where the reference's `sum = 0; sum += c * x` gives +0.  The sign of a zero cannot reach a
// dibit (the slicer only adds, subtracts and compares filtered samples), so the slicer kernels drop the eight packed
// adds and the zeroing; the materialised RRC output (k_rrc_tile) keeps the reference's bits.
template <int NZ, bool FAST, int B, bool SG = false, bool ZINIT = true, bool SAFE = false> struct DhFirBatch {
    // tap I = B * DH_FIR_G + G: accumulate it on the eight window pairs, then slide the window by one sample
    template <int G> static __device__ __forceinline__ void tap(const float* taps, dh_f2 (&accp)[DH_FIR_H], dh_f2 (&w)[DH_FIR_H], dh_f2 (&cur)[DH_FIR_G]) {
        constexpr int I = B * DH_FIR_G + G;
        if constexpr (I <= NZ) {
            constexpr int TI = I <= NZ / 2 ? I : NZ - I;
            if constexpr (SG) {
                const dh_f2 cc = reinterpret_cast<const dh_f2*>(taps)[TI >> 1];     // wave-uniform 64-bit load
#pragma unroll
                for (int j = 0; j < DH_FIR_H; j++) accp[j] = dh_f2_mac_pair<FAST, TI & 1>(cc, w[j], accp[j]);
            } else if constexpr (I == 0 && !ZINIT) {
                const float c = taps[TI];
#pragma unroll
                for (int j = 0; j < DH_FIR_H; j++) accp[j] = c * w[j];              // v_pk_mul_f32
            } else {
                const float c = taps[TI];
#pragma unroll
                for (int j = 0; j < DH_FIR_H; j++) accp[j] = dh_f2_mac<FAST>(c, w[j], accp[j]);
            }
            if constexpr (I < NZ) {
#pragma unroll
                for (int j = 0; j < DH_FIR_H - 1; j++) w[j] = w[j + 1];
                w[DH_FIR_H - 1] = cur[G];
            }
        }
    }
    static __device__ __forceinline__ void run(const float* taps, uint32_t addr, dh_f2 (&accp)[DH_FIR_H], dh_f2 (&w)[DH_FIR_H], dh_f2 (&cur)[DH_FIR_G]) {
        if constexpr (B * DH_FIR_G <= NZ) {
            dh_f2 nxt[DH_FIR_G];
            if constexpr (SAFE) dh_fir_issue_wait<NZ, B + 1>(addr, nxt); else dh_fir_issue<NZ, B + 1>(addr, nxt);
            tap<0>(taps, accp, w, cur); tap<1>(taps, accp, w, cur); tap<2>(taps, accp, w, cur); tap<3>(taps, accp, w, cur);
            static_assert(DH_FIR_G == 4, "four taps per batch");
            if constexpr (!SAFE && (B + 1) * DH_FIR_G < NZ) dh_fir_arrived(nxt);
            DhFirBatch<NZ, FAST, B + 1, SG, ZINIT, SAFE>::run(taps, addr, accp, w, nxt);
        }
    }
};

template <int NZ, bool FAST, bool SG = false, bool SAFE = false>
__device__ __forceinline__ void dh_fir_lane(const float* taps, double gain, double rgain, float inv_gain, const float* xs_all, int lane, float* out16, bool* nonfinite = nullptr) {
    constexpr bool ZINIT = SG;                             // the stand-alone RRC kernel (scalar taps) materialises its output
    // element e of this lane's window sits at DH_XPAD(16*lane + e) = 17*lane + e + (e >> 4): static offsets from one base
    const uint32_t addr = (uint32_t) (uintptr_t) (const __attribute__((address_space(3))) float*) (xs_all + (DH_FIR_L + 1) * lane);
    dh_f2 accp[DH_FIR_H], w[DH_FIR_H], cur[DH_FIR_G];
    if constexpr (SAFE) {
        // (the first eight window pairs as two batches of the slide sequence shifted by -8: pair i = (x[i], x[8 + i]))
        asm volatile("ds_read2_b32 %0, %8 offset0:%9 offset1:%10\n\tds_read2_b32 %1, %8 offset0:%11 offset1:%12\n\t"
                     "ds_read2_b32 %2, %8 offset0:%13 offset1:%14\n\tds_read2_b32 %3, %8 offset0:%15 offset1:%16\n\t"
                     "ds_read2_b32 %4, %8 offset0:%17 offset1:%18\n\tds_read2_b32 %5, %8 offset0:%19 offset1:%20\n\t"
                     "ds_read2_b32 %6, %8 offset0:%21 offset1:%22\n\tds_read2_b32 %7, %8 offset0:%23 offset1:%24\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7]) : "v"(addr),
                       "n"(DH_XLOFF(0)), "n"(DH_XLOFF(8)), "n"(DH_XLOFF(1)), "n"(DH_XLOFF(9)), "n"(DH_XLOFF(2)), "n"(DH_XLOFF(10)), "n"(DH_XLOFF(3)), "n"(DH_XLOFF(11)),
                       "n"(DH_XLOFF(4)), "n"(DH_XLOFF(12)), "n"(DH_XLOFF(5)), "n"(DH_XLOFF(13)), "n"(DH_XLOFF(6)), "n"(DH_XLOFF(14)), "n"(DH_XLOFF(7)), "n"(DH_XLOFF(15)) : "memory");
        dh_fir_issue_wait<NZ, 0>(addr, cur);
    } else {
    w[0] = dh_lds_read2<DH_XLOFF(0), DH_XLOFF(8)>(addr);  w[1] = dh_lds_read2<DH_XLOFF(1), DH_XLOFF(9)>(addr);
    w[2] = dh_lds_read2<DH_XLOFF(2), DH_XLOFF(10)>(addr); w[3] = dh_lds_read2<DH_XLOFF(3), DH_XLOFF(11)>(addr);
    w[4] = dh_lds_read2<DH_XLOFF(4), DH_XLOFF(12)>(addr); w[5] = dh_lds_read2<DH_XLOFF(5), DH_XLOFF(13)>(addr);
    w[6] = dh_lds_read2<DH_XLOFF(6), DH_XLOFF(14)>(addr); w[7] = dh_lds_read2<DH_XLOFF(7), DH_XLOFF(15)>(addr);
    dh_fir_issue<NZ, 0>(addr, cur);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) :: "memory");
    dh_fir_arrived(cur);
    }
    if constexpr (ZINIT) {
#pragma unroll
        for (int j = 0; j < DH_FIR_H; j++) accp[j] = dh_f2_make(0.0f, 0.0f);
    }
    DhFirBatch<NZ, FAST, 0, SG, ZINIT, SAFE>::run(taps, addr, accp, w, cur);
    float acc[DH_FIR_L];
#pragma unroll
    for (int j = 0; j < DH_FIR_H; j++) { acc[j] = accp[j].x; acc[j + DH_FIR_H] = accp[j].y; }
    if (FAST && nonfinite) {                               // 0 * acc is NaN for a NaN or infinite accumulator: eight packed FMAs
        dh_f2 t = dh_f2_make(0.0f, 0.0f);
        const dh_f2 zero = dh_f2_make(0.0f, 0.0f);
#pragma unroll
        for (int j = 0; j < DH_FIR_H; j++) t = __builtin_elementwise_fma(accp[j], zero, t);
        *nonfinite = !(t.x + t.y == 0.0f);
        dh_fir_finish<FAST>(acc, gain, rgain, inv_gain, out16, nullptr);
    } else dh_fir_finish<FAST>(acc, gain, rgain, inv_gain, out16, nonfinite);
}
#else
template <int NZ, bool FAST>
inline void dh_fir_lane(const float* taps, double gain, double rgain, float inv_gain, const float* xs_all, int lane, float* out16, bool* nonfinite = nullptr) {
    const float* xs = xs_all + (DH_FIR_L + 1) * lane;
#define DH_XL(e) xs[DH_XLOFF(e)]
    dh_f2 accp[DH_FIR_H], w[DH_FIR_H];
    for (int j = 0; j < DH_FIR_H; j++) {
        accp[j] = dh_f2_make(0.0f, 0.0f);
        w[j] = dh_f2_make(DH_XL(j), DH_XL(j + DH_FIR_H));
    }
    for (int i = 0; i <= NZ; i++) {
        const float c = taps[i <= NZ / 2 ? i : NZ - i];
        for (int j = 0; j < DH_FIR_H; j++) accp[j] = dh_f2_mac<FAST>(c, w[j], accp[j]);
        if (i < NZ) {
            for (int j = 0; j < DH_FIR_H - 1; j++) w[j] = w[j + 1];
            w[DH_FIR_H - 1] = dh_f2_make(DH_XL(DH_FIR_H + i), DH_XL(2 * DH_FIR_H + i));
        }
    }
    float acc[DH_FIR_L];
    for (int j = 0; j < DH_FIR_H; j++) { acc[j] = accp[j].x; acc[j + DH_FIR_H] = accp[j].y; }
    dh_fir_finish<FAST>(acc, gain, rgain, inv_gain, out16, nonfinite);
}
#endif

// four consecutive floats from a 4-byte-aligned address (global_load_dwordx4: gfx950 allows dword alignment)
struct __attribute__((aligned(4))) dh_f4 { float x, y, z, w; };
struct alignas(16) dh_f4a { float x, y, z, w; };     // 16-byte aligned: one ds_read_b128
DH_HD dh_f4 dh_load4_unaligned(const float* p) { dh_f4 v; __builtin_memcpy(&v, p, sizeof(v)); return v; }
// the same for the one-touch input stream: DH_NT_INPUT=1 marks the loads non-temporal (A/B: see DESIGN.md section 5)
DH_HD dh_f4 dh_load4_stream(const float* p) { return dh_load4_unaligned(p); }
DH_HD void dh_store4(float* q, const dh_f4& v) { q[0] = v.x; q[1] = v.y; q[2] = v.z; q[3] = v.w; }
DH_HD void dh_store4_unaligned(float* q, const dh_f4& v) { __builtin_memcpy(q, &v, sizeof(v)); }   // global_store_dwordx4

// ---------------------------------------------------------------------------------------------
// The fused FIR on the matrix pipe (error-bounded and FAST wide-filter kernels): v_mfma_f32_16x16x4_f32 is bit for bit a
// k-ordered chain of f32 FMAs, at the same FLOP rate as v_pk_fma_f32 -- but it issues beside the VALU, which the rest
// of the slicer keeps busy, and takes one register per operand instead of a ds_read2 per packed FMA.
//
// Toeplitz form.  The 1024 outputs of a pass are 64 blocks of 16: y[16 b + n] = sum_p x[16 b + p] c[p - n], p = 0..NZ+15,
// with c = 0 outside 0..NZ.  One MFMA tile takes 16 blocks as its rows (A operand: the raw samples) and the 16 samples of
// a block as its columns (B operand: the taps, the same for every block and tile); K runs over p in (NZ + 16) / 4 slices
// of 4.  Multiplying by an exact zero changes nothing, so each output is the (NZ + 1)-term fused chain in tap order --
// the value the VALU form (dh_fir_lane<NZ, true>) produces, and the one the bound of "Error-bounded FIR" is proven for
// (a non-finite sample under a zero tap turns into a NaN, which is exactly what sends the run to the reference's
// arithmetic anyway).
//   A: lane (m = l & 15, q = l >> 4) holds x[16 b(m) + 4 s + q];   B: lane (n = l & 15, q = l >> 4) holds c[4 s + q - n]
//   D: lane (n = l & 15, g = l >> 4), register r holds y[16 b(4 g + r) + n]
// Blocks of tile T: b(m) = 2 m + (T & 1) + 32 (T >> 1).  In the padded window (one word per 16) block b starts at word
// 17 b, so the sixteen rows of a tile are 34 words apart: with the q = 0 / 1 (2 / 3) lanes of a ds_read_b32's lane group
// one word apart that is 32 different banks.  Tiles T and T + 1 interleave their blocks, so slice s of the odd tile IS
// slice s + 4 of the even one: 28 window reads serve the 48 MFMAs of a tile pair.  Results go back unpadded (sample o at
// word o): sixteen ds_write_b32 at compile-time offsets from one per-lane base (2-way bank conflicts, which a 4-byte
// store hides behind its own issue time).
#define DH_MFMA_FIR 1
#define DH_MF_OUT(T, r, g, n) (128u * (uint32_t) (g) + (uint32_t) (n) + 32u * (uint32_t) (r) + 16u * ((uint32_t) (T) & 1u) + 512u * ((uint32_t) (T) >> 1))
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
typedef float dh_f32x4 __attribute__((ext_vector_type(4)));
// acc[4 T + r] of lane (n, g) = sum over the chain for output DH_MF_OUT(T, r, g, n); `tapsf` points at tap 0 of the
// zero-framed table, `xs` at the padded window
template <int NZ>
__device__ __forceinline__ void dh_fir_mfma(const float* tapsf, const float* xs, int lane, float* acc16) {
    static_assert((NZ + 16) % 4 == 0 && NZ <= 80, "K = NZ + 16 in slices of four; the zero-framed tap table exists for the wide filter");
    constexpr int NS = (NZ + 16) / 4;
    const int m = lane & 15, q = lane >> 4;
    const float* xb = xs + 34 * m + q;
    const float* tb = tapsf + (q - m);
    float tap[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) tap[s] = tb[4 * s];
    dh_f32x4 a0 = { 0.0f, 0.0f, 0.0f, 0.0f }, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
    for (int s = 0; s < NS + 4; s++) {
        const int off = 4 * s + (s >> 2);
        const float x01 = xb[off], x23 = xb[544 + off];
        if (s < NS) { a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x01, tap[s], a0, 0, 0, 0); a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x23, tap[s], a2, 0, 0, 0); }
        if (s >= 4) { a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x01, tap[s - 4], a1, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x23, tap[s - 4], a3, 0, 0, 0); }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { acc16[r] = a0[r]; acc16[4 + r] = a1[r]; acc16[8 + r] = a2[r]; acc16[12 + r] = a3[r]; }
}
#else
template <int NZ>
inline void dh_fir_mfma(const float* tapsf, const float* xs, int lane, float* acc16) {
    const int n = lane & 15, g = lane >> 4;
    for (int T = 0; T < 4; T++) for (int r = 0; r < 4; r++) {
        const uint32_t o = DH_MF_OUT(T, r, g, n), b16 = o - (uint32_t) n;
        float acc = 0.0f;
        for (int p = 0; p < NZ + 16; p++) acc = __builtin_fmaf(xs[DH_XPAD(b16 + (uint32_t) p)], tapsf[p - n], acc);
        acc16[4 * T + r] = acc;
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// The fused FIR as a SPLIT-f16 product on the matrix cores (DH_FIR_F16; the error-bounded wide-filter kernels).
//
// f32 arithmetic -- v_pk_fma_f32 and v_mfma_f32_16x16x4_f32 alike -- runs at 64 flop / clock / SIMD, and an f32 MFMA keeps
// the vector ALU as busy as the packed FMAs it replaces (tools/microbench/mfma_valu_overlap.hip: MFMA wave + VALU wave on one
// SIMD take the SUM of their times): 81 taps x 1024 outputs are 2 600 SIMD cycles per run whichever instruction does them.
// v_mfma_f32_16x16x32_f16 does 16 times the work per cycle.  The error-bounded scheme does not need the reference's floats,
// only values with a KNOWN distance from them, so the filter runs there:
//   samples   x_s = x 2^-k  (k from the run's max |x|, which staging computes anyway: max |x_s| in [0.5, 1))
//             h1 = f16(x_s), h2 = f16((x_s - h1) 2^11):   x_s = h1 + 2^-11 h2 + eps,  |eps| <= 2^-24   (both conversions round to
//             nearest; x_s - h1 is exact in f32; f16 subnormals are honoured by the MFMA -- tools/microbench/mfma_f16_numerics.hip)
//   taps      c = g1 + 2^-11 g2 + delta, g1 = f16(c), g2 = f16((c - g1) 2^11)   (host, once per engine; delta known exactly)
//   products  M = sum g1 h1  (3 MFMAs of K = 32 per tile of 256 outputs),  R = sum (g2 h1 + g1 h2)  (6 MFMAs),
//             y = (M + 2^-11 R) / (gain 2^-k)  as  fma(R, k2, M k1);   the g2 h2 products (2^-22) are dropped
// f16 x f16 products are exact in f32; what the hardware does to their sum is not documented, so the bound rests on a
// stated assumption, checked on the device by tests/test_numerics.py (known-answer probes + 10^6 random dot products):
//   (H1) inside one MFMA every addend (32 products and C) enters the sum with an alignment error below 2^-24 mu, and each
//        of its four passes rounds with a relative error <= 2^-23, where mu bounds every addend and partial sum:
//        |D - (C + sum a b)| <= 41 u mu,  u = 2^-24.  (Observed on gfx950: products are truncated to 2^-24 of the largest
//        product of their group of eight, groups are added in order with a final round to nearest; worst case over 10^6
//        random sums 3.5 u (|C| + sum |a b|) -- a tenth of what (H1) allows.)
// dh_f16_error_coefficient() adds it all up (reference chain gamma_82 + its division; 123 u for the three chained MFMAs of
// M; the split residuals; the dropped products; the two roundings of the combine) and keeps the 1.44 of head room for the
// slicer's own roundings that the f32 bound has.  Toeplitz layout as in dh_fir_mfma, natural block order:
//   A: lane (m = l & 15, q = l >> 4) holds h[16 (16 T + b(m)) + 32 s + 8 q + j], j = 0..7 -- ONE ds_read_b128 from the plain f16
//      window (32 bytes per row, 16 per q: conflict-free in every lane group of a b128 read); b(m) = DH_F16_ROW_BLOCK(m), a
//      permutation of the tile's sixteen blocks inside each half of the rows (see DH_F16_OUT)
//   B: lane (n = l & 15, q = l >> 4) holds g[32 s + 8 q + j - n]: per-lane fragments precomputed on the host (6 x 1 KiB,
//      dh_f16_tap_fragments), fetched with the samples
//   D: lane (n, g), register r holds y[256 T + 16 b(4 g + r) + n] = y[DH_F16_OUT(T, r, g, n)]
#define DH_FIR_F16 1
#define DH_PLAN_FAST 1                       // sps-10 kernels: the run planning of a whole-block run in three compares
#define DH_EXACT_BATCH 2                     // exact evaluations of the 81-tap kernels: this many products at a time, their LDS reads in flight together (the 161-tap ones: 8)
#define DH_PF_REG 1                          // split-f16 kernels: the next window is fetched into registers behind P3 and split into halves in P7
#define DH_F16_EDGE_WINDOWS 1                // split-f16 kernels: the first / last windows of a push take the split-f16 FIR too (0: the reference-order FIR, as before)
// K = taps + 15 rounded up to whole MFMAs of 32: three for the wide filter (96), six for the narrow one (192; taps beyond the
// response are zeros in the fragments, and the halves beyond the window are stored as zeros)
#define DH_F16_KSTEPS_OF(nz) (((nz) + 16 + 31) / 32)
#define DH_F16_MAX_KSTEPS DH_F16_KSTEPS_OF(DH_MAX_NZ)
#define DH_F16_HALVES_OF(nz) (1008 + 32 * DH_F16_KSTEPS_OF(nz))     // halves per array: the last block (63) reads up to here
#define DH_F16_H2_OFFSET_OF(nz) (DH_F16_HALVES_OF(nz) / 2)           // word offset of the h2 array inside the window block
// Row m = 4 g + r of a tile (the block of sixteen outputs an A row produces) is block 2 r + (g & 1) + 8 (g >> 1) of the tile, not block m:
// register r of the lanes of ONE store pass (32 lanes: g = 0, 1 or g = 2, 3) then covers 32 consecutive words -- in natural order the
// two groups of a pass were 64 words apart and every result store ran into a two-way bank conflict.
#define DH_F16_ROW_BLOCK(m) (2u * ((uint32_t) (m) & 3u) + (((uint32_t) (m) >> 2) & 1u) + 8u * ((uint32_t) (m) >> 3))
#define DH_F16_OUT(T, r, g, n) (256u * (uint32_t) (T) + 128u * ((uint32_t) (g) >> 1) + 32u * (uint32_t) (r) + 16u * ((uint32_t) (g) & 1u) + (uint32_t) (n))

// IEEE binary16 <-> binary32 in integer arithmetic (host and harness; the device converts in hardware): round to nearest even
DH_HD uint16_t dh_f16_bits(float f) {
    union { float f; uint32_t u; } v; v.f = f;
    const uint32_t sign = (v.u >> 16) & 0x8000u, a = v.u & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return (uint16_t) (sign | (a > 0x7F800000u ? 0x7E00u : 0x7C00u));
    if (a >= 0x477FF000u) return (uint16_t) (sign | 0x7C00u);                         // >= 65520 rounds to infinity
    if (a < 0x33000001u) return (uint16_t) sign;                                      // <= 2^-25: zero (the tie goes to even = 0)
    const int e = (int) (a >> 23) - 127;
    uint32_t man = (a & 0x7FFFFFu) | 0x800000u, shift = e < -14 ? (uint32_t) (13 + (-14 - e)) : 13u;
    uint32_t q = man >> shift; const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1u);
    if (rem > half || (rem == half && (q & 1u))) q++;
    const uint32_t bits = e < -14 ? q : ((uint32_t) (e + 15 - 1) << 10) + q;           // (the implicit bit of q carries into the exponent)
    return (uint16_t) (sign | bits);
}
DH_HD float dh_f16_value(uint16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 1023u;
    union { float f; uint32_t u; } v;
    if (e == 31u) v.u = sign | 0x7F800000u | (m << 13);
    else if (e == 0u) { v.f = (float) m * 5.9604644775390625e-08f; v.u |= sign; }     // m 2^-24
    else v.u = sign | ((e + 112u) << 23) | (m << 13);
    return v.f;
}

// the split of a tap and the per-lane B operands (host, once per engine): frag[f][lane][d], f = 0..2 the g1 slices, 3..5 the g2 slices
struct DhF16Taps {
    uint32_t frag[2 * DH_F16_MAX_KSTEPS][DH_WAVE][4];        // the first 2 * DH_F16_KSTEPS_OF(nz) are used
    uint32_t ksteps;
    double l1, l1_g1, l1_g2, sum_delta;
};
inline void dh_f16_tap_fragments(const float* taps_half, uint32_t nz, DhF16Taps& F) {
    uint16_t g1[DH_MAX_NZ + 1], g2[DH_MAX_NZ + 1];
    F.l1 = F.l1_g1 = F.l1_g2 = F.sum_delta = 0.0;
    for (uint32_t i = 0; i <= nz; i++) {
        const float c = taps_half[i <= nz / 2 ? i : nz - i];
        g1[i] = dh_f16_bits(c);
        const double r = (double) c - (double) dh_f16_value(g1[i]);
        g2[i] = dh_f16_bits((float) (r * 2048.0));                                     // (r has at most 13 significant bits: exact)
        const double v2 = (double) dh_f16_value(g2[i]);
        const double delta = r - v2 / 2048.0;
        F.l1 += c < 0 ? -(double) c : (double) c; F.l1_g1 += __builtin_fabs((double) dh_f16_value(g1[i]));
        F.l1_g2 += __builtin_fabs(v2); F.sum_delta += __builtin_fabs(delta);
    }
    const int KS = (int) DH_F16_KSTEPS_OF(nz);
    F.ksteps = (uint32_t) KS;
    for (int f = 0; f < 2 * KS; f++) for (int lane = 0; lane < DH_WAVE; lane++) for (int d = 0; d < 4; d++) {
        const int n = lane & 15, q = lane >> 4, s = f % KS;
        uint32_t w = 0;
        for (int hh = 0; hh < 2; hh++) {
            const int t = 32 * s + 8 * q + 2 * d + hh - n;
            const uint16_t v = (t >= 0 && t <= (int) nz) ? (f < KS ? g1[t] : g2[t]) : 0;
            w |= (uint32_t) v << (16 * hh);
        }
        F.frag[f][lane][d] = w;
    }
}
// error radius of a filtered sample per unit of max |x| for this path (see above; everything in units of u / gain):
//   reference: (taps + 1) (gamma, + second order) L1 + 1.0001 L1 (its division and rounding): 82 for the wide filter
//   main sum:  41 KS L1(g1) (1 + 2^-11) ((H1), KS chained MFMAs -- 123 for the wide filter; |h1| <= max |x_s| (1 + 2^-11))
//   second sum: 2^-11 82 KS (L1(g2) + L1(g1) / 2)
//   residuals (absolute in the scaled domain, where max |x_s| >= 1/2: twice as much per unit of max |x|):
//              2 (2^-22 L1(g2) / 2 + sum |delta| (1 + 2^-12) + u L1) / u
//   combine:   3.1 L1 (fl32(1 / gain), the multiply, the fma)
// times 1.44 for the slicer's own roundings, as in dh_fir_error_coefficient.
inline float dh_f16_error_coefficient(const DhF16Taps& F, uint32_t nz, double gain) {
    const double u = 5.9604644775390625e-08;
    const double ks = (double) F.ksteps, nref = 2.0 + (double) nz;                     // gamma_(taps + 1) for the reference's chain
    const double kl1 = (nref + 0.01) * F.l1 + 1.0001 * F.l1 + 41.0 * ks * F.l1_g1 * (1.0 + 1.0 / 2048.0) + 82.0 * ks / 2048.0 * (F.l1_g2 + 0.5 * F.l1_g1)
                     + 2.0 * (0.5 * F.l1_g2 / 4194304.0 + F.sum_delta * (1.0 + 1.0 / 4096.0) + u * F.l1) / u + 3.1 * F.l1;
    const double coef = 1.44 * kl1 * u / (gain < 0 ? -gain : gain) * 1.0001;
    float f = (float) coef;
    if ((double) f < coef) { union { float f; uint32_t u; } b; b.f = f; b.u++; f = b.f; }
    return f;
}

#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
typedef _Float16 dh_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 dh_h4 __attribute__((ext_vector_type(4)));
typedef uint32_t dh_u4 __attribute__((ext_vector_type(4)));
// four samples -> four halves of h1 and of h2 (scale = 2^-k as a float)
typedef _Float16 dh_h2 __attribute__((ext_vector_type(2)));
// (Round 4 tried the second half as ONE fused operation per sample, h2 = f16(fma(x, 2^11 scale, -2^11 h1)) with v_fma_mixlo / mixhi_f16
// reading 2^11 h1 as a half: bit-identical, two and a half instructions per sample instead of three and a half -- and no faster,
// because v_fma_mix* issues at 8.2 cycles per wavefront against 4.2 for the conversions and packed operations it replaces
// (tools/microbench/valu_rate.hip).  Dropped.)
__device__ __forceinline__ void dh_f16_split4(const dh_f4& v, float scale, dh_h4& h1, dh_h4& h2) {
    const float x[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float xs = x[i] * scale;                            // exact (power of two)
        const _Float16 a = (_Float16) xs;                         // v_cvt_f16_f32: round to nearest even
        const float r = xs - (float) a;                           // exact
        h1[i] = a; h2[i] = (_Float16) (r * 2048.0f);
    }
}
// acc16[4 T + r] of lane (n, g) = y-before-gain of output DH_F16_OUT(T, r, g, n) in the scaled domain: main + 2^-11 second sum
// is formed by the caller's fma (k1, k2).  `hw` = the window block (h1 at byte 0, h2 at DH_F16_H2_OFFSET words), G = the six
// tap fragments of this lane.
template <int NZ>
__device__ __forceinline__ void dh_fir_f16(const float* hw, const dh_u4 (&G)[2 * DH_F16_KSTEPS_OF(NZ)], int lane, float k1, float k2, float* out16) {
    constexpr int DH_F16_KSTEPS = DH_F16_KSTEPS_OF(NZ), DH_F16_FRAGS = 2 * DH_F16_KSTEPS, DH_F16_H2_OFFSET = DH_F16_H2_OFFSET_OF(NZ);
    const int m = lane & 15, q = lane >> 4;
    const char* base = reinterpret_cast<const char*>(hw) + 32 * (int) DH_F16_ROW_BLOCK(m) + 16 * q;
    dh_h8 g[DH_F16_FRAGS];
#pragma unroll
    for (int f = 0; f < DH_F16_FRAGS; f++) g[f] = __builtin_bit_cast(dh_h8, G[f]);
    // Two tiles at a time, so that four accumulation chains (main and second sum of each) alternate and no MFMA waits for
    // the one before it.  Only six 16-byte fragments of the window are live at once: the h2 fragment of a K step is read
    // into the registers of the h1 fragment the step has just finished with, two steps before it is needed.
#pragma unroll
    for (int TT = 0; TT < 4; TT += 2) {
        dh_h8 a1[2][DH_F16_KSTEPS], a2[2][DH_F16_KSTEPS];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int s = 0; s < DH_F16_KSTEPS; s++) a1[t][s] = *reinterpret_cast<const dh_h8*>(base + 512 * (TT + t) + 64 * s);
        dh_f32x4 mn[2] = { { 0.0f, 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f, 0.0f } }, sc[2] = { { 0.0f, 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f, 0.0f } };
#pragma unroll
        for (int s = 0; s < DH_F16_KSTEPS; s++) {
#pragma unroll
            for (int t = 0; t < 2; t++) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[t][s], g[DH_F16_KSTEPS + s], sc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; t++) mn[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[t][s], g[s], mn[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; t++) a2[t][s] = *reinterpret_cast<const dh_h8*>(base + 512 * (TT + t) + 64 * s + 4 * DH_F16_H2_OFFSET);
        }
#pragma unroll
        for (int s = 0; s < DH_F16_KSTEPS; s++)
#pragma unroll
            for (int t = 0; t < 2; t++) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[t][s], g[s], sc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) out16[4 * (TT + t) + r] = __builtin_fmaf(sc[t][r], k2, mn[t][r] * k1);
        // scheduling groups (DS read = 0x100, MFMA = 0x008): the order written above
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * DH_F16_KSTEPS, 0);
#pragma unroll
        for (int s = 0; s < DH_F16_KSTEPS; s++) { __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * DH_F16_KSTEPS, 0);
    }
}
#else
// harness: the same split and the same sums as plain f32 chains (products of halves are exact in f32; a sequential chain of
// 96 rounded additions stays inside what (H1) allows the hardware, so the decisions downstream are the same)
inline void dh_f16_split4(const dh_f4& v, float scale, uint16_t* h1, uint16_t* h2) {
    const float x[4] = { v.x, v.y, v.z, v.w };
    for (int i = 0; i < 4; i++) {
        const float xs = x[i] * scale;
        h1[i] = dh_f16_bits(xs);
        const float r = xs - dh_f16_value(h1[i]);
        h2[i] = dh_f16_bits(r * 2048.0f);
    }
}
template <int NZ>
inline void dh_fir_f16(const float* hw, const uint32_t (*G)[DH_WAVE][4], int lane, float k1, float k2, float* out16) {
    constexpr int DH_F16_KSTEPS = DH_F16_KSTEPS_OF(NZ), DH_F16_H2_OFFSET = DH_F16_H2_OFFSET_OF(NZ);
    const uint16_t* h1 = reinterpret_cast<const uint16_t*>(hw);
    const uint16_t* h2 = reinterpret_cast<const uint16_t*>(hw + DH_F16_H2_OFFSET);
    const int n = lane & 15, g = lane >> 4;
    for (int T = 0; T < 4; T++) for (int r = 0; r < 4; r++) {
        const uint32_t o = DH_F16_OUT(T, r, g, n), b16 = o - (uint32_t) n;
        float mn = 0.0f, sc = 0.0f;
        for (int pass = 0; pass < 3; pass++)                       // main; taps' second halves; samples' second halves (the device's order)
            for (int s = 0; s < DH_F16_KSTEPS; s++) for (int qq = 0; qq < 4; qq++) for (int j = 0; j < 8; j++) {
                const int kk = 8 * qq + j;                         // B fragment of lane (n, qq): element j
                const uint32_t w = G[(pass == 1 ? DH_F16_KSTEPS : 0) + s][16 * qq + n][j >> 1];
                const float tap = dh_f16_value((uint16_t) (w >> (16 * (j & 1))));
                const uint16_t hv = (pass == 2 ? h2 : h1)[b16 + 32u * (uint32_t) s + (uint32_t) kk];
                if (pass == 0) mn = __builtin_fmaf(dh_f16_value(hv), tap, mn); else sc = __builtin_fmaf(dh_f16_value(hv), tap, sc);
            }
        out16[4 * T + r] = __builtin_fmaf(sc, k2, mn * k1);
    }
}
#endif

// Four consecutive floats to LDS at `base` + a compile-time offset (the padded window is only dword aligned, so the
// widest store the hardware takes is a dword pair; the compiler adds the offset to the base with one VALU instruction
// per pair because a ds_write2_b32 offset only reaches 255 dwords -- four ds_write_b32 with 16-bit byte offsets cost
// LDS issue slots, which this kernel has to spare, and no VALU at all).  The stores are invisible to the compiler's
// wait-count tracking: dh_lds_stores_done() must come before the barrier that publishes them.
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
template <int OFF_WORDS> __device__ __forceinline__ void dh_lds_store4_at(float* base, const dh_f4& v) {
    static_assert(4 * OFF_WORDS + 12 < 65536, "ds_write_b32 offsets are 16-bit byte counts");
    const uint32_t addr = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) float*) base;
    asm volatile("ds_write_b32 %0, %1 offset:%5\n\tds_write_b32 %0, %2 offset:%6\n\tds_write_b32 %0, %3 offset:%7\n\tds_write_b32 %0, %4 offset:%8"
                 :: "v"(addr), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "n"(4 * OFF_WORDS), "n"(4 * OFF_WORDS + 4), "n"(4 * OFF_WORDS + 8), "n"(4 * OFF_WORDS + 12) : "memory");
}
__device__ __forceinline__ void dh_lds_stores_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// ten floats into one column of the transposed variance ring (row i is DH_VARIANCE_SYMBOLS words further; K = column offset):
// ten ds_write_b32 with immediate offsets from one address, no vector arithmetic (see dh_lds_store4_at)
template <int K> __device__ __forceinline__ void dh_lds_store_row10(float* base, float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8, float a9) {
    const uint32_t addr = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) float*) base;
    asm volatile("ds_write_b32 %0, %1 offset:%11\n\tds_write_b32 %0, %2 offset:%12\n\tds_write_b32 %0, %3 offset:%13\n\tds_write_b32 %0, %4 offset:%14\n\t"
                 "ds_write_b32 %0, %5 offset:%15\n\tds_write_b32 %0, %6 offset:%16\n\tds_write_b32 %0, %7 offset:%17\n\tds_write_b32 %0, %8 offset:%18\n\t"
                 "ds_write_b32 %0, %9 offset:%19\n\tds_write_b32 %0, %10 offset:%20"
                 :: "v"(addr), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(a8), "v"(a9),
                    "n"(4 * K), "n"(4 * (K + 100)), "n"(4 * (K + 200)), "n"(4 * (K + 300)), "n"(4 * (K + 400)), "n"(4 * (K + 500)), "n"(4 * (K + 600)),
                    "n"(4 * (K + 700)), "n"(4 * (K + 800)), "n"(4 * (K + 900)) : "memory");
}
#else
template <int K> inline void dh_lds_store_row10(float* base, float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8, float a9) {
    const float a[10] = { a0, a1, a2, a3, a4, a5, a6, a7, a8, a9 };
    for (int i = 0; i < 10; i++) base[i * 100 + K] = a[i];
}
template <int OFF_WORDS> inline void dh_lds_store4_at(float* base, const dh_f4& v) { dh_store4(base + OFF_WORDS, v); }
inline void dh_lds_stores_done() {}
#endif

// virtual input stream of a channel for this push: carried tail followed by the new samples
DH_HD float dh_virtual_sample(const float* tail, uint32_t tc, const float* in, uint32_t idx) {
    return idx < tc ? tail[idx] : in[idx - tc];
}

// ---------------------------------------------------------------------------------------------
// AGC window extremes for the symbols k0..k1-1 of the current block (calibrateAudio,
// gfsk_demodulator.cpp:109-116): S.mn[k] / S.mx[k] = min / max over the volume ring as it stands
// right after symbol k was written.  The reference seeds min with FLT_MAX and max with FLT_MIN
// (sic: the smallest positive float), which are exactly the identities used for padding here.
#define DH_FLT_MAX 3.402823466e+38f
#define DH_DBL_MAX 1.7976931348623157e+308
#define DH_FLT_MIN 1.175494351e-38f
// an upper bound of sqrt(x), x >= 0 (a tolerance may be generous, never short): the hardware's v_sqrt_f32 (1 ulp, one
// instruction -- the correctly rounded square root this file is compiled for is twenty) on an argument kept away from
// the denormals, times 1 + 2^-20
DH_HD float dh_sqrt_upper(float x) {
    const float a = x > 1e-30f ? x : 1e-30f;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(a) * 1.00000095367431640625f;
#else
    return __builtin_sqrtf(a) * 1.00000095367431640625f;
#endif
}
DH_HD float dh_fmin_(float a, float b) { return b < a ? b : a; }
DH_HD float dh_fmax_(float a, float b) { return b > a ? b : a; }

#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
// v_min_f32 / v_max_f32 return the operand that is not NaN -- exactly what `if (v < min) min = v` does with a
// NaN entry (it is skipped) -- and cost one VALU instruction where compare + select cost two.
__device__ __forceinline__ float dh_vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float dh_vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Inclusive wave64 prefix min (mn) and max (mx) in lane order, on the VALU's DPP network: row_shr 1/2/4/8 inside
// each row of 16, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3.  Lanes without a source
// keep their value.  A DPP read needs two wait states after the VALU write of its source: the min / max streams
// are interleaved and padded with s_nop (inline asm gets no hazard handling from the compiler).
__device__ __forceinline__ void dh_wave_prefix_minmax(float& mn, float& mx) {
#define DH_SCAN_STEP(ctrl) \
    "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\t"
    asm volatile("s_nop 4\n\t"                 // a VALU write of EXEC just before would need 5 wait states ahead of a DPP op
                 DH_SCAN_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
                 DH_SCAN_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                 DH_SCAN_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
                 DH_SCAN_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                 DH_SCAN_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 DH_SCAN_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(mn), "+v"(mx));
#undef DH_SCAN_STEP
}
// The same for TWO pairs at once (prefix and suffix scans of the AGC): four streams interleaved, so that a DPP read always finds its
// source written three instructions earlier (two wait states are needed) -- no s_nop between the steps, 25 instructions instead of 40.
__device__ __forceinline__ void dh_wave_prefix_minmax2(float& mn, float& mx, float& mn2, float& mx2) {
#define DH_SCAN_STEP4(ctrl) \
    "v_min_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\tv_min_f32_dpp %2, %2, %2 " ctrl "\n\tv_max_f32_dpp %3, %3, %3 " ctrl "\n\t"
    asm volatile("s_nop 4\n\t"
                 DH_SCAN_STEP4("row_shr:1 row_mask:0xf bank_mask:0xf")
                 DH_SCAN_STEP4("row_shr:2 row_mask:0xf bank_mask:0xf")
                 DH_SCAN_STEP4("row_shr:4 row_mask:0xf bank_mask:0xf")
                 DH_SCAN_STEP4("row_shr:8 row_mask:0xf bank_mask:0xf")
                 DH_SCAN_STEP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 DH_SCAN_STEP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(mn), "+v"(mx), "+v"(mn2), "+v"(mx2));
#undef DH_SCAN_STEP4
}
// the previous lane's value of four registers at once (lane 0 keeps the identities it is given)
__device__ __forceinline__ void dh_wave_prev4(float& a, float& b, float& c, float& d, float va, float vb, float vc, float vd) {
    asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %2, %6 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(va), "v"(vb), "v"(vc), "v"(vd));
}
// value of the previous lane (lane 0 keeps `first`): wave_shr:1
__device__ __forceinline__ float dh_wave_prev(float v, float first) {
    float r = first;
    asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(r) : "v"(v));
    return r;
}

// lane l owns ring slots 2l and 2l+1 for the prefix part; the suffix part is the same scan over the ring read
// backwards (lane l takes slots 126-2l and 127-2l), handed back to the owning lanes -- lane 63 - l -- by four ds_bpermute
// (until round 5 through S.mn / S.mx: four stores, a barrier, four loads, a barrier -- one LDS round trip more).  The lane's
// results (slots 2l and 2l+1) are also returned: the slicing phase of a run that starts its block takes them from there.
struct DhAgcPair { float mn0, mx0, mn1, mx1; };
__device__ __forceinline__ DhAgcPair dh_agc_scan(DhDspShared& S, uint32_t k0, uint32_t k1) {
    const int lane = dh_fresh_lane_id_();
    const uint32_t e0 = 2u * (uint32_t) lane, e1 = e0 + 1u;
    // prefix source: old below k0, new in [k0, k1), identity above
    const float c0 = e0 < k0 ? S.vol_old[e0] : S.vol_new[e0];
    const float c1 = e1 < k0 ? S.vol_old[e1] : S.vol_new[e1];
    const bool v0 = e0 < k1, v1 = e1 < k1;
    const float pmn0 = v0 ? c0 : DH_FLT_MAX, pmn1 = v1 ? c1 : DH_FLT_MAX;
    const float pmx0 = v0 ? c0 : DH_FLT_MIN, pmx1 = v1 ? c1 : DH_FLT_MIN;
    float pmn = dh_vmin(pmn0, pmn1), pmx = dh_vmax(pmx0, pmx1);
    // suffix source: the old ring backwards (slots >= 100 hold the identity)
    const uint32_t r0 = 126u - e0, r1 = r0 + 1u;
    const float o0 = S.vol_old[r0], o1 = S.vol_old[r1];
    const float omn0 = r0 < DH_VOLUME_RB_SIZE ? o0 : DH_FLT_MAX, omn1 = r1 < DH_VOLUME_RB_SIZE ? o1 : DH_FLT_MAX;
    const float omx0 = r0 < DH_VOLUME_RB_SIZE ? o0 : DH_FLT_MIN, omx1 = r1 < DH_VOLUME_RB_SIZE ? o1 : DH_FLT_MIN;
    float smn = dh_vmin(omn0, omn1), smx = dh_vmax(omx0, omx1);
    dh_wave_prefix_minmax2(pmn, pmx, smn, smx);
    // exclusive parts: everything before this lane's pair (prefix), everything after slot r1 (suffix)
    float epmn = DH_FLT_MAX, epmx = DH_FLT_MIN, esmn = DH_FLT_MAX, esmx = DH_FLT_MIN;
    dh_wave_prev4(epmn, epmx, esmn, esmx, pmn, pmx, smn, smx);
    // suffix (exclusive of the slot itself) for slots r1 (esmn / esmx) and r0; slot e of lane l is slot r of lane 63 - l
    const int from = (63 - lane) << 2;
    const float s1n = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(esmn)));
    const float s1x = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(esmx)));
    const float s0n = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(dh_vmin(omn1, esmn))));
    const float s0x = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(dh_vmax(omx1, esmx))));
    // slot e0: prefix through e0 = before this lane + own e0; slot e1: inclusive prefix of the lane
    DhAgcPair r;
    r.mn0 = dh_vmin(dh_vmin(epmn, pmn0), s0n); r.mx0 = dh_vmax(dh_vmax(epmx, pmx0), s0x);
    r.mn1 = dh_vmin(pmn, s1n); r.mx1 = dh_vmax(pmx, s1x);
    S.mn[e0] = r.mn0; S.mx[e0] = r.mx0; S.mn[e1] = r.mn1; S.mx[e1] = r.mx1;
    return r;
}
#else
inline void dh_agc_scan(DhDspShared& S, uint32_t k0, uint32_t k1) {
    // plain sequential statement of the same thing (harness only)
    float pmn = DH_FLT_MAX, pmx = DH_FLT_MIN;
    for (uint32_t j = 0; j < k0; j++) { pmn = dh_fmin_(pmn, S.vol_old[j]); pmx = dh_fmax_(pmx, S.vol_old[j]); }
    for (uint32_t k = k0; k < k1; k++) {
        pmn = dh_fmin_(pmn, S.vol_new[k]); pmx = dh_fmax_(pmx, S.vol_new[k]);
        float smn = DH_FLT_MAX, smx = DH_FLT_MIN;
        for (uint32_t j = k + 1; j < DH_VOLUME_RB_SIZE; j++) { smn = dh_fmin_(smn, S.vol_old[j]); smx = dh_fmax_(smx, S.vol_old[j]); }
        S.mn[k] = dh_fmin_(pmn, smn); S.mx[k] = dh_fmax_(pmx, smx);
    }
}
#endif

#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
// minimum of lanes 0..15 (one DPP row), valid in lane 15 and returned wave-uniform: four v_min_f32 with row_shr
__device__ __forceinline__ float dh_row_min_to_lane15(float v) {
    asm volatile("s_nop 4\n\tv_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                 : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
}
#endif

// ---------------------------------------------------------------------------------------------
// Error-bounded FIR (DH_BOUNDED_FIR; the exact wide-filter kernels at sps 10, i.e. the DMR / YSF pipes).
//
// The reference's dibits and timing steps depend on the filtered samples only through comparisons.  The kernel therefore
// filters with the fused multiply-add FIR (half the vector instructions of the rounded-product chain), carries a proven
// bound on how far each value can be from the reference's, decides every comparison that the bound leaves no doubt about,
// and evaluates the rest -- a few symbols in ten thousand on noise, none on a clean signal -- with the reference's own
// arithmetic (dh_exact_filtered: rounded products, rounded sums in tap order, the double division).  Outputs are the
// reference's bits; only values that never leave the kernel (AGC ring, variance ring) are approximations.
//
// Bound.  u = 2^-24, A = sum |c_i| |x_i| <= ||c||_1 max|x| over the run's raw window, y = value the reference computes.
//   reference chain: |acc_ref - sum c_i x_i| <= g82 A;  FMA chain: |acc_fma - sum c_i x_i| <= g81 A   (g_k = k u / (1 - k u))
//   y_ref = acc_ref / gain (1 + d1), |d1| <= u (1 + 2^-29) (double division, then one rounding to float)
//   y_fma = acc_fma fl32(1 / gain) (1 + d3) = acc_fma / gain (1 + d2)(1 + d3), |d2|, |d3| <= u (1 + 2^-29)
//   => |y_fma - y_ref| <= (g82 + g81 + 3.1 u (1 + g82)) A / gain <= 167 u A / gain           (no overflow, no underflow:
//   runs whose max |x| is 0 are exact, runs with max |x| outside [1e-25, 1e15] or a non-finite sample use the exact FIR)
// The slicer's own float operations (window sums, /10, (max + min) / 2, the 0.625 thresholds) run on both sides with the
// same operation sequence; on perturbed operands they add at most 62 u A / gain between a mid-symbol average and a
// threshold (counted operation by operation in DESIGN.md section 3: 39.8 u A / gain for both sides together).  With
//   e = 240 u (||c||_1 / gain) max|x|      (dh_fir_error_coefficient: 240 u ||c||_1 / gain, rounded up)
// an average and a threshold computed from values with radii <= e differ from the reference's difference by at most
// (1 + 2.25) 167/240 e + 62/240 e < 2.53 e; a comparison is DECIDED when they are further apart than T = 3.5 e.
// e of a ring entry is the e of the run that produced it: kept as two maxima over >= 100-symbol buckets (DH_ST_E_*).
#define DH_BOUNDED_FIR 1
#define DH_BOUNDED_NARROW 1                 // the same for the narrow filter at a run-time samples-per-symbol (the NXDN pipe)
template <int NZ, bool FAST, int SPS> struct DhIsBounded {
    static constexpr bool value = DH_BOUNDED_FIR && !FAST && ((NZ == 80 && SPS == 10) || (DH_BOUNDED_NARROW && NZ == 160 && SPS != 10));
};
// Diagnostic builds only (tools/phase_budget.sh): -DDH_STOP_AFTER=n leaves out the phases after Pn of every run (the
// results are then wrong; the instruction counters of such builds, subtracted from each other, give the per-phase budget)
#ifndef DH_STOP_AFTER
#define DH_STOP_AFTER 99
#endif
#define DH_BOUND_T_FACTOR 3.5f
#define DH_BOUND_XMAX_LO 1e-25f
#define DH_BOUND_XMAX_HI 1e15f

// kappa u ||c||_1 / gain, rounded up (host side, once per engine).  kappa = 240 for the 81 taps the bound above was written
// for (gamma_82 + gamma_81 + 3.1 u = 167 u, times 1.44 of head room for the slicer's own roundings); for n taps the two
// accumulation chains give (2 n + 5) u, and the same proportion is kept: kappa = 1.44 (2 n + 5).  The slicer's roundings
// grow with the samples per symbol (sums of sps terms): both sides together <= 2 (avg + umid) u A / gain with
//   vol = (sps (sps + 1) / 2 - 1) / sps + 1,  centre = vol + 1,  umid = (vol + centre + 1) 0.625 + centre + 1,
//   avg = (w (w + 1) / 2 - 1) / w + 1 for a mid-symbol window of w samples
// = 68 at sps 20 (w 6), 119 at sps 40 (w 14): (3.25 (2 n + 5) + that) / kappa stays below 2.8 for every filter and sps the
// engine accepts, under the decision threshold T = 3.5 e.
inline float dh_fir_error_coefficient(const float* taps_half, uint32_t nz, double gain) {
    double l1 = 0.0;
    for (uint32_t i = 0; i <= nz; i++) { const double c = taps_half[i <= nz / 2 ? i : nz - i]; l1 += c < 0 ? -c : c; }
    const double kappa = nz == 80 ? 240.0 : 1.44 * (2.0 * (nz + 1) + 5.0);
    const double coef = kappa * 5.9604644775390625e-08 * l1 / (gain < 0 ? -gain : gain) * 1.0001;
    float f = (float) coef;
    if ((double) f < coef) { union { float f; uint32_t u; } b; b.f = f; b.u++; f = b.f; }
    return f;
}

// The reference's filtered sample at filtered position f of this push's virtual stream V = tail ++ in
// (rrc_filter.cpp:22-34 for one output): needs V[f .. f + NZ].  Per lane, no barriers.  tapsf = first half + centre.
template <int NZ>
DH_HD float dh_exact_filtered(const float* tail, uint32_t tc, const float* in, uint32_t nv, const float* tapsf, double gain, double rgain, int32_t f) {
    if (f < 0 || (uint32_t) f + (uint32_t) NZ >= nv) return 0.0f;       // not available (callers only ask for positions they hold)
    float acc = 0.0f;
    for (int i = 0; i <= NZ; i++) {
        const float c = tapsf[i <= NZ / 2 ? i : NZ - i];
        const float x = dh_virtual_sample(tail, tc, in, (uint32_t) f + (uint32_t) i);
        const float prod = c * x;
        acc = acc + prod;
    }
    return dh_div_gain(acc, gain, rgain);
}

// maximum over the wavefront of a per-lane value (device: six ds_swizzle / bpermute exchanges)
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
// (row_shr 1/2/4/8 inside each row of 16, row_bcast:15 / :31 across rows -- the running maximum ends up in lane 63;
// v_max_f32 skips a NaN operand, which is what the caller wants: NaN samples are caught behind the FIR)
__device__ __forceinline__ float dh_wave_max(float v) {
#define DH_MAX_STEP(ctrl) "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 " ctrl "\n\t"
    asm volatile("s_nop 4\n\t"
                 DH_MAX_STEP("row_shr:1 row_mask:0xf bank_mask:0xf") DH_MAX_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
                 DH_MAX_STEP("row_shr:4 row_mask:0xf bank_mask:0xf") DH_MAX_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
                 DH_MAX_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf") DH_MAX_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1" : "+v"(v));
#undef DH_MAX_STEP
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// max(|a|, |b|, c) in one instruction
__device__ __forceinline__ float dh_max3_abs(float a, float b, float c) {
    float r; asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
#endif

// The rare paths of the error-bounded kernels (inlined: as real calls they would force the kernel's argument block into
// scratch memory).  What they need to know lives in LDS between runs (DhBoundState), not in registers: every scalar that
// stays live across the FIR -- where all 128 registers are taken -- is a spill in the hot loop.
#define DH_COLD DH_HD

// bookkeeping of the error-bounded kernels between runs, in LDS (words 48.. of the tap block: the wide filter uses 41)
struct DhBoundState {
    int32_t cur_start, cur_off, prev_start, prev_off;   // filtered positions of symbol 0 of the current / previous variance block, their offsets
    uint32_t blk_flags;                                 // bit 0: the current block's start is known, bit 1: the previous block's
    uint32_t e_count, n_uncertain, n_exact_runs, n_exact_blocks;
    float e_cur, e_prev, e_blk;                         // error radii: current / previous >= 100-symbol bucket, current variance block
};
#define DH_BOUND_STATE(S) (reinterpret_cast<DhBoundState*>((S).bound))
DH_HD float dh_uniform_f(float x) { union { float f; uint32_t u; } b; b.f = x; b.u = dh_uniform(b.u); return b.f; }

// What the exact evaluation of one symbol needs to know about the stream (error-bounded kernels)
struct DhExactCtx {
    const float* tail; uint32_t tc; const float* in; uint32_t nv;      // the virtual stream V = tail ++ in
    const float* tapsf; double gain, rgain; float sps_rcp;
    int32_t cur_start, cur_off, prev_start, prev_off; uint32_t blk_flags;
    uint32_t k0;                                                        // ring slots [k0, k] hold this run's volumes (S.vol_new)
    float e_eff; int32_t levels, invert;
};
// filtered position of sample 0 of ring slot j as seen from symbol k of the current block: slots <= k belong to the
// current block, the others still hold the previous block's symbols; INT32_MIN = a slot never written (exact zero)
DH_HD int32_t dh_slot_position(const DhExactCtx& C, uint32_t j, uint32_t k, uint32_t sps) {
    if (j <= k) return (C.blk_flags & 1u) ? C.cur_start + (int32_t) (j * sps) + (j ? C.cur_off : 0) : INT32_MIN;
    return (C.blk_flags & 2u) ? C.prev_start + (int32_t) (j * sps) + (j ? C.prev_off : 0) : INT32_MIN;
}

// ---- the same two exact evaluations with the raw samples staged through LDS (generic sps; used by the narrow filter, whose
// 161-tap chains would otherwise fetch every sample with a dependent load of its own from HBM / L2)
// raw[0 ..] = V[base ..]: the reference's filtered sample at filtered position f (rrc_filter.cpp:22-34), 0 where the stream
// does not hold all of V[f .. f + NZ]
template <int NZ>
DH_HD float dh_exact_filtered_lds(const float* raw, int32_t base, uint32_t nv, const float* tapsf, double gain, double rgain, int32_t f) {
    if (f < 0 || (uint32_t) f + (uint32_t) NZ >= nv) return 0.0f;
    const float* x = raw + (f - base);
    float acc = 0.0f;
    // eight products at a time, their sixteen LDS reads in flight together, then the eight additions in tap order (written
    // tap by tap this compiled to a read - wait - multiply - add per tap: 110 cycles each, 620 000 for a variance ring)
    // (the 161-tap kernels only: in the 81-tap ones, at 128 registers, the eight products push spills into the hot loop --
    // 24 scratch accesses in the commit phase, DMR chain 6.8 -> 11.9 ms -- and their exact evaluations are 4x rarer and 2x shorter)
    int i = 0;
    constexpr int B = NZ > 80 ? 8 : DH_EXACT_BATCH;
    for (; B > 1 && i + B <= NZ + 1; i += B) {
        float prod[B];
#pragma unroll
        for (int j = 0; j < B; j++) { const int t = i + j; prod[j] = tapsf[t <= NZ / 2 ? t : NZ - t] * x[t]; }
#pragma unroll
        for (int j = 0; j < B; j++) acc = acc + prod[j];
    }
    for (; i <= NZ; i++) {
        const float prod = tapsf[i <= NZ / 2 ? i : NZ - i] * x[i];
        acc = acc + prod;
    }
    return dh_div_gain(acc, gain, rgain);
}
// V[base .. base + count) into raw[] (zeros outside the stream); barriers on both sides
DH_HD void dh_stage_raw(float* raw, const float* tail, uint32_t tc, const float* in, uint32_t nv, int32_t base, uint32_t count) {
    DH_BARRIER();
    DH_FOR_LANES_FRESH(lane) {
        // four loads per lane in flight (one address each: the tail or the input), then their stores
        for (uint32_t e0 = (uint32_t) lane; e0 < count; e0 += 4u * DH_WAVE) {
            float v[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) {
                const uint32_t e = e0 + u * DH_WAVE;
                const int32_t idx = base + (int32_t) e;
                const bool have = e < count && idx >= 0 && (uint32_t) idx < nv;
                const uint32_t at = have ? (uint32_t) idx : tc;                  // (any valid address for the lanes that store a zero)
                const float* src = at < tc ? tail + at : in + (at - tc);
                v[u] = (have || nv > tc) ? *src : 0.0f;
                if (!have) v[u] = 0.0f;
            }
#pragma unroll
            for (uint32_t u = 0; u < 4u; u++) { const uint32_t e = e0 + u * DH_WAVE; if (e < count) raw[e] = v[u]; }
        }
    }
    DH_BARRIER();
}
// `stage`: `cap` LDS floats free at the caller's phase.  The mid-symbol window of symbol k and the ten (sps) samples of every
// candidate slot are evaluated TOGETHER: their raw samples are staged side by side (areas of STRIDE words), all chains run at once
// on different lanes, one round of two barriers for the window and up to A - 1 slots -- one after the other (a staging round trip
// to HBM / L2 and an 81-long dependent chain each, three or four times per doubtful symbol) these evaluations were 3 % of the DMR
// chain kernel's time for 0.008 % of its symbols.
template <int NZ>
DH_COLD uint8_t dh_exact_symbol_staged(const DhExactCtx& C, DhDspShared& S, uint32_t k, float* stage, uint32_t cap, uint32_t sps, uint32_t LO, uint32_t HI) {
    const uint32_t W = HI - LO;
    const uint32_t STRIDE = ((W > sps ? W : sps) + (uint32_t) NZ + 1u + 3u) & ~3u;
    const uint32_t A = (cap - 80u) / STRIDE;                          // areas in front of 64 result words + 16 words of slot table (A >= 2: area 0 is the window's)
    // (per_round >= 1 -- or the loop below would never clear its bits -- needs sps + W <= DH_WAVE, W = round(2 sps / 3) - round(sps / 3)
    // <= sps / 3 + 1, and two areas in `stage`: both hold for every sps up to DH_MAX_SPS with the window blocks of dh_dsp_xf_words().
    // The staging areas run over S.sum, behind the filtered samples: the slicing phase has consumed it by the time this is called.)
    static_assert(DH_MAX_SPS + DH_MAX_SPS / 3 + 2 <= DH_WAVE, "a candidate slot's samples and the evaluation window fit one round of lanes");
    static_assert(DH_MAX_SPS + DH_MAX_NZ + 1 <= 4 * DH_WAVE, "an area is staged with four elements per lane");
    uint32_t per_round = dh_min<uint32_t>(A - 1u, (DH_WAVE - W) / sps);
    if (per_round > 8u) per_round = 8u;
    float* scratch = stage + A * STRIDE;
    int32_t* tbl_pos = reinterpret_cast<int32_t*>(scratch + DH_WAVE);  // [8] filtered position of the round's slots (INT32_MIN: never written)
    uint32_t* tbl_slot = reinterpret_cast<uint32_t*>(scratch + DH_WAVE + 8);
    const float mn_a = S.mn[k], mx_a = S.mx[k];
    const float lo_thr = mn_a + 2.5f * C.e_eff, hi_thr = mx_a - 2.5f * C.e_eff;
    uint64_t cand_lo[2] = { 0, 0 }, cand_hi[2] = { 0, 0 };
    for (int h = 0; h < 2; h++) {
        uint64_t vlo = 0, vhi = 0;
        DH_FOR_LANES_FRESH(lane) {
            const uint32_t j = (uint32_t) lane + 64u * (uint32_t) h;
            bool lo = false, hi = false;
            if (j < DH_VOLUME_RB_SIZE) {
                const float v = (j >= C.k0 && j <= k) ? S.vol_new[j] : S.vol_old[j];
                lo = v <= lo_thr; hi = v >= hi_thr;
            }
            DH_BALLOT_ACC(vlo, lo, lane); DH_BALLOT_ACC(vhi, hi, lane);
        }
        cand_lo[h] = vlo; cand_hi[h] = vhi;
    }
    const int32_t mid_base = dh_slot_position(C, k, k, sps) + (int32_t) LO;
    float exact_min = DH_FLT_MAX, exact_max = DH_FLT_MIN;            // the reference's seeds (gfsk_demodulator.cpp:110-111)
    float avg_sum = 0.0f;
    bool first = true;
    uint64_t todo[2] = { cand_lo[0] | cand_hi[0], cand_lo[1] | cand_hi[1] };
    while (first || todo[0] || todo[1]) {
        // this round's slots: the first `ns` set bits of todo, their positions into the table
        const uint32_t pending = (uint32_t) (dh_popc64(todo[0]) + dh_popc64(todo[1]));
        const uint32_t ns = dh_min<uint32_t>(pending, per_round);
        DH_BARRIER();                                                  // the previous round's results and table have been read
        DH_FOR_LANES_FRESH(lane) {
            if ((uint32_t) lane < ns) {
                uint64_t t0 = todo[0], t1 = todo[1];
                for (uint32_t b = 0; b < (uint32_t) lane; b++) { if (t0) t0 &= t0 - 1; else t1 &= t1 - 1; }
                const uint32_t j = t0 ? (uint32_t) dh_ffs64(t0) : 64u + (uint32_t) dh_ffs64(t1);
                tbl_slot[lane] = j; tbl_pos[lane] = dh_slot_position(C, j, k, sps);
            }
        }
        for (uint32_t b = 0; b < ns; b++) { if (todo[0]) todo[0] &= todo[0] - 1; else todo[1] &= todo[1] - 1; }
        DH_BARRIER();
        // raw samples, area by area (wave-uniform loop: area 0 = V[mid_base ..), first round only; area 1 + c = V[pos_c ..)): a lane takes
        // elements lane, lane + 64, ... of the area, its (up to four) loads in flight together.  (As ONE loop over all areas' elements
        // every element paid a division by the area stride, a run-time value wherever the evaluation window is.)
        for (uint32_t a = first ? 0u : 1u; a <= ns; a++) {
            const int32_t base = a == 0u ? mid_base : (int32_t) dh_uniform((uint32_t) tbl_pos[a - 1u]);
            const uint32_t count = (a == 0u ? W : sps) + (uint32_t) NZ + 1u;
            float* const area = stage + a * STRIDE;
            DH_FOR_LANES_FRESH(lane) {
                float v[4];
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) {
                    const uint32_t i = (uint32_t) lane + u * DH_WAVE;
                    const int32_t idx = base + (int32_t) i;
                    const bool have = i < count && base != INT32_MIN && idx >= 0 && (uint32_t) idx < C.nv;
                    const uint32_t at = have ? (uint32_t) idx : C.tc;            // (any valid address for the lanes that store a zero)
                    const float* src = at < C.tc ? C.tail + at : C.in + (at - C.tc);
                    v[u] = (have || C.nv > C.tc) ? *src : 0.0f;
                    if (!have) v[u] = 0.0f;
                }
#pragma unroll
                for (uint32_t u = 0; u < 4u; u++) { const uint32_t i = (uint32_t) lane + u * DH_WAVE; if (i < count) area[i] = v[u]; }
            }
        }
        DH_BARRIER();
        DH_FOR_LANES_FRESH(lane) {
            // lanes 0 .. W - 1: the evaluation window (first round); W + c sps + i: sample i of the round's slot c -- one chain per lane,
            // from ONE call site (area, position and sample index are per-lane values)
            float y = 0.0f;
            uint32_t area = 0; int32_t s0 = mid_base, f = mid_base + lane; bool run = (uint32_t) lane < W && first;
            if ((uint32_t) lane >= W) {
                const uint32_t c = ((uint32_t) lane - W) / sps, i = ((uint32_t) lane - W) - c * sps;
                if (c < ns) { s0 = tbl_pos[c]; f = s0 + (int32_t) i; area = 1u + c; run = s0 != INT32_MIN; }
            }
            if (run) y = dh_exact_filtered_lds<NZ>(stage + area * STRIDE, s0, C.nv, C.tapsf, C.gain, C.rgain, f);
            scratch[lane] = y;
        }
        DH_BARRIER();
        if (first) { for (uint32_t i = 0; i < W; i++) avg_sum += scratch[i]; first = false; }
        for (uint32_t c = 0; c < ns; c++) {
            float vol = 0.0f;                                        // (a slot never written holds an exact zero: ten zeros above)
            for (uint32_t i = 0; i < sps; i++) vol += scratch[W + c * sps + i];
            vol = dh_div_const(vol, (float) sps, C.sps_rcp);
            const uint32_t j = dh_uniform(tbl_slot[c]);
            if ((cand_lo[j >> 6] >> (j & 63u)) & 1u) exact_min = dh_fmin_(exact_min, vol);
            if ((cand_hi[j >> 6] >> (j & 63u)) & 1u) exact_max = dh_fmax_(exact_max, vol);
        }
    }
    const float center = (exact_max + exact_min) / 2.0f;
    const float average = avg_sum / (float) W;
    if (C.levels == 4) {
        const float umid = __builtin_fmaf(exact_max - center, 0.625f, center);
        const float lmid = __builtin_fmaf(exact_min - center, 0.625f, center);
        if (average > center) return average > umid ? 1 : 0;
        return average < lmid ? 3 : 2;
    }
    return average > center ? (uint8_t) !C.invert : (uint8_t) (C.invert != 0);
}
// the variance ring of the block that just ended, recomputed exactly, a chunk of symbols at a time through `stage` (cap floats)
// `rows`: the phases (rows of the transposed ring) to recompute -- all of them, or only those the float estimate could not
// rule out as the arg-min (the others keep their approximate values and are not looked at: see P6)
template <int NZ>
DH_COLD void dh_exact_var_ring_staged(const DhExactCtx& C, DhDspShared& S, uint32_t sps, float* stage, uint32_t cap, uint64_t rows) {
    const uint32_t spc = (cap - (uint32_t) NZ - 3u) / sps;            // symbols per chunk
    const uint64_t all = sps >= 64u ? ~0ull : ((1ull << sps) - 1ull);
    const bool full = (rows & all) == all;
    const uint32_t nrows = full ? sps : (uint32_t) (dh_popc32((uint32_t) (rows & all)) + dh_popc32((uint32_t) ((rows & all) >> 32)));
    for (uint32_t j0 = 0; j0 < DH_VARIANCE_SYMBOLS; j0 += spc) {
        const uint32_t nj = dh_min<uint32_t>(spc, DH_VARIANCE_SYMBOLS - j0);
        const int32_t base = C.cur_start + (int32_t) (j0 * sps) - 1;      // one sample of slack for the block's +-1 step
        dh_stage_raw(stage, C.tail, C.tc, C.in, C.nv, base, nj * sps + (uint32_t) NZ + 3u);
        for (uint32_t r = 0; r * DH_WAVE < nj * nrows; r++) {
            DH_FOR_LANES_FRESH(lane) {
                const uint32_t e = r * DH_WAVE + (uint32_t) lane;
                if (e < nj * nrows) {
                    uint32_t jj, i;
                    if (full) { jj = e / sps; i = e - jj * sps; }
                    else {
                        const uint32_t c = e / nj;                     // the c-th phase of the set
                        jj = e - c * nj;
                        uint64_t m = rows & all;
                        for (uint32_t b = 0; b < c; b++) m &= m - 1;
                        i = (uint32_t) dh_ffs64(m);
                    }
                    const uint32_t j = j0 + jj;
                    const int32_t f = C.cur_start + (int32_t) (j * sps) + (j ? C.cur_off : 0) + (int32_t) i;
                    S.var_rb[i * DH_VARIANCE_SYMBOLS + j] = dh_exact_filtered_lds<NZ>(stage, base, C.nv, C.tapsf, C.gain, C.rgain, f);
                }
            }
        }
    }
    DH_BARRIER();
}

// the rounded-product FIR over the staged window (runs the bound does not cover): outputs into fo[16] per lane
template <int NZ>
DH_COLD void dh_exact_fir_pass(const DhDspParams& P, DhDspShared& S, uint32_t need, float (*fo_all)[DH_FIR_L], float* fo_dev) {
    float tv[NZ / 2 + 1];
    for (int i = 0; i <= NZ / 2; i++) tv[i] = S.tapsf[i];
    DH_FOR_LANES_FRESH(lane) {
        if ((uint32_t) (lane * DH_FIR_L) < need) {
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
            (void) fo_all;
            dh_fir_lane<NZ, false, false, DH_FIR_F16 != 0>(tv, P.gain, P.rgain, P.inv_gain, S.xf, lane, fo_dev);      // (a rare path next to the split-f16 FIR: loads waited for where they are issued)
#else
            (void) fo_dev;
            dh_fir_lane<NZ, false>(tv, P.gain, P.rgain, P.inv_gain, S.xf, lane, fo_all[lane]);
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ---- P3: symbol windows (gfsk_demodulator.cpp:28-35, 82-83)
// sps 10: two symbols per lane -- q = lane + 1 and (lanes 0..34) q = lane + 65, lane 35: symbol 0, the only one in front of the pending
// timing step -- each read as five ds_read_b64 at compile-time offsets from one per-lane base (eight bytes from a four-byte aligned
// address: the symbols start 40 bytes apart and a timing step moves them by four).  Ten words per lane, read in pairs, touch every
// bank twice per 32 lanes: conflict-free.  Until the end of round 5 the two symbols of a lane were 24 symbols apart and a
// ds_read2_b32 delivered sample i of both as a register pair for packed additions (16 vector instructions less per run): a
// ten-word lane stride read ONE word at a time only ever reaches the sixteen banks of its parity, every read was a two-way conflict --
// SQ_LDS_BANK_CONFLICT 4.75e8 -> 1.74e8 per launch, SQ_LDS_IDX_ACTIVE -21 %, chain -2 % (profiles/r05_a_ab_logs.txt).  Each
// symbol's sums run in sample order, as the reference's do.  Lanes beyond the run read words of the window block that nothing will
// look at and store nothing.
// `fbuf`: the run's filtered samples (the window block); symbol q of the run (ring slot k0 + q) starts at q sps (+ step_off behind symbol 0).
// Leaves S.sum[q], S.vol_new[k0 + q] and column k0 + q of the transposed ring; the caller's barrier publishes them.
template <int SPS>
DH_HD void dh_symbol_windows(DhDspShared& S, const float* fbuf, uint32_t k0, uint32_t m, int32_t step_off, uint32_t sps, uint32_t ev_lo, uint32_t ev_hi, float sps_rcp) {
#define DH_FB(n) fbuf[n]
    if (SPS == 10 && DH_STOP_AFTER >= 3) DH_FOR_LANES_FRESH(lane) {
        const uint32_t l = (uint32_t) lane;
        const uint32_t qa = l + 1u, qb = l < 35u ? l + 65u : 0u;
        const bool va = qa < m, vb = l < 35u ? qb < m : l == 35u;
        const float* srca = fbuf + (int32_t) (qa * 10u) + step_off;
        const float* srcb = l < 35u ? srca + 640 : fbuf;
        float v[10], w[10];
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
        {
            const uint32_t aa = (uint32_t) (uintptr_t) (const __attribute__((address_space(3))) float*) srca;
            const uint32_t ab = (uint32_t) (uintptr_t) (const __attribute__((address_space(3))) float*) srcb;
            dh_f2 x0 = dh_lds_read_b64<0>(aa), x1 = dh_lds_read_b64<8>(aa), x2 = dh_lds_read_b64<16>(aa), x3 = dh_lds_read_b64<24>(aa), x4 = dh_lds_read_b64<32>(aa);
            dh_f2 y0 = dh_lds_read_b64<0>(ab), y1 = dh_lds_read_b64<8>(ab), y2 = dh_lds_read_b64<16>(ab), y3 = dh_lds_read_b64<24>(ab), y4 = dh_lds_read_b64<32>(ab);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4) :: "memory");
            v[0] = x0.x; v[1] = x0.y; v[2] = x1.x; v[3] = x1.y; v[4] = x2.x; v[5] = x2.y; v[6] = x3.x; v[7] = x3.y; v[8] = x4.x; v[9] = x4.y;
            w[0] = y0.x; w[1] = y0.y; w[2] = y1.x; w[3] = y1.y; w[4] = y2.x; w[5] = y2.y; w[6] = y3.x; w[7] = y3.y; w[8] = y4.x; w[9] = y4.y;
        }
#else
        for (int i = 0; i < 10; i++) { v[i] = srca[i]; w[i] = srcb[i]; }
#endif
        float vola = v[0], volb = w[0];
#pragma unroll
        for (int i = 1; i < 10; i++) { vola += v[i]; volb += w[i]; }
        const float mida = ((v[3] + v[4]) + v[5]) + v[6], midb = ((w[3] + w[4]) + w[5]) + w[6];          // samples ev_lo .. ev_hi - 1 = 3 .. 6
        const float volume_a = dh_div_const(vola, 10.0f, sps_rcp), volume_b = dh_div_const(volb, 10.0f, sps_rcp);
        if (va) {
            const uint32_t ka = k0 + qa;
            dh_lds_store_row10<0>(S.var_rb + ka, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]);      // transposed ring: phase-major
            S.sum[qa] = mida; S.vol_new[ka] = volume_a;
        }
        if (vb) {
            const uint32_t kb = k0 + qb;
            dh_lds_store_row10<0>(S.var_rb + kb, w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8], w[9]);
            S.sum[qb] = midb; S.vol_new[kb] = volume_b;
        }
        dh_lds_stores_done();
    }
    // sps 20 (a FIR pass holds at most 51 symbols): one symbol per lane, q = lane + 1, lane 63: symbol 0; its twenty samples as five
    // ds_read_b128 (sixteen bytes from a four-byte aligned address).  Twenty words per lane read in fours touch every bank once per
    // eight lanes; read one word at a time (the general form below) every read was a four-way conflict.
    if (SPS == 20 && DH_STOP_AFTER >= 3) DH_FOR_LANES_FRESH(lane) {
        const uint32_t l = (uint32_t) lane;
        const uint32_t q = l < 63u ? l + 1u : 0u;
        const bool valid = q < m;
        const float* src = valid && q > 0u ? fbuf + (int32_t) (q * 20u) + step_off : fbuf;       // (lanes beyond the run read symbol 0 and store nothing)
        float v[20];
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
        {
            const uint32_t a = (uint32_t) (uintptr_t) (const __attribute__((address_space(3))) float*) src;
            dh_v4f x0 = dh_lds_read_b128<0>(a), x1 = dh_lds_read_b128<16>(a), x2 = dh_lds_read_b128<32>(a), x3 = dh_lds_read_b128<48>(a), x4 = dh_lds_read_b128<64>(a);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4) :: "memory");
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w; v[8] = x2.x; v[9] = x2.y;
            v[10] = x2.z; v[11] = x2.w; v[12] = x3.x; v[13] = x3.y; v[14] = x3.z; v[15] = x3.w; v[16] = x4.x; v[17] = x4.y; v[18] = x4.z; v[19] = x4.w;
        }
#else
        for (int i = 0; i < 20; i++) v[i] = src[i];
#endif
        float vol = v[0];
#pragma unroll
        for (int i = 1; i < 20; i++) vol += v[i];
        const float mid = ((((v[7] + v[8]) + v[9]) + v[10]) + v[11]) + v[12];                    // samples ev_lo .. ev_hi - 1 = 7 .. 12
        const float volume = dh_div_const(vol, 20.0f, sps_rcp);
        if (valid) {
            const uint32_t k = k0 + q;
            dh_lds_store_row10<0>(S.var_rb + k, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]);      // transposed ring: phase-major
            dh_lds_store_row10<1000>(S.var_rb + k, v[10], v[11], v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19]);
            S.sum[q] = mid; S.vol_new[k] = volume;
        }
        dh_lds_stores_done();
    }
    // sps 40 (a pass holds at most 25 symbols): the same with ten ds_read_b128 per symbol (a forty-word lane stride read one word at a
    // time reaches four banks: eight-way conflicts in the general loop)
    if (SPS == 40 && DH_STOP_AFTER >= 3) DH_FOR_LANES_FRESH(lane) {
        const uint32_t l = (uint32_t) lane;
        const uint32_t q = l < 63u ? l + 1u : 0u;
        const bool valid = q < m;
        const float* src = valid && q > 0u ? fbuf + (int32_t) (q * 40u) + step_off : fbuf;       // (lanes beyond the run read symbol 0 and store nothing)
        float v[40];
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
        {
            const uint32_t a = (uint32_t) (uintptr_t) (const __attribute__((address_space(3))) float*) src;
            dh_v4f x0 = dh_lds_read_b128<0>(a), x1 = dh_lds_read_b128<16>(a), x2 = dh_lds_read_b128<32>(a), x3 = dh_lds_read_b128<48>(a), x4 = dh_lds_read_b128<64>(a);
            dh_v4f x5 = dh_lds_read_b128<80>(a), x6 = dh_lds_read_b128<96>(a), x7 = dh_lds_read_b128<112>(a), x8 = dh_lds_read_b128<128>(a), x9 = dh_lds_read_b128<144>(a);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(x8), "+v"(x9) :: "memory");
            const dh_v4f x[10] = { x0, x1, x2, x3, x4, x5, x6, x7, x8, x9 };
#pragma unroll
            for (int i = 0; i < 10; i++) { v[4 * i] = x[i].x; v[4 * i + 1] = x[i].y; v[4 * i + 2] = x[i].z; v[4 * i + 3] = x[i].w; }
        }
#else
        for (int i = 0; i < 40; i++) v[i] = src[i];
#endif
        float vol = v[0], mid = v[13];
#pragma unroll
        for (int i = 1; i < 40; i++) vol += v[i];
#pragma unroll
        for (int i = 14; i < 27; i++) mid += v[i];                                               // samples ev_lo .. ev_hi - 1 = 13 .. 26
        const float volume = dh_div_const(vol, 40.0f, sps_rcp);
        if (valid) {
            const uint32_t k = k0 + q;
            dh_lds_store_row10<0>(S.var_rb + k, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]);      // transposed ring: phase-major
            dh_lds_store_row10<1000>(S.var_rb + k, v[10], v[11], v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19]);
            dh_lds_store_row10<2000>(S.var_rb + k, v[20], v[21], v[22], v[23], v[24], v[25], v[26], v[27], v[28], v[29]);
            dh_lds_store_row10<3000>(S.var_rb + k, v[30], v[31], v[32], v[33], v[34], v[35], v[36], v[37], v[38], v[39]);
            S.sum[q] = mid; S.vol_new[k] = volume;
        }
        dh_lds_stores_done();
    }
    if (SPS != 10 && SPS != 20 && SPS != 40 && DH_STOP_AFTER >= 3) DH_FOR_LANES_FRESH(lane) {
        for (uint32_t q = lane; q < m; q += DH_WAVE) {
            const uint32_t k = k0 + q;
            const uint32_t s = q * sps + (q > 0 ? (uint32_t) step_off : 0u);   // relative to p
            float sum = 0.0f, volume_sum = 0.0f;
            {
                // (the ring stores below may alias the samples as far as the compiler knows: read one by one, every sample would wait out
                // its own LDS round trip before the next is even requested)
                // four samples at a time, all four requested before the first is used; sums in sample order
                uint32_t i = 0;
                for (; i + 4u <= sps; i += 4u) {
                    float value[4];
#pragma unroll
                    for (uint32_t j = 0; j < 4u; j++) value[j] = DH_FB(s + i + j);
#pragma unroll
                    for (uint32_t j = 0; j < 4u; j++) {
                        if (i + j >= ev_lo && i + j < ev_hi) sum += value[j];
                        volume_sum += value[j];
                        S.var_rb[(i + j) * DH_VARIANCE_SYMBOLS + k] = value[j];       // transposed ring: phase-major
                    }
                }
                for (; i < sps; i++) {
                    const float value = DH_FB(s + i);
                    if (i >= ev_lo && i < ev_hi) sum += value;
                    volume_sum += value;
                    S.var_rb[i * DH_VARIANCE_SYMBOLS + k] = value;
                }
            }
            S.sum[q] = sum;
            S.vol_new[k] = dh_div_const(volume_sum, (float) sps, sps_rcp);
        }
    }
#undef DH_FB
}

// ---------------------------------------------------------------------------------------------
// ---- P6: end of a variance block -> timing decision (gfsk_demodulator.cpp:41-80)
// The reference's result depends on the per-phase variances only through (arg-min position, vmin <= 0,
// vmin > 5e6).  Its sums run in symbol order (float total, then double sum of squared deviations): a
// 100-long dependent chain on `sps` lanes.  For sps = 10 a float estimate V' with a proven error bound is
// taken first: 50 lanes = 10 phases x 5 groups of 20 symbols, partials combined through LDS.  With
// u = 2^-24, mu / sigma^2 the exact mean / variance of the phase, A = mean |x|, F(m) = sum (m - x)^2 / 100
// = sigma^2 + (m - mu)^2:
//   |mean_ref - mu| <= 100.1 u A,  |mean' - mu| <= 25.2 u A   =>  |F(mean') - F(mean_ref)| <= (100.1 u A)^2
//   (the estimate's sums run as two interleaved chains of ten per lane -- packed adds / FMAs -- plus one add, so
//   they are shorter than the 20-long chains these constants were derived for: 16 u A and 18 u below)
//   A^2 <= mean x^2 = sigma^2 + mu^2 <= 1.01 (F(mean') + 2 mean'^2)   =>  that difference < 1e-10 (V' + mean'^2)
//   V_ref = F(mean_ref)(1 + 1.2e-14);  V' = F(mean')(1 + 28 u) up to 25 subnormal roundings (< 1e-42)
// so |V' - V_ref| <= tol = 4e-6 V' + 1.2e-10 (V' + mean'^2) + 1e-42, provided nothing overflows (max |x|
// < 1e16 is checked).  If the intervals [V' - tol, V' + tol] separate the smallest phase from all others,
// from 0 and from 5e6, the reference's decision is known; an all-zero phase gives vmin = 0 exactly.
// Anything else (ties, constant input, non-finite or huge samples) is decided by the ordered chain.
// `V`: the virtual stream of the push (for the exact recomputation of the ring); k0: the run's first slot (for the exact context only).
// Returns the step the reference would take: +1, -1 or 0.
struct DhStreamView { const float* tail; uint32_t tc; const float* in; uint32_t nv; float sps_rcp; };
template <int NZ, int SPS, bool BOUNDED>
DH_HD int32_t dh_timing_decision(const DhDspParams& P, DhDspShared& S, DhBoundState* const BS, const DhStreamView& V, uint32_t sps, uint32_t k0, float e_blk) {
    const float* const tail = V.tail; const uint32_t tc = V.tc; const float* const in = V.in; const uint32_t nv = V.nv; const float sps_rcp = V.sps_rcp;
    int32_t new_off = 0;
    {
        bool ordered = true;
        // Phases that can still be the reference's arg-min once an estimate has been taken and could not decide: those whose
        // interval [V' - tol, V' + tol] reaches below the smallest upper end.  Every other phase has V_ref > that upper end
        // >= the smallest V_ref, strictly, so neither the minimum nor a tie: the ordered chain (and, in the error-bounded
        // kernels, the exact recomputation of the ring in front of it) only has to look at these rows.  All rows when no
        // valid set of intervals exists (no estimate, NaN / overflow, an estimate of exactly 0).
        uint64_t chain_rows = ~0ull;
        bool est_done = false;
        // The same estimate without its LDS exchange and with two votes instead of six.  Lane 16 r + 5 j + g takes piece g (20
        // symbols) of phase i = 3 r + j -- three phases per DPP row, so that the five partial sums of a phase are five neighbouring
        // lanes of one row and meet through row_shr (lane g = 4 of each group: ((s4 + s3) + (s2 + s1)) + s0, three roundings
        // where the exchange form has four: the tolerance derived below covers it).  The ten lanes that then hold a phase form
        // its interval; `above` (ruled out by the smallest upper end) and `fine` (guard holds, estimate not exactly 0, and the
        // lane is either ruled out or positive and below 5e6) are the only votes: one candidate and every lane fine is the case
        // decided here -- everything else (an estimate of exactly 0, NaN / overflow, ties) is left to the full form below, which
        // starts over.  ~100 instructions instead of ~260 per block.
        if (SPS == 10 && !P.ordered_timing) {
            DH_LANE_ARRAY(float, gs, 1); DH_LANE_ARRAY(float, gq, 1);
            DH_FOR_LANES_FRESH(lane) {
                const uint32_t r = (uint32_t) lane >> 4, c = (uint32_t) lane & 15u;
                const uint32_t j = (c * 205u) >> 10, g = c - 5u * j, i = 3u * r + j;      // c / 5, c % 5
                const bool act = c < 15u && i < 10u;
                const dh_f4a* row = reinterpret_cast<const dh_f4a*>(S.var_rb + (act ? i * DH_VARIANCE_SYMBOLS + g * 20u : 0u));
                dh_f2 s2 = dh_f2_make(0.0f, 0.0f), q2 = dh_f2_make(0.0f, 0.0f);
#pragma unroll
                for (int q = 0; q < 5; q++) {
                    const dh_f4a v = row[q];
                    const dh_f2 a = dh_f2_make(v.x, v.y), b = dh_f2_make(v.z, v.w);
                    s2 = dh_f2_add(s2, a); s2 = dh_f2_add(s2, b);
                    q2 = dh_f2_fma(a, a, q2); q2 = dh_f2_fma(b, b, q2);
                }
                DH_LA(gs, lane)[0] = act ? s2.x + s2.y : 0.0f; DH_LA(gq, lane)[0] = act ? q2.x + q2.y : 0.0f;
            }
            DH_LANE_ARRAY(float, lo, 1); DH_LANE_ARRAY(float, hi, 1);
            uint64_t vote_above = 0, vote_fine = 0;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
            {
                float s = gs[0], q = gq[0], t1, t2, u1, u2;
                // (row_shr with bound_ctrl: lanes without a source read 0 -- only lanes 4, 9, 14 of a row are looked at, and theirs exist)
                asm volatile("s_nop 4\n\t"               // (a VALU write of EXEC just before would need 5 wait states ahead of a DPP op)
                             "v_add_f32_dpp %0, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                             "v_add_f32_dpp %2, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                             "s_nop 0\n\t"                // (a DPP read needs two wait states after the write of its source: the other stream's add + this)
                             "v_add_f32_dpp %1, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                             "v_add_f32_dpp %3, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                             "v_add_f32_dpp %1, %4, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                             "v_add_f32_dpp %3, %5, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                             : "=&v"(t1), "=&v"(t2), "=&v"(u1), "=&v"(u2) : "v"(s), "v"(q));
                gs[0] = t2; gq[0] = u2;
            }
#else
            {
                float ts[DH_WAVE], tq[DH_WAVE];
                for (int l = 0; l < DH_WAVE; l++) {
                    const int c = l & 15;
                    auto at = [&](float (*arr)[1], int d) { return c - d >= 0 ? arr[l - d][0] : 0.0f; };
                    ts[l] = ((at(gs, 0) + at(gs, 1)) + (at(gs, 2) + at(gs, 3))) + at(gs, 4);
                    tq[l] = ((at(gq, 0) + at(gq, 1)) + (at(gq, 2) + at(gq, 3))) + at(gq, 4);
                }
                for (int l = 0; l < DH_WAVE; l++) { gs[l][0] = ts[l]; gq[l][0] = tq[l]; }
            }
#endif
            DH_FOR_LANES_FRESH(lane) {
                const uint32_t r = (uint32_t) lane >> 4, c = (uint32_t) lane & 15u;
                const bool own = (((r == 3u ? 0x0010u : 0x4210u) >> c) & 1u) != 0u;      // this lane holds a phase: lane 4, 9 or 14 of its row (row 3: phase 9 only)
                const float total = DH_LA(gs, lane)[0], e = DH_LA(gq, lane)[0] * 0.01f;      // e = mean x^2
                const float mean = total * 0.01f;
                const float v = __builtin_fmaf(-mean, mean, e);
                float tol = __builtin_fmaf(e, 4e-6f, 1e-42f);
                if (BOUNDED && e_blk > 0.0f) tol += 8.0f * e_blk * dh_sqrt_upper(__builtin_fmaxf(v, 0.0f) + 4.0f * e_blk * e_blk) + 8.0f * e_blk * e_blk;      // (see the full form below)
                const float l = own ? v - tol : DH_FLT_MAX, h = own ? v + tol : DH_FLT_MAX;
                DH_LA(lo, lane)[0] = l; DH_LA(hi, lane)[0] = h;
                DH_LA(gs, lane)[0] = own ? ((e < 1e30f && v != 0.0f && l > 0.0f && h < 4999999.0f) ? 1.0f : 0.0f) : 1.0f;      // fine as a candidate
                DH_LA(gq, lane)[0] = own ? ((e < 1e30f && v != 0.0f) ? 1.0f : 0.0f) : 1.0f;                                      // fine when ruled out
            }
            float hmin;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
            hmin = -dh_wave_max(-hi[0]);
#else
            hmin = DH_FLT_MAX;
            for (int q = 0; q < DH_WAVE; q++) hmin = dh_fmin_(hmin, hi[q][0]);
#endif
            DH_FOR_LANES_FRESH(lane) {
                const bool above = DH_LA(lo, lane)[0] > hmin;
                DH_BALLOT_ACC(vote_above, above, lane);
                DH_BALLOT_ACC(vote_fine, (above ? DH_LA(gq, lane)[0] : DH_LA(gs, lane)[0]) != 0.0f, lane);
            }
            const uint64_t cand = ~vote_above;
            if (DH_LIKELY(vote_fine == ~0ull && cand != 0 && (cand & (cand - 1)) == 0)) {
                est_done = true; ordered = false;
                const uint32_t b = (uint32_t) dh_ffs64(cand), vmin_pos = 3u * (b >> 4) + ((b & 15u) - 4u) / 5u;
                if (vmin_pos > 0 && vmin_pos < 5) new_off = +1;
                else if (vmin_pos >= 5 && vmin_pos < 9) new_off = -1;
            }
        }
        if (SPS == 10 && !P.ordered_timing && !est_done) {
            DH_BARRIER();                                   // mn / mx are dead from here: scratch
            float* psum = S.mn; float* pd = S.mx;
            // ONE pass over the ring: sum and sum of squares together, two interleaved chains per lane (packed adds / FMAs).
            // V' = Q' / 100 - mean'^2 loses more to cancellation than the reference's two passes, and its tolerance says
            // so: with S', Q' the float sums (chains of 10 + 1 + 4 roundings: 15 u), 0.01f for 1 / 100 (0.4 u) and m2 =
            // mean x^2 = sigma^2 + mu^2 >= A^2:
            //   |Q' 0.01f - m2| <= 16.4 u m2,  |mean' - mu| <= 16.4 u A  =>  |mean'^2 - mu^2| <= 32.8 u m2,  the final fma 1 u m2
            //   |V' - sigma^2| <= 50.2 u m2 = 3.0e-6 m2;  V_ref within (100.1 u A)^2 + 1.2e-14 V of sigma^2: 3.6e-11 m2
            // so tol = 4e-6 m2' + 1e-42 (m2' = Q' 0.01f, within 1e-6 of m2; 1e-42 for up to ~120 subnormal roundings).
            // A strong DC component (mu^2 >> sigma^2) widens the intervals; what they cannot separate goes to the ordered
            // chain as before.
            DH_FOR_LANES_FRESH(lane) {
                if (lane < 50) {
                    // lane = 5 i + g: eight consecutive lanes then read 16-byte pieces 20 or 40 words apart, which fall into
                    // eight different groups of four banks (with lane = 10 g + i two of every eight collided: 50 LDS cycles
                    // per block, all of this phase's bank conflicts)
                    const uint32_t i = ((uint32_t) lane * 205u) >> 10, g = (uint32_t) lane - 5u * i;      // lane / 5, lane % 5
                    const uint32_t ro = i * DH_VARIANCE_SYMBOLS + g * 20u;
                    const dh_f4a* row = reinterpret_cast<const dh_f4a*>(S.var_rb + ro);
                    dh_f2 s2 = dh_f2_make(0.0f, 0.0f), q2 = dh_f2_make(0.0f, 0.0f);
#pragma unroll
                    for (int q = 0; q < 5; q++) {
                        const dh_f4a v = row[q];
                        const dh_f2 a = dh_f2_make(v.x, v.y), b = dh_f2_make(v.z, v.w);
                        s2 = dh_f2_add(s2, a); s2 = dh_f2_add(s2, b);
                        q2 = dh_f2_fma(a, a, q2); q2 = dh_f2_fma(b, b, q2);
                    }
                    psum[lane] = s2.x + s2.y; pd[lane] = q2.x + q2.y;
                }
            }
            DH_BARRIER();
            // lanes 0..9 hold one phase each; the others hold neutral values
            DH_LANE_ARRAY(float, lo, 1); DH_LANE_ARRAY(float, hi, 1);
            uint64_t vote_guard = 0, vote_vzero = 0, vote_zero = 0, vote_pos = 0, vote_small = 0, vote_above = 0;
            DH_FOR_LANES_FRESH(lane) {
                float l = DH_FLT_MAX, h = DH_FLT_MAX;
                bool guard = true, vzero = false;
                if (lane < 10) {
                    const int i = lane;
                    const float total = (((psum[5 * i] + psum[5 * i + 1]) + psum[5 * i + 2]) + psum[5 * i + 3]) + psum[5 * i + 4];
                    const float e = ((((pd[5 * i] + pd[5 * i + 1]) + pd[5 * i + 2]) + pd[5 * i + 3]) + pd[5 * i + 4]) * 0.01f;      // mean x^2
                    const float mean = total * 0.01f;
                    const float v = __builtin_fmaf(-mean, mean, e);
                    float tol = __builtin_fmaf(e, 4e-6f, 1e-42f);
                    // error-bounded mode: the ring holds values within e_blk of the reference's; moving every sample by
                    // up to e_blk moves the mean by <= e_blk, every deviation by <= 2 e_blk and the variance by
                    // <= 4 e_blk sqrt(V) + 4 e_blk^2 (Cauchy-Schwarz); taken twice over for the float mean's own rounding
                    if (BOUNDED && e_blk > 0.0f) tol += 8.0f * e_blk * dh_sqrt_upper(__builtin_fmaxf(v, 0.0f) + 4.0f * e_blk * e_blk) + 8.0f * e_blk * e_blk;
                    guard = e < 1e30f;                           // false for NaN, and for samples beyond ~1e15 (e overflows first)
                    vzero = v == 0.0f;
                    l = v - tol; h = v + tol;
                }
                DH_LA(lo, lane)[0] = l; DH_LA(hi, lane)[0] = h;
                DH_BALLOT_ACC(vote_guard, guard, lane);
                DH_BALLOT_ACC(vote_vzero, vzero, lane);
                DH_BALLOT_ACC(vote_pos, l > 0.0f, lane);
                DH_BALLOT_ACC(vote_small, h < 4999999.0f, lane);
            }
            const uint32_t ten = 0x3FFu;
            if (((uint32_t) vote_vzero & ten) && !(BOUNDED && e_blk > 0.0f)) {       // (approximate samples prove nothing about exact zeros)
                // an estimate of exactly 0: only a phase whose hundred samples are all (+-)0 has vmin == 0 for sure (tiny
                // samples square to 0 in float, not in the reference's double) -- look at the bits
                uint64_t vote_nz = 0;
                DH_FOR_LANES_FRESH(lane) {
                    uint32_t any = 0;
                    if (lane < 50) {
                        const uint32_t g = ((uint32_t) lane * 205u) >> 11, i = (uint32_t) lane - 10u * g;
                        const uint32_t* row = reinterpret_cast<const uint32_t*>(S.var_rb + i * DH_VARIANCE_SYMBOLS + g * 20u);
#pragma unroll
                        for (int q = 0; q < 20; q++) any |= row[q];
                    }
                    DH_BALLOT_ACC(vote_nz, (any << 1) != 0u, lane);
                }
                const uint32_t nz = (uint32_t) (vote_nz | (vote_nz >> 10) | (vote_nz >> 20) | (vote_nz >> 30) | (vote_nz >> 40));
                vote_zero = ~nz & ten;
            }
            float hmin;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
            hmin = dh_row_min_to_lane15(hi[0]);
#else
            hmin = DH_FLT_MAX;
            for (int q = 0; q < 10; q++) hmin = dh_fmin_(hmin, hi[q][0]);
#endif
            DH_FOR_LANES_FRESH(lane) { DH_BALLOT_ACC(vote_above, DH_LA(lo, lane)[0] > hmin, lane); }
            const uint32_t above = (uint32_t) vote_above & ten;
            const uint32_t cand = ~above & ten;                 // phases whose interval reaches below hmin
            if (((uint32_t) vote_guard & ten) != ten) {
            } else if ((uint32_t) vote_zero & ten) {
                ordered = false;                                // vmin == 0 exactly: no step
            } else if (dh_popc32(cand) == 1 && ((uint32_t) vote_pos & cand) && ((uint32_t) vote_small & cand)) {
                ordered = false;
                const uint32_t vmin_pos = (uint32_t) dh_ffs64((uint64_t) cand);
                if (vmin_pos > 0 && vmin_pos < 5) new_off = +1;
                else if (vmin_pos >= 5 && vmin_pos < 9) new_off = -1;
            }
            // (the candidates are not handed on here: one more live value costs the 81-tap kernels four scratch accesses in
            // their hot loop, and they reach this point in 0.02 % of the blocks -- the 161-tap, sps-20 kernel in 1.3 %)
        }
        if (ordered && SPS != 10 && sps <= DH_WAVE && !P.ordered_timing && (!BOUNDED || P.exact_mode == 0)) {
            // (sps 33 .. 64, e.g. POCSAG's 40: one lane per phase, chains of 100 + 1 fused terms -- (1 + u)^102 - 1 < 6.1e-6, still
            // inside the 8e-6 V' below; the float mean is then the reference's own chain, within 100.1 u A of the true one.  The
            // in-order double chain cost that slicer a fifth of its time, every block.)
            // Run-time sps with at least one lane per phase: a float estimate like the sps-10 one.  G = 64 / sps groups
            // per phase, lane g sps + i takes a contiguous piece of row i -- `seg` = 4 ceil(25 / G) ring entries, read 16
            // bytes at a time -- and the partial sums meet in LDS.  The bound of the sps-10 estimate holds with 8e-6 V'
            // for 4e-6 V' (chains of up to 52 fused terms + G partials); in the error-bounded kernels the ring holds
            // values within e_blk of the reference's, which moves a variance by |2 cov(x, d) + var(d)| <= 2 e sigma +
            // e^2 with sigma <= sqrt(V) + e, i.e. <= 2 e sqrt(V) + 3 e^2 (taken as 2.5 e sqrt(V) + 4 e^2).  An estimate
            // of exactly 0, a NaN or an overflow is left to the chain below.
            DH_BARRIER();                                   // mn / mx are dead from here: scratch
            float* psum = S.mn; float* pd = S.mx; float* pmean = S.mn + 64;
            const uint32_t G = DH_WAVE / sps, active = G * sps, quads = (25u + G - 1u) / G;
            for (int pass = 0; pass < 2; pass++) {
                DH_FOR_LANES_FRESH(lane) {
                    if ((uint32_t) lane < active) {
                        const uint32_t g = (uint32_t) lane / sps, i = (uint32_t) lane - g * sps;
                        const uint32_t q0 = g * quads;                                 // first 16-byte piece of this lane
                        const dh_f4a* row = reinterpret_cast<const dh_f4a*>(S.var_rb + i * DH_VARIANCE_SYMBOLS);
                        float mean = 0.0f;
                        if (pass == 1) {
                            float total = 0.0f;
                            for (uint32_t gg = 0; gg < G; gg++) total += psum[gg * sps + i];
                            mean = total * 0.01f;
                            if (g == 0) pmean[i] = mean;
                        }
                        float acc = 0.0f;
                        for (uint32_t q = 0; q < quads; q += 3u) {                     // three pieces in flight
                            dh_f4a v[3];
#pragma unroll
                            for (uint32_t j = 0; j < 3u; j++) v[j] = row[dh_min<uint32_t>(q0 + q + j, 24u)];
#pragma unroll
                            for (uint32_t j = 0; j < 3u; j++) {
                                if (q + j < quads && q0 + q + j < 25u) {
                                    if (pass == 0) { acc += v[j].x; acc += v[j].y; acc += v[j].z; acc += v[j].w; }
                                    else {
                                        const float d0 = mean - v[j].x, d1 = mean - v[j].y, d2 = mean - v[j].z, d3 = mean - v[j].w;
                                        acc = __builtin_fmaf(d0, d0, acc); acc = __builtin_fmaf(d1, d1, acc);
                                        acc = __builtin_fmaf(d2, d2, acc); acc = __builtin_fmaf(d3, d3, acc);
                                    }
                                }
                            }
                        }
                        if (pass == 0) psum[lane] = acc; else pd[lane] = acc;
                    }
                }
                DH_BARRIER();
            }
            DH_LANE_ARRAY(float, iv_lo, 1); DH_LANE_ARRAY(float, iv_hi, 1);
            uint64_t v_ok = 0, v_pos = 0, v_small = 0, v_big = 0, v_above = 0;
            DH_FOR_LANES_FRESH(lane) {
                float l = DH_FLT_MAX, h = DH_FLT_MAX;
                bool ok = true;
                if ((uint32_t) lane < sps) {
                    float v = 0.0f;
                    for (uint32_t gg = 0; gg < G; gg++) v += pd[gg * sps + (uint32_t) lane];
                    v *= 0.01f;
                    const float mean = pmean[lane];
                    const float e = __builtin_fmaf(mean, mean, v);
                    float tol = __builtin_fmaf(v, 8e-6f, __builtin_fmaf(e, 1.2e-10f, 1e-42f));
                    if (BOUNDED && e_blk > 0.0f) tol += 2.5f * e_blk * __builtin_sqrtf(v) + 4.0f * e_blk * e_blk;
                    ok = e < 1e30f && v != 0.0f;
                    l = v - tol; h = v + tol;
                }
                DH_LA(iv_lo, lane)[0] = l; DH_LA(iv_hi, lane)[0] = h;
                DH_BALLOT_ACC(v_ok, ok, lane);
                DH_BALLOT_ACC(v_pos, l > 0.0f, lane);
                DH_BALLOT_ACC(v_small, h < 4999999.0f, lane);
                DH_BALLOT_ACC(v_big, l > 5000001.0f, lane);
            }
            float hmin;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
            hmin = -dh_wave_max(-iv_hi[0]);
#else
            hmin = DH_FLT_MAX;
            for (int q = 0; q < DH_WAVE; q++) hmin = dh_fmin_(hmin, iv_hi[q][0]);
#endif
            DH_FOR_LANES_FRESH(lane) { DH_BALLOT_ACC(v_above, DH_LA(iv_lo, lane)[0] > hmin, lane); }
            const uint64_t phases = sps >= 64u ? ~0ull : ((1ull << sps) - 1ull);
            const uint64_t cand = ~v_above & phases;                          // phases whose interval reaches below the smallest upper end
            const bool one = cand != 0 && (cand & (cand - 1)) == 0;
            if (one && (v_ok & phases) == phases && (((v_pos & cand) && (v_small & cand)) || (v_big & cand))) {
                ordered = false;
                const uint32_t vmin_pos = (uint32_t) dh_ffs64(cand);
                if (v_big & cand) {
                } else if (vmin_pos > 0 && vmin_pos < sps / 2) new_off = +1;
                else if (vmin_pos >= sps / 2 && vmin_pos < sps - 1) new_off = -1;
            } else if (cand != 0 && (v_ok & phases) == phases) {
                chain_rows = cand;
            }
        }
        if (DH_UNLIKELY(ordered)) {
            // both sums of a phase in symbol order: one phase per lane, its 100 samples contiguous in the transposed
            // ring and fetched 16 bytes at a time.
            // Error-bounded kernels: the ring holds values within e_blk of the reference's.  At sps 10 the estimate above
            // has already failed, so the chain runs on the reference's samples -- all thousand of this block, recomputed
            // exactly.  At a run-time sps there is no estimate: the chain first runs on the ring as it is (attempt 0) and
            // its result stands if the intervals [V - tol, V + tol] separate the smallest phase from all others, from 0 and
            // from 5e6; otherwise attempt 1 recomputes the ring exactly.  tol: samples moved by <= e move the variance
            // by |2 cov(x, d) + var(d)| <= 2 e sigma + e^2 with sigma <= sqrt(V) + e, i.e. <= 2 e sqrt(V) + 3 e^2 (taken
            // as 2.5 e sqrt(V) + 4 e^2); each chain's float mean is within 100.1 u A of the true one, which moves its
            // sum of squared deviations by (that)^2 <= 1e-10 (V + mean^2); the double arithmetic adds 1e-14 V.
            const bool approx_ring = BOUNDED && e_blk > 0.0f;
            for (int attempt = (approx_ring && SPS != 10 && sps > DH_WAVE && !P.ordered_timing && P.exact_mode == 0) ? 0 : 1; attempt < 2; attempt++) {      // (attempt 0 only where the estimate above does not run)
                DH_BARRIER();
                if (approx_ring && attempt == 1) {
                    DhExactCtx C;
                    C.tail = tail; C.tc = tc; C.in = in; C.nv = nv; C.tapsf = S.tapsf; C.gain = P.gain; C.rgain = P.rgain; C.sps_rcp = sps_rcp;
                    C.cur_start = (int32_t) dh_uniform((uint32_t) BS->cur_start); C.cur_off = (int32_t) dh_uniform((uint32_t) BS->cur_off);
                    C.prev_start = 0; C.prev_off = 0; C.blk_flags = dh_uniform(BS->blk_flags);
                    C.k0 = k0; C.e_eff = 0.0f; C.levels = P.levels; C.invert = P.invert;
                    // (the window block is dead here except words 512..575, where the L2 touch of the next window may
                    // still be dropping its dwords: the staged variant uses the words behind them)
                    dh_exact_var_ring_staged<NZ>(C, S, sps, S.xf + 576, dh_dsp_xf_words(NZ) - 576u, chain_rows);
                    BS->n_exact_blocks++;
                    DH_BARRIER();
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
#endif
                }
                DH_LANE_ARRAY(float, iv_lo, 1); DH_LANE_ARRAY(float, iv_hi, 1); DH_LANE_ARRAY(uint32_t, iv_ok, 1);
                DH_FOR_LANES_FRESH(lane) {
                    if ((uint32_t) lane < sps && !((chain_rows >> (uint32_t) lane) & 1ull)) {
                        S.variance[lane] = DH_DBL_MAX;               // ruled out by its interval: never the smallest
                        if (BOUNDED && SPS != 10 && attempt == 0) { DH_LA(iv_lo, lane)[0] = DH_FLT_MAX; DH_LA(iv_hi, lane)[0] = DH_FLT_MAX; DH_LA(iv_ok, lane)[0] = 1u; }
                    } else if ((uint32_t) lane < sps) {
                        const dh_f4a* row = reinterpret_cast<const dh_f4a*>(S.var_rb + lane * DH_VARIANCE_SYMBOLS);
                        float total = 0.0f;
#pragma unroll 5
                        for (int q = 0; q < DH_VARIANCE_SYMBOLS / 4; q++) {
                            const dh_f4a v = row[q];
                            total += v.x; total += v.y; total += v.z; total += v.w;
                        }
                        const double mean = (double) (total / (float) DH_VARIANCE_SYMBOLS);
                        double dsum = 0.0;
#pragma unroll 5
                        for (int q = 0; q < DH_VARIANCE_SYMBOLS / 4; q++) {
                            const dh_f4a v = row[q];
                            const double d0 = mean - (double) v.x, d1 = mean - (double) v.y, d2 = mean - (double) v.z, d3 = mean - (double) v.w;
                            const double s0 = d0 * d0, s1 = d1 * d1, s2 = d2 * d2, s3 = d3 * d3;
                            dsum += s0; dsum += s1; dsum += s2; dsum += s3;
                        }
                        const double var = dsum / (double) DH_VARIANCE_SYMBOLS;
                        S.variance[lane] = var;
                        if (BOUNDED && SPS != 10 && attempt == 0) {
                            // this phase's interval, as floats rounded outwards (2e-7 V covers the two conversions)
                            const double eb = (double) e_blk;
                            const double tol = 2.5 * eb * __builtin_sqrt(var) + 4.0 * eb * eb + 2e-10 * (var + mean * mean) + 2e-7 * var + 1e-40;
                            DH_LA(iv_lo, lane)[0] = (float) (var - tol); DH_LA(iv_hi, lane)[0] = (float) (var + tol);
                            DH_LA(iv_ok, lane)[0] = (var + mean * mean < 1e30) ? 1u : 0u;      // false for NaN / overflow, as in the estimate
                        }
                    } else if (BOUNDED && SPS != 10 && attempt == 0) {
                        DH_LA(iv_lo, lane)[0] = DH_FLT_MAX; DH_LA(iv_hi, lane)[0] = DH_FLT_MAX; DH_LA(iv_ok, lane)[0] = 1u;
                    }
                }
                DH_BARRIER();
                if (BOUNDED && SPS != 10 && attempt == 0) {
                    // is the reference's (arg-min, vmin <= 0, vmin > 5e6) beyond doubt?  One vote per question.
                    float hmin;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
                    hmin = -dh_wave_max(-iv_hi[0]);
#else
                    hmin = DH_FLT_MAX;
                    for (int q = 0; q < DH_WAVE; q++) hmin = dh_fmin_(hmin, iv_hi[q][0]);
#endif
                    uint64_t v_above = 0, v_ok = 0, v_pos = 0, v_small = 0, v_big = 0;
                    DH_FOR_LANES_FRESH(lane) {
                        const float l = DH_LA(iv_lo, lane)[0], h = DH_LA(iv_hi, lane)[0];
                        DH_BALLOT_ACC(v_above, l > hmin, lane);
                        DH_BALLOT_ACC(v_ok, DH_LA(iv_ok, lane)[0] != 0u, lane);
                        DH_BALLOT_ACC(v_pos, l > 0.0f, lane);
                        DH_BALLOT_ACC(v_small, h < 4999999.0f, lane);
                        DH_BALLOT_ACC(v_big, l > 5000001.0f, lane);
                    }
                    const uint64_t phases = sps >= 64u ? ~0ull : ((1ull << sps) - 1ull);
                    const uint64_t cand = ~v_above & phases;                  // phases whose interval reaches below the smallest upper end
                    const bool one = cand != 0 && (cand & (cand - 1)) == 0;
                    const bool sure = one && (v_ok & phases) == phases && (((v_pos & cand) && (v_small & cand)) || (v_big & cand));
                    if (!sure) continue;                                      // attempt 1: the exact ring
                    DH_FOR_LANES_FRESH(lane) { if (DH_IS_LANE0(lane)) S.stats[1]++; }
                    const uint32_t vmin_pos = (uint32_t) dh_ffs64(cand);
                    if (v_big & cand) {
                    } else if (vmin_pos > 0 && vmin_pos < sps / 2) new_off = +1;
                    else if (vmin_pos >= sps / 2 && vmin_pos < sps - 1) new_off = -1;
                    break;
                }
                double vmin = S.variance[0]; uint32_t vmin_pos = 0;
                for (uint32_t i = 1; i < sps; i++) if (S.variance[i] < vmin) { vmin = S.variance[i]; vmin_pos = i; }
                DH_FOR_LANES_FRESH(lane) { if (DH_IS_LANE0(lane)) S.stats[1]++; }
                if (vmin <= 0 || vmin > 5000000) {
                } else if (vmin_pos > 0 && vmin_pos < sps / 2) new_off = +1;
                else if (vmin_pos >= sps / 2 && vmin_pos < sps - 1) new_off = -1;
                break;
            }
        }
    }
    return new_off;
}

// ---------------------------------------------------------------------------------------------
// ---- P5: thresholds + slice (gfsk_demodulator.cpp:88-106 / fsk_demodulator.cpp:89-99)
// error-bounded mode: a comparison whose two sides are closer than T cannot be trusted to come out as the
// reference's; those symbols are not stored here but decided exactly below
// TWO symbols per lane, q = 2 lane and 2 lane + 1 (a run has at most 100): their AGC extremes, window sums and everything
// derived from them are pairs, so centre, average, the two thresholds and the three distances are packed operations for
// both symbols -- one pass of ~40 vector instructions instead of two of ~37.  Every operation is the one the one-symbol
// form performs (same operands, same order), so the dibits and the doubts are the same.
// `out`: where symbol 0 of the run goes; `agc` (device): the scan's own results for the lane's two symbols, used when the run starts its
// block.  Doubtful symbols are not stored; their votes come back in vote_a (symbols 2 l) and vote_b (2 l + 1).
struct DhSliceSetup { bool four_levels, e_pos, width_pow2, pair_store; float T_eff, inv_width; };
#if !(DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__))
struct DhAgcPair { float mn0, mx0, mn1, mx1; };          // (harness: the scan leaves its results in S.mn / S.mx only)
#endif
template <int SPS, int LV, bool BOUNDED>
DH_HD void dh_slice_symbols(const DhDspParams& P, DhDspShared& S, const DhAgcPair agc, uint8_t* out, uint32_t k0, uint32_t m, uint32_t ev_lo, uint32_t ev_hi,
                            const DhSliceSetup& U, uint64_t& vote_a, uint64_t& vote_b) {
    (void) agc;
    const bool four_levels = U.four_levels, e_pos = U.e_pos, width_pow2 = U.width_pow2, pair_store = U.pair_store;
    const float T_eff = U.T_eff, inv_width = U.inv_width;
    DH_FOR_LANES_FRESH(lane) {
        const uint32_t qa = 2u * (uint32_t) lane;
        const bool va = qa < m, vb = qa + 1u < m;
        const uint32_t qq = va ? qa : 0u, k = k0 + qq;                            // (lanes beyond the run recompute symbols 0 / 1: in-range reads, no store)
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
        // (a run that starts its block: the lane's symbols 2 l, 2 l + 1 are the ring slots the AGC scan left in its registers)
        dh_f2 mn, mx;
        if (k0 == 0u) { mn = dh_f2_make(agc.mn0, agc.mn1); mx = dh_f2_make(agc.mx0, agc.mx1); }
        else { mn = dh_f2_make(S.mn[k], S.mn[k + 1u]); mx = dh_f2_make(S.mx[k], S.mx[k + 1u]); }
#else
        const dh_f2 mn = dh_f2_make(S.mn[k], S.mn[k + 1u]), mx = dh_f2_make(S.mx[k], S.mx[k + 1u]);
#endif
        const dh_f2 sumq = dh_f2_make(S.sum[qq], S.sum[qq + 1u]);
        const dh_f2 center = dh_f2_scale(dh_f2_add(mx, mn), 0.5f);                // (max + min) / 2.0f: the division by two is exact
        // (sps 20: the division by the window's six samples as reciprocal product + exact residual + correction, dh_div_const; tests/test_numerics.py)
        const dh_f2 average = width_pow2 ? dh_f2_scale(sumq, inv_width) : SPS == 20 ? dh_div_const2(sumq, 6.0f, inv_width)
                                                                                    : dh_f2_make(sumq.x / (float) (ev_hi - ev_lo), sumq.y / (float) (ev_hi - ev_lo));
        const dh_f2 c625 = dh_f2_make(0.625f, 0.625f);
        const dh_f2 umid = dh_f2_fma(dh_f2_sub(mx, center), c625, center);        // one float FMA each: see the one-symbol form below
        const dh_f2 lmid = dh_f2_fma(dh_f2_sub(mn, center), c625, center);
        const bool above_a = average.x > center.x, above_b = average.y > center.y;
        const uint8_t sym4a = above_a ? (average.x > umid.x ? 1 : 0) : (average.x < lmid.x ? 3 : 2);
        const uint8_t sym4b = above_b ? (average.y > umid.y ? 1 : 0) : (average.y < lmid.y ? 3 : 2);
        const uint8_t sym2a = LV == 4 ? (uint8_t) 0 : above_a ? (uint8_t) !P.invert : (uint8_t) (P.invert != 0);
        const uint8_t sym2b = LV == 4 ? (uint8_t) 0 : above_b ? (uint8_t) !P.invert : (uint8_t) (P.invert != 0);
        const uint8_t sa = four_levels ? sym4a : sym2a, sb = four_levels ? sym4b : sym2b;
        bool doubt_a = false, doubt_b = false;
        if (BOUNDED) {
            const dh_f2 du = dh_f2_sub(average, umid), dl = dh_f2_sub(average, lmid), dc = dh_f2_sub(average, center);
            const float da = dh_min3_abs(four_levels ? du.x : DH_FLT_MAX, four_levels ? dl.x : DH_FLT_MAX, dc.x);
            const float db = dh_min3_abs(four_levels ? du.y : DH_FLT_MAX, four_levels ? dl.y : DH_FLT_MAX, dc.y);
            doubt_a = va && e_pos && !(da > T_eff);                               // (!(d > T): a NaN distance is a doubt)
            doubt_b = vb && e_pos && !(db > T_eff);
        }
#ifdef DH_IGNORE_DOUBT
        doubt_a = false; doubt_b = false;
#endif
        // the lane's two dibits are neighbours: one 16-bit store when the row position is even (wave-uniform) and both are decided
        if (pair_store && va && !doubt_a && vb && !doubt_b) *reinterpret_cast<uint16_t*>(out + qa) = (uint16_t) ((uint32_t) sa | ((uint32_t) sb << 8));
        else {
            if (va && !doubt_a) out[qa] = sa;
            if (vb && !doubt_b) out[qa + 1u] = sb;
        }
        DH_BALLOT_ACC(vote_a, doubt_a, lane);
        DH_BALLOT_ACC(vote_b, doubt_b, lane);
    }
}

// the doubtful symbols of a run (votes of dh_slice_symbols), one exact evaluation at a time -- the even symbols' votes first (the evaluations do
// not depend on one another); ONE call site: the evaluation is a thousand instructions of text
template <int NZ>
DH_COLD void dh_settle_doubts(const DhDspParams& P, DhDspShared& S, DhBoundState* const BS, const DhStreamView& V, uint8_t* out, uint32_t k0, float e_eff,
                              uint32_t sps, uint32_t ev_lo, uint32_t ev_hi, uint64_t vote_a, uint64_t vote_b) {
    const float* const tail = V.tail; const uint32_t tc = V.tc; const float* const in = V.in; const uint32_t nv = V.nv; const float sps_rcp = V.sps_rcp;
        DhExactCtx C;
        C.tail = tail; C.tc = tc; C.in = in; C.nv = nv; C.tapsf = S.tapsf; C.gain = P.gain; C.rgain = P.rgain; C.sps_rcp = sps_rcp;
        C.cur_start = (int32_t) dh_uniform((uint32_t) BS->cur_start); C.cur_off = (int32_t) dh_uniform((uint32_t) BS->cur_off);
        C.prev_start = (int32_t) dh_uniform((uint32_t) BS->prev_start); C.prev_off = (int32_t) dh_uniform((uint32_t) BS->prev_off);
        C.blk_flags = dh_uniform(BS->blk_flags);
        C.k0 = k0; C.e_eff = e_eff; C.levels = P.levels; C.invert = P.invert;
        // one evaluation at a time, the even symbols' votes first (the evaluations do not depend on one another); ONE call site:
        // the evaluation is a thousand instructions of text
        uint64_t ta = vote_a, tb = vote_b;
        while ((ta | tb) != 0) {
            uint32_t q;
            if (ta) { q = 2u * (uint32_t) dh_ffs64(ta); ta &= ta - 1; } else { q = 2u * (uint32_t) dh_ffs64(tb) + 1u; tb &= tb - 1; }
            // (the raw samples behind each evaluation are staged through the dead part of the window block: fetched one by one from
            // HBM / L2 by the 81-tap chains, a doubtful symbol cost as much as several whole runs)
            const uint8_t sym = dh_exact_symbol_staged<NZ>(C, S, k0 + q, S.xf + 576, dh_dsp_xf_words(NZ) - 576u, sps, ev_lo, ev_hi);
            DH_FOR_LANES_FRESH(lane) { if (DH_IS_LANE0(lane)) out[q] = sym; }
            BS->n_uncertain++;
        }
}

// ---------------------------------------------------------------------------------------------
// One channel, one push.  `S` is this wavefront's LDS block.  Called by all 64 lanes (device) or
// once (host harness; the DH_FOR_LANES loops then iterate the lanes).
// SPS = 10 bakes the DMR / YSF samples-per-symbol (and its evaluation window 3..6) into the code so the
// per-symbol loops unroll; SPS = 0 takes them from the parameters.
// part_lo / part_hi / sym_base: the tail split of the chain kernels -- this call takes the samples [part_lo, part_hi) of the
// push and appends its symbols behind the sym_base symbols the earlier parts produced (a part is a push: results do not
// depend on where pushes end).
// LV = 4 / 2: the number of levels is known where the kernel is instantiated (the chain kernels: 4 for DMR / YSF / NXDN, 2 for
// D-Star; engine.hip checks P.levels against it) and the other slicer's selects, its invert mask and two scalar register
// pairs drop out of the slicing phase; 0 = taken from P.levels.
// KEEPF (DH_FLAG_KEEP_FILTERED | DH_FLAG_ONE_LAUNCH): the run's filtered samples also leave, from where they stand in LDS -- 1: those of the
// split-f16 product (2.5e-6 of the reference's); 2 (with DH_FLAG_FAST_FIR): the error-bounded kernel filters with the f32 FMA chain instead
// (v_mfma_f32_16x16x4_f32: the arithmetic of DH_FLAG_FAST_FIR, 1e-6 -- BASELINE configs[1] in ONE launch, dibits still the reference's).
template <int NZ, bool FAST, int SPS, int LV = 0, int KEEPF = 0>
DH_HD void dh_rrc_demod_channel(const DhDspParams& P, uint32_t ch, DhDspShared& S, uint32_t part_lo = 0, uint32_t part_hi = 0xFFFFFFFFu, uint32_t sym_base = 0) {
    static_assert(SPS == 0 || SPS == 10 || SPS == 20 || SPS == 40, "0 = run-time samples per symbol; 10 has its own window and timing code, 20 / 40 are the generic code with the constant folded in");
    constexpr bool BOUNDED = DhIsBounded<NZ, FAST, SPS>::value;                      // see "Error-bounded FIR" above
    constexpr bool MF16 = DH_FIR_F16 && BOUNDED && (NZ == 80 || NZ == 160) && KEEPF != 2;           // its fused FIR as a split-f16 product on the matrix cores
    DhBoundState* const BS = DH_BOUND_STATE(S);
    float* st = P.state + (size_t) ch * P.state_stride;
    uint32_t* sth = (uint32_t*) st;
    const uint32_t sps = SPS ? (uint32_t) SPS : P.sps;
    const uint32_t ev_lo = SPS == 10 ? 3u : SPS == 20 ? 7u : SPS == 40 ? 13u : P.lo, ev_hi = SPS == 10 ? 7u : SPS == 20 ? 13u : SPS == 40 ? 27u : P.hi;        // (round(sps / 3), round(2 sps / 3): engine_impl.hpp)
    const float sps_rcp = 1.0f / (float) sps;           // correctly rounded (IEEE division, once per push)
    float* tail = st + DH_ST_VAR + DH_VARIANCE_SYMBOLS * sps;
    const float* in = P.in + (size_t) ch * P.in_stride + part_lo;
    uint8_t* syms = P.syms + (size_t) ch * P.sym_stride;
    const float* in_end = P.in + (size_t) (P.n_channels - 1u) * P.in_stride + P.n;    // end of the readable input

    // raw-sample window layout: padded one word per 16 for the FIR's lane-strided reads; plain without an RRC stage
    // (the samples are then used where they were staged, and the window sums sit right behind them)
#define DH_XP(e) (NZ > 0 ? DH_XPAD(e) : (e))
    DH_PHASE_MARK_BEGIN();
    // ---- load carried state
    uint32_t k0 = sth[DH_ST_K];
    int32_t off = (int32_t) sth[DH_ST_OFF];
    const uint32_t tc = sth[DH_ST_TAIL];
    const uint32_t n_push = P.n_per ? dh_min<uint32_t>(dh_uniform(P.n_per[ch]), P.n) : P.n;
    const uint32_t n_new = dh_min<uint32_t>(n_push, part_hi) > part_lo ? dh_min<uint32_t>(n_push, part_hi) - part_lo : 0u;
    const uint32_t nv = tc + n_new;                     // length of the virtual input stream
    const uint32_t nf = nv >= NZ ? nv - NZ : 0u;        // filtered samples available this push
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < DH_SCAN_N; j += DH_WAVE) {
            S.vol_old[j] = j < DH_VOLUME_RB_SIZE ? st[DH_ST_VOL + j] : 0.0f;
            S.vol_new[j] = 0.0f;
        }
        for (uint32_t j = lane; j < DH_VARIANCE_SYMBOLS * sps; j += DH_WAVE) S.var_rb[j] = st[DH_ST_VAR + j];
    }
    DH_BARRIER();

    // FIR taps are parked in LDS and pulled into VGPRs at the start of every FIR pass (see P2): their live range
    // must end with the FIR, or they pin ~80 registers through the latency-bound phases where the prefetch lives.
    if (BOUNDED) {
        BS->cur_start = (int32_t) sth[DH_ST_CUR_START]; BS->cur_off = (int32_t) sth[DH_ST_CUR_OFF];
        BS->prev_start = (int32_t) sth[DH_ST_PREV_START]; BS->prev_off = (int32_t) sth[DH_ST_PREV_OFF];
        BS->blk_flags = sth[DH_ST_BLOCK_FLAGS]; BS->e_count = sth[DH_ST_E_COUNT]; BS->n_uncertain = 0; BS->n_exact_runs = 0; BS->n_exact_blocks = 0;
        BS->e_cur = st[DH_ST_E_CUR]; BS->e_prev = st[DH_ST_E_PREV]; BS->e_blk = st[DH_ST_E_BLOCK];
    }
    DH_FOR_LANES(lane) {
        if (NZ > 80) { for (int i = lane; i <= NZ / 2; i += DH_WAVE) S.tapsf[i] = P.taps[i]; }
        else if (NZ > 0) {                              // the whole response between two runs of zeros (see DhDspShared)
            for (int j = lane; j < NZ + 1 + 2 * DH_TAP_LEAD; j += DH_WAVE) {
                const int k = j - DH_TAP_LEAD;
                S.tapsf[k] = (k >= 0 && k <= NZ) ? P.taps[k <= NZ / 2 ? k : NZ - k] : 0.0f;
            }
        }
        if (lane < 2) S.stats[lane] = 0;
    }

    // error-bounded mode (see DH_BOUNDED_FIR above): the tail then starts with DH_ST_P0 samples of HISTORY, so that the
    // raw samples behind every entry of the 100-symbol rings are still at hand when a comparison has to be settled exactly
    uint32_t p = BOUNDED ? sth[DH_ST_P0] : 0u;          // read position in the filtered stream
    bool staged = false; uint32_t staged_p = 0;         // the LDS window already holds V[staged_p ...) (prefetch)
    // the next window fetched into registers behind P3 and put into the window block in P7: the split-f16 kernels (as two arrays of
    // halves, with the constants of its run below) and the kernels without an RRC stage (as it is: they have the registers, and staged
    // from L2 at the start of the next run every run waited out an L2 round trip: POCSAG slicer 4.86 -> 4.34 ms, D-Star chain 3.94 -> 3.65,
    // r05_a_ab_logs.txt pf1)
    constexpr bool PF_PLAIN = DH_PF_REG && NZ == 0 && !BOUNDED && !KEEPF;
    constexpr bool PF_REG = (MF16 && DH_PF_REG) || PF_PLAIN;
    float st_e_run = 0.0f, st_k1 = 0.0f, st_k2 = 0.0f;
    // split-f16 FIR: the window held in `varr` (five 16-byte groups per lane, as loaded) -> zeros beyond `have`, max |x| of the
    // wavefront, the power-of-two scale that puts it into [0.5, 1), the two arrays of halves in the window block.  Returns
    // false (nothing stored) when max |x| is outside the range the bound covers.
    auto stage_f16 = [&](auto& varr, uint32_t have, float& e_out, float& k1_out, float& k2_out) __attribute__((always_inline)) -> bool {
        constexpr uint32_t LAST_LANES = (DH_FTILE + NZ - 4u * DH_WAVE * (DH_PF_N - 1)) / 4u;
        DH_LANE_ARRAY(float, xm, 1);
        DH_FOR_LANES_FRESH(lane) {
            const uint32_t l4 = 4u * (uint32_t) lane;
            if (DH_UNLIKELY(have < DH_FTILE + NZ)) {   // the last window of the push: zeros beyond the input
#pragma unroll
                for (int r = 0; r < DH_PF_N; r++) {
                    const uint32_t e = l4 + 4u * DH_WAVE * (uint32_t) r;
                    dh_f4 w = DH_LA(varr, lane)[r];
                    w.x = e + 0u < have ? w.x : 0.0f; w.y = e + 1u < have ? w.y : 0.0f;
                    w.z = e + 2u < have ? w.z : 0.0f; w.w = e + 3u < have ? w.w : 0.0f;
                    DH_LA(varr, lane)[r] = w;
                }
            }
            float mx = 0.0f;                            // (a NaN is skipped here and caught behind the FIR)
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
            {
                // ten v_max3_f32 in ONE statement (as ten statements the compiler put an s_nop behind each)
                static_assert(DH_PF_N == 5, "five groups of four");
                const dh_f4 w0 = varr[0], w1 = varr[1], w2 = varr[2], w3 = varr[3], w4 = varr[4];
                asm("v_max3_f32 %0, |%1|, |%2|, 0\n\tv_max3_f32 %0, |%3|, |%4|, %0\n\tv_max3_f32 %0, |%5|, |%6|, %0\n\tv_max3_f32 %0, |%7|, |%8|, %0\n\t"
                    "v_max3_f32 %0, |%9|, |%10|, %0\n\tv_max3_f32 %0, |%11|, |%12|, %0\n\tv_max3_f32 %0, |%13|, |%14|, %0\n\tv_max3_f32 %0, |%15|, |%16|, %0\n\t"
                    "v_max3_f32 %0, |%17|, |%18|, %0\n\tv_max3_f32 %0, |%19|, |%20|, %0"
                    : "=&v"(mx) : "v"(w0.x), "v"(w0.y), "v"(w0.z), "v"(w0.w), "v"(w1.x), "v"(w1.y), "v"(w1.z), "v"(w1.w), "v"(w2.x), "v"(w2.y), "v"(w2.z), "v"(w2.w),
                                  "v"(w3.x), "v"(w3.y), "v"(w3.z), "v"(w3.w), "v"(w4.x), "v"(w4.y), "v"(w4.z), "v"(w4.w));
            }
#else
#pragma unroll
            for (int r = 0; r < DH_PF_N; r++) {
                const dh_f4 w = DH_LA(varr, lane)[r];
                mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(w.x), __builtin_fabsf(w.y)));
                mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(w.z), __builtin_fabsf(w.w)));
            }
#endif
            DH_LA(xm, lane)[0] = mx;
        }
        float xmax;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
        xmax = dh_wave_max(xm[0]);
#else
        xmax = 0.0f;
        for (int l = 0; l < DH_WAVE; l++) xmax = __builtin_fmaxf(xmax, xm[l][0]);
#endif
        float scale = 1.0f;
        if (xmax == 0.0f) { e_out = 0.0f; k1_out = P.inv_gain; }                       // all zeros in, all zeros out
        else if (xmax >= DH_BOUND_XMAX_LO && xmax <= DH_BOUND_XMAX_HI) {
            union { float f; uint32_t u; } b; b.f = xmax;
            const uint32_t ex = b.u >> 23;                                              // xmax in [2^(ex - 127), 2^(ex - 126))
            b.u = (253u - ex) << 23; scale = b.f;                                       // 2^(126 - ex)
            b.u = (ex + 1u) << 23; k1_out = P.inv_gain * b.f;                           // fl32(1 / gain) 2^(ex - 126): exact scaling
            e_out = P.err_coef_f16 * xmax;
        } else return false;                                                            // tiny, huge or infinite samples: outside the bound's assumptions
        k2_out = k1_out * 0.00048828125f;                                               // 2^-11
        // (the arrays are as long as the last block's K range reaches: the halves behind the window, under zero taps, must be
        // finite -- the lanes of the last group beyond the window store zeros there)
        constexpr uint32_t STORE_LANES = (DH_F16_HALVES_OF(NZ) - 4u * DH_WAVE * (DH_PF_N - 1)) / 4u;
        static_assert(STORE_LANES >= LAST_LANES && STORE_LANES <= DH_WAVE, "the partial group covers the padded arrays");
        DH_FOR_LANES_FRESH(lane) {
            const bool in_last = (uint32_t) lane < LAST_LANES, in_store = (uint32_t) lane < STORE_LANES;
            if (STORE_LANES > LAST_LANES && !in_last) { dh_f4 z; z.x = 0.0f; z.y = 0.0f; z.z = 0.0f; z.w = 0.0f; DH_LA(varr, lane)[DH_PF_N - 1] = z; }
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
            dh_h4* d1 = reinterpret_cast<dh_h4*>(S.xf) + lane;                          // halves 4 lane .. 4 lane + 3 of group 0; group r is 256 halves on
            dh_h4* d2 = reinterpret_cast<dh_h4*>(S.xf + DH_F16_H2_OFFSET_OF(NZ)) + lane;
#pragma unroll
            for (int r = 0; r < DH_PF_N; r++) {
                dh_h4 a, b;
                dh_f16_split4(varr[r], scale, a, b);
                if (r < DH_PF_N - 1 || in_store) { d1[DH_WAVE * r] = a; d2[DH_WAVE * r] = b; }
            }
#else
            uint16_t* d1 = reinterpret_cast<uint16_t*>(S.xf) + 4 * lane;
            uint16_t* d2 = reinterpret_cast<uint16_t*>(S.xf + DH_F16_H2_OFFSET_OF(NZ)) + 4 * lane;
            for (int r = 0; r < DH_PF_N; r++)
                if (r < DH_PF_N - 1 || in_store) dh_f16_split4(varr[lane][r], scale, d1 + 4 * DH_WAVE * r, d2 + 4 * DH_WAVE * r);
#endif
        }
        return true;
    };
    uint32_t nsym = sym_base;                           // symbols of this push so far (a later part of a split push starts behind the earlier ones')
    bool overflow = false;
    const uint32_t max_run = (DH_FTILE - 2) / sps;      // symbols whose windows fit one FIR pass
    // (run planning, sps 10: a whole block fits from every position up to fast_p_end while the symbol count is at most fast_sym_end; -1 = never)
    const int32_t fast_p_end = (nf >= 1003u && nf < 0x40000000u) ? (int32_t) (nf - 1003u) : -1;
    const int32_t fast_sym_end = (P.sym_cap >= (uint32_t) DH_VARIANCE_SYMBOLS && P.sym_cap < 0x40000000u) ? (int32_t) (P.sym_cap - DH_VARIANCE_SYMBOLS) : -1;

    DH_PHASE_MARK(7);
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    // split-f16 FIR: this lane's six tap fragments -- 16-byte loads, 1 KiB per instruction, the same 6 KiB for every wavefront
    // of the chip (L1 / L2 hits).  Requested one iteration ahead (here, and again at the end of every iteration, when the
    // staging registers are free), so they have landed when P2 wants them.
    dh_u4 tapfrag_regs[2 * DH_F16_KSTEPS_OF(NZ > 0 ? NZ : 80)];
#define DH_TAPFRAG_LOAD() do { if constexpr (MF16) { const dh_u4* tf_ = reinterpret_cast<const dh_u4*>(P.tapfrag) + dh_fresh_lane_id_(); \
        _Pragma("unroll") for (int f_ = 0; f_ < 2 * DH_F16_KSTEPS_OF(NZ > 0 ? NZ : 80); f_++) tapfrag_regs[f_] = tf_[DH_WAVE * f_]; } } while (0)
    DH_TAPFRAG_LOAD();
    // Every path through P2 must leave these loads landed, also the rare ones that never look at the fragments: otherwise the
    // compiler's wait-count bookkeeping still sees them in flight further down, and the first instruction that reuses one of
    // their registers -- in the slicing phase -- gets an s_waitcnt vmcnt(0), which also waits for the NEXT window's loads,
    // requested after P3 precisely so that they can stay in flight through P4 - P6 (it cost 1 ms of a 6.8 ms step).
#define DH_TAPFRAG_SETTLE() do { if constexpr (MF16) { _Pragma("unroll") for (int f_ = 0; f_ < 2 * DH_F16_KSTEPS_OF(NZ > 0 ? NZ : 80); f_++) \
        asm volatile("" :: "v"(tapfrag_regs[f_])); } } while (0)
#else
#define DH_TAPFRAG_LOAD() ((void) 0)
#define DH_TAPFRAG_SETTLE() ((void) 0)
#endif
    for (;;) {
        // ---- run planning (wave-uniform): symbols k0 .. k0+m-1 of the current variance block.
        // A symbol at filtered position s is produced iff nf - s > sps + 1 (gfsk_demodulator.cpp:18-22);
        // s_0 = p, s_q = p + q*sps + step_off for q >= 1.
        const int32_t step_off = (k0 == 0) ? off : 0;   // applied after the first symbol of a block
        uint32_t m = 0;
        if (DH_PLAN_FAST && SPS == 10 && DH_LIKELY(k0 == 0 && (int32_t) p <= fast_p_end && (int32_t) nsym <= fast_sym_end)) {
            // the usual run: it starts a variance block and holds all of it -- the stream has the samples whatever the pending step
            // is (nf - p >= 1003 >= 1002 + step_off) and the symbol buffer the room: what the general form below comes to, in three
            // compares instead of forty-five scalar instructions and a load of the symbol capacity from the argument block
            m = DH_VARIANCE_SYMBOLS;
        } else {
            uint32_t lim = dh_min<uint32_t>(DH_VARIANCE_SYMBOLS - k0, max_run);
            lim = dh_min<uint32_t>(lim, P.sym_cap - nsym);
            const int64_t room = (int64_t) nf - (int64_t) p - (int64_t) sps - 2;     // >= 0  <=>  symbol 0 fits
            if (DH_LIKELY(nf >= p && nf - p < 0x40000000u)) {
                // (the same in 32 bits -- every push but an absurdly long one: there is no 64-bit scalar compare or divide, the
                // general form below costs two vector compares and twenty scalar instructions per run)
                const int32_t room32 = (int32_t) (nf - p) - (int32_t) sps - 2;
                if (room32 >= 0 && lim > 0) {
                    const int32_t r1 = room32 - step_off;
                    const uint32_t qmax = r1 >= (int32_t) sps ? (uint32_t) r1 / sps : 0u;
                    m = dh_min<uint32_t>(lim, qmax + 1u);
                }
            } else if (room >= 0 && lim > 0) {
                const int64_t r1 = room - step_off;                                   // q*sps <= r1 for q >= 1
                const uint32_t qmax = r1 >= (int64_t) sps ? (uint32_t) (r1 / sps) : 0u;
                m = dh_min<uint32_t>(lim, qmax + 1u);
            }
            if (m == 0) {
                if (lim == 0 && room >= 0) overflow = true;
                break;
            }
        }
        const uint32_t last_start = p + (m - 1) * sps + (m > 1 ? (uint32_t) step_off : 0u);
        const uint32_t need = last_start + sps - p;     // filtered samples [p, p+need) feed this run
        const uint32_t need_fir = KEEPF ? (uint32_t) DH_FTILE : need;      // (a kernel that also delivers the filtered samples wants the whole pass)
        // (bits 2.. of blk_flags, ring-less kernels: 4 = some run of the current block had a non-zero window, bits 8..15 = runs the block has been cut into)
        if (BOUNDED && k0 == 0) { BS->cur_start = (int32_t) p; BS->cur_off = step_off; BS->blk_flags = (BS->blk_flags & 3u) | 1u; BS->e_blk = 0.0f; }   // symbol k of this block sits at cur_start + k sps + (k ? cur_off : 0)
        bool use_exact = BOUNDED && P.exact_mode == 2;  // this run through the exact FIR (odd samples, odd staging path)
        DH_LANE_ARRAY(float, xmax_lane, 1);
        float e_run = 0.0f;                             // error radius of this run's filtered samples (0: exact)
        bool f16_staged = false, xmax_done = false;     // split-f16 FIR: the window block holds the two arrays of halves
        float k1 = 0.0f, k2 = 0.0f;                     // ... and its outputs are fma(second sum, k2, main sum * k1)

        // ---- P1: stage raw samples V[p .. p+need+NZ) into the padded LDS window (zeros beyond)
        // After the first run of a push the whole window comes straight from `in`: 16 B per lane per load,
        // 16 B per LDS store (a group of four never straddles a pad slot: pads sit every 16 elements).
        // From the second run on, the window was already put there by the previous iteration's prefetch (the kernels that fetch it into
        // registers: PF_REG; the others have had it pulled into L2).
        if (DH_LIKELY(staged && staged_p == p)) {
            if (PF_REG && MF16) { f16_staged = true; xmax_done = true; e_run = st_e_run; k1 = st_k1; k2 = st_k2; }      // (put there by P7 of the previous run)
        } else if (p >= tc && in + (p - tc) + (DH_FTILE + NZ) <= in_end) {
            // whole window inside the input buffer: unconditional loads (all in flight together, see the prefetch
            // below for why that matters), samples past the end of the stream zeroed afterwards
            const float* src = in + (p - tc);
            const uint32_t have = dh_min<uint32_t>(DH_FTILE + NZ, nv - p);
            // Per lane ONE global base and ONE LDS base: group r sits 1024 bytes further in memory (immediate offset of
            // the load) and DH_XP(4 lane + 256 r) = DH_XP(4 lane) + 272 r words further in the padded window -- written
            // out, because the compiler otherwise rebuilds every address from the lane id (6 VALU per group).  Only the
            // last group is partial (elements < DH_FTILE + NZ): its lanes beyond the window re-read group 0.
            constexpr uint32_t GSTEP = NZ > 0 ? DH_XPAD(4u * DH_WAVE) : 4u * DH_WAVE;
            constexpr uint32_t LAST_LANES = (DH_FTILE + NZ - 4u * DH_WAVE * (DH_PF_N - 1)) / 4u;     // lanes of the last group inside the window
            static_assert((DH_FTILE + NZ) % 4u == 0 && (DH_FTILE + NZ) >= 4u * DH_WAVE * (DH_PF_N - 1) && LAST_LANES <= DH_WAVE, "four full groups + a partial one cover the window");
            DH_LANE_ARRAY(dh_f4, v, DH_PF_N);
            DH_FOR_LANES_FRESH(lane) {
                const uint32_t l4 = 4u * (uint32_t) lane;
                const bool in_last = (uint32_t) lane < LAST_LANES;
                const float* lsrc = src + l4;          // one 64-bit address per lane; the groups are immediate offsets of it
#pragma unroll
                for (int r = 0; r < DH_PF_N - 1; r++) DH_LA(v, lane)[r] = dh_load4_unaligned(lsrc + 4 * DH_WAVE * r);
                if constexpr (LAST_LANES > 0) DH_LA(v, lane)[DH_PF_N - 1] = dh_load4_unaligned(lsrc + (in_last ? 4 * DH_WAVE * (DH_PF_N - 1) : 0));
                else DH_LA(v, lane)[DH_PF_N - 1] = DH_LA(v, lane)[0];
                if (!MF16 || use_exact) {
                    if (have < DH_FTILE + NZ) {        // the last window of the push: zeros beyond the input
#pragma unroll
                        for (int r = 0; r < DH_PF_N; r++) {
                            const uint32_t e = l4 + 4u * DH_WAVE * (uint32_t) r;
                            dh_f4 w = DH_LA(v, lane)[r];
                            w.x = e + 0u < have ? w.x : 0.0f; w.y = e + 1u < have ? w.y : 0.0f;
                            w.z = e + 2u < have ? w.z : 0.0f; w.w = e + 3u < have ? w.w : 0.0f;
                            DH_LA(v, lane)[r] = w;
                        }
                    }
                    if (BOUNDED) {                     // max |x| of the window (a NaN is skipped here and caught behind the FIR)
                        float mx = 0.0f;
#pragma unroll
                        for (int r = 0; r < DH_PF_N; r++) {
                            const dh_f4 w = DH_LA(v, lane)[r];
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
                            mx = dh_max3_abs(w.x, w.y, mx); mx = dh_max3_abs(w.z, w.w, mx);
#else
                            mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(w.x), __builtin_fabsf(w.y)));
                            mx = __builtin_fmaxf(mx, __builtin_fmaxf(__builtin_fabsf(w.z), __builtin_fabsf(w.w)));
#endif
                        }
                        DH_LA(xmax_lane, lane)[0] = mx;
                    }
                }
            }
            if (MF16 && !use_exact) {
                // split-f16 FIR: the window goes to LDS as two arrays of halves (stage_f16)
                xmax_done = true;
                if (stage_f16(v, have, e_run, k1, k2)) f16_staged = true; else use_exact = true;
            }
            if (!f16_staged) {
                DH_FOR_LANES_FRESH(lane) {
                    const uint32_t l4 = 4u * (uint32_t) lane;
                    const bool in_last = (uint32_t) lane < LAST_LANES;
                    float* ldst = &S.xf[DH_XP(l4)];
                    dh_lds_store4_at<0>(ldst, DH_LA(v, lane)[0]); dh_lds_store4_at<GSTEP>(ldst, DH_LA(v, lane)[1]);
                    dh_lds_store4_at<2 * GSTEP>(ldst, DH_LA(v, lane)[2]); dh_lds_store4_at<3 * GSTEP>(ldst, DH_LA(v, lane)[3]);
                    static_assert(DH_PF_N == 5, "four full groups + a partial one");
                    if constexpr (LAST_LANES > 0) { if (in_last) dh_lds_store4_at<4 * GSTEP>(ldst, DH_LA(v, lane)[4]); }
                    dh_lds_stores_done();
                }
            }
        } else if (MF16 && !use_exact && DH_F16_EDGE_WINDOWS) {
            // The first window of a push (it starts in the carried tail) and the last ones (they end within DH_FTILE + NZ of
            // the input buffer's end): gathered sample by sample, then the same two arrays of halves as everywhere else.  These
            // runs used to go through the reference-order FIR -- one or two of every push, which is most of a push of a few
            // thousand samples (tools/push_size.py).
            const uint32_t have = dh_min<uint32_t>(DH_FTILE + NZ, nv - p);
            DH_LANE_ARRAY(dh_f4, v, DH_PF_N);
            DH_FOR_LANES_FRESH(lane) {
#pragma unroll
                for (int r = 0; r < DH_PF_N; r++) {
                    const uint32_t e = 4u * (uint32_t) lane + 4u * DH_WAVE * (uint32_t) r;
                    dh_f4 w;
                    w.x = e + 0u < have ? dh_virtual_sample(tail, tc, in, p + e + 0u) : 0.0f;
                    w.y = e + 1u < have ? dh_virtual_sample(tail, tc, in, p + e + 1u) : 0.0f;
                    w.z = e + 2u < have ? dh_virtual_sample(tail, tc, in, p + e + 2u) : 0.0f;
                    w.w = e + 3u < have ? dh_virtual_sample(tail, tc, in, p + e + 3u) : 0.0f;
                    DH_LA(v, lane)[r] = w;
                }
            }
            xmax_done = true;
            if (stage_f16(v, have, e_run, k1, k2)) f16_staged = true;
            else {
                use_exact = true;
                DH_FOR_LANES_FRESH(lane) {
                    for (uint32_t e = lane; e < DH_FTILE + NZ; e += DH_WAVE)
                        S.xf[DH_XP(e)] = p + e < nv ? dh_virtual_sample(tail, tc, in, p + e) : 0.0f;
                }
            }
        } else if (p >= tc) {
            use_exact = BOUNDED;
            const float* src = in + (p - tc);
            const uint32_t have = dh_min<uint32_t>(DH_FTILE + NZ, nv - p);
            DH_FOR_LANES_FRESH(lane) {
                for (uint32_t e = 4u * (uint32_t) lane; e < DH_FTILE + NZ; e += 4u * DH_WAVE) {
                    dh_f4 v;
                    if (e + 4u <= have) v = dh_load4_unaligned(src + e);
                    else {
                        v.x = e + 0u < have ? src[e + 0u] : 0.0f; v.y = e + 1u < have ? src[e + 1u] : 0.0f;
                        v.z = e + 2u < have ? src[e + 2u] : 0.0f; v.w = e + 3u < have ? src[e + 3u] : 0.0f;
                    }
                    dh_store4(&S.xf[DH_XP(e)], v);
                }
            }
        } else {
            use_exact = BOUNDED;
            DH_FOR_LANES_FRESH(lane) {
                for (uint32_t e = lane; e < DH_FTILE + NZ; e += DH_WAVE)
                    S.xf[DH_XP(e)] = p + e < nv ? dh_virtual_sample(tail, tc, in, p + e) : 0.0f;
            }
        }
        DH_BARRIER();
        if (BOUNDED && !use_exact && !xmax_done) {
            float xmax;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
            xmax = dh_wave_max(xmax_lane[0]);
#else
            xmax = 0.0f;
            for (int l = 0; l < DH_WAVE; l++) xmax = __builtin_fmaxf(xmax, xmax_lane[l][0]);
#endif
            if (xmax == 0.0f) e_run = 0.0f;             // all zeros in, all zeros out of either FIR
            else if (xmax >= DH_BOUND_XMAX_LO && xmax <= DH_BOUND_XMAX_HI) e_run = P.err_coef * xmax;
            else use_exact = true;                      // tiny, huge or infinite samples: outside the bound's assumptions
        }
        DH_PHASE_MARK(0);

        // ---- P2: FIR into registers, then (after every lane has read its window) back into the window block.
        // The filtered samples go back UNPADDED (element n at word n, four 16-byte stores per lane): the symbol
        // windows of P3 then sit at compile-time offsets from one per-lane base, which saves two address
        // instructions per sample on the VALU; the bank conflicts of the 16-words-apart stores cost LDS cycles
        // only, and the LDS pipe has slack.  Without an RRC stage the staged (padded) samples are used as they are.
        if (NZ > 0 && DH_STOP_AFTER >= 2) {
            DH_LANE_ARRAY(float, fo, DH_FIR_L);
            DH_COMPILER_FENCE();                        // forces the tap loads below to stay inside this pass
            // (taps from the LDS copy into vector registers: as scalar operands they free 40 VGPRs but cost this kernel 5 %,
            // see DhFirBatch)
            constexpr bool MFMA = DH_MFMA_FIR && NZ > 0 && NZ <= 80 && (BOUNDED || FAST) && !MF16;      // the fused FIR as f32 MFMAs (dh_fir_mfma)
            int mf_layout = 0;                          // fo[] holds a matrix-core output layout: 1 = DH_MF_OUT, 2 = DH_F16_OUT
            if (BOUNDED) {
                uint64_t vote_bad = 0;
                if (DH_LIKELY(!use_exact)) {
                    if constexpr (MF16) {
                        // (f16_staged holds: every staging path that does not fill the two arrays of halves sets use_exact)
                        DH_FOR_LANES_FRESH(lane) {
                            float t = 0.0f;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
                            dh_fir_f16<MF16 ? NZ : 80>(S.xf, tapfrag_regs, lane, k1, k2, fo);
#else
                            dh_fir_f16<MF16 ? NZ : 80>(S.xf, reinterpret_cast<const uint32_t (*)[DH_WAVE][4]>(P.tapfrag), lane, k1, k2, fo[lane]);
#endif
#pragma unroll
                            for (int j = 0; j < DH_FIR_L; j++) t = __builtin_fmaf(DH_LA(fo, lane)[j], 0.0f, t);       // NaN or infinity anywhere: 0 * y is NaN
                            DH_BALLOT_ACC(vote_bad, !(t == 0.0f), lane);
                        }
                        mf_layout = 2;
                    } else if constexpr (MFMA) {
                        DH_FOR_LANES_FRESH(lane) {
                            bool bad = false;
                            float acc[DH_FIR_L];
                            dh_fir_mfma<MFMA ? NZ : 80>(S.tapsf, S.xf, lane, acc);
                            dh_fir_finish<true>(acc, P.gain, P.rgain, P.inv_gain, DH_LA(fo, lane), &bad);
                            DH_BALLOT_ACC(vote_bad, bad, lane);
                        }
                        mf_layout = 1;
                    } else {
                        float tv[NZ / 2 + 1];
#pragma unroll
                        for (int i = 0; i <= NZ / 2; i++) tv[i] = S.tapsf[i];
                        DH_FOR_LANES_FRESH(lane) {
                            bool bad = false;
                            if ((uint32_t) (lane * DH_FIR_L) < need_fir)
                                dh_fir_lane<NZ, true>(tv, P.gain, P.rgain, P.inv_gain, S.xf, lane, DH_LA(fo, lane), &bad);
                            DH_BALLOT_ACC(vote_bad, bad, lane);
                        }
                    }
                }
                if (DH_UNLIKELY(use_exact || vote_bad)) {            // a NaN / infinity among the samples: the reference's arithmetic decides
                    use_exact = true; e_run = 0.0f; BS->n_exact_runs++; mf_layout = 0;
                    if (MF16 && f16_staged) {           // the window block holds halves: the reference's FIR wants the raw samples back
                        DH_BARRIER();
                        DH_FOR_LANES_FRESH(lane) {
                            for (uint32_t e = lane; e < DH_FTILE + NZ; e += DH_WAVE)
                                S.xf[DH_XP(e)] = p + e < nv ? dh_virtual_sample(tail, tc, in, p + e) : 0.0f;
                        }
                        DH_BARRIER();
                    }
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
                    dh_exact_fir_pass<NZ>(P, S, need_fir, nullptr, fo);
#else
                    dh_exact_fir_pass<NZ>(P, S, need_fir, fo, nullptr);
#endif
                }
            } else if constexpr (MFMA) {
                DH_FOR_LANES_FRESH(lane) {
                    float acc[DH_FIR_L];
                    dh_fir_mfma<MFMA ? NZ : 80>(S.tapsf, S.xf, lane, acc);
                    dh_fir_finish<true>(acc, P.gain, P.rgain, P.inv_gain, DH_LA(fo, lane), nullptr);
                }
                mf_layout = 1;
            } else {
                float tv[NZ / 2 + 1];
#pragma unroll
                for (int i = 0; i <= NZ / 2; i++) tv[i] = S.tapsf[i];
                DH_FOR_LANES_FRESH(lane) {
                    if ((uint32_t) (lane * DH_FIR_L) < need_fir)
                        dh_fir_lane<NZ, FAST>(tv, P.gain, P.rgain, P.inv_gain, S.xf, lane, DH_LA(fo, lane));
                }
            }
            DH_BARRIER();
            if (MF16 && mf_layout == 2) {
                DH_FOR_LANES_FRESH(lane) {
                    float* dst = S.xf + DH_F16_OUT(0, 0, lane >> 4, lane & 15);
#pragma unroll
                    for (int T = 0; T < 4; T++)
#pragma unroll
                        for (int r = 0; r < 4; r++) dst[DH_F16_OUT(T, r, 0, 0)] = DH_LA(fo, lane)[4 * T + r];
                }
            } else if (MFMA && mf_layout == 1) {
                // sample DH_MF_OUT(T, r, g, n) of the run from register 4 T + r of lane 16 g + n: compile-time offsets
                DH_FOR_LANES_FRESH(lane) {
                    float* dst = S.xf + DH_MF_OUT(0, 0, lane >> 4, lane & 15);
#pragma unroll
                    for (int T = 0; T < 4; T++)
#pragma unroll
                        for (int r = 0; r < 4; r++) dst[DH_MF_OUT(T, r, 0, 0)] = DH_LA(fo, lane)[4 * T + r];
                }
            } else {
                DH_FOR_LANES_FRESH(lane) {
                    if ((uint32_t) (lane * DH_FIR_L) < need_fir) {
                        dh_f4a* dst = reinterpret_cast<dh_f4a*>(S.xf + DH_FIR_L * lane);
#pragma unroll
                        for (int j = 0; j < DH_FIR_L / 4; j++) {
                            dh_f4a v; v.x = DH_LA(fo, lane)[4 * j]; v.y = DH_LA(fo, lane)[4 * j + 1];
                            v.z = DH_LA(fo, lane)[4 * j + 2]; v.w = DH_LA(fo, lane)[4 * j + 3];
                            dst[j] = v;
                        }
                    }
                }
            }
            DH_BARRIER();
        }
        DH_TAPFRAG_SETTLE();
        DH_PHASE_MARK(1);
        const float* fbuf = S.xf;

        // ---- prefetch: the raw window of the NEXT run.  Its start is already known (the timing decision of this
        // block only moves symbols 1.. of the next one), the window block is idle from here to the end of the
        // iteration, and P4-P6 are latency-bound with few live registers: the HBM latency of these loads hides
        // behind them.  Five 16-byte loads per lane, parked in registers until P7.
        const uint32_t p_next = last_start + sps + ((k0 == 0 && m == 1) ? (uint32_t) step_off : 0u);
        const bool pf_ok = p_next >= tc && p_next < nv;
        const uint32_t pf_have = pf_ok ? dh_min<uint32_t>(DH_FTILE + NZ, nv - p_next) : 0u;
        DH_LANE_ARRAY(dh_f4, pfr, PF_REG ? DH_PF_N : 1);  // split-f16 kernels: the next window, parked in registers through P4 - P6
        // Only when the whole window lies inside the input buffer (every run but the last ones of the last
        // channel): the loads are then unconditional, there is nothing to merge, and all five are in flight
        // together; samples past pf_have are zeroed in P7.  Otherwise the next iteration stages in P1.
        const bool pf_plain = pf_ok && (in + (p_next - tc) + (DH_FTILE + NZ) <= in_end);
        const bool pf_reg = PF_REG && pf_plain && P.exact_mode != 2;
        if constexpr (KEEPF) {
            // This run's filtered samples [p, p_next) leave as they stand in LDS (each store instruction a run of 64 consecutive
            // words); what the previous push already delivered -- positions in front of tc - NZ, the first new sample's -- is
            // skipped, and nothing is delivered beyond the last complete output.
            const uint32_t f_lo = tc > (uint32_t) NZ ? tc - (uint32_t) NZ : 0u;
            const uint32_t s0 = p < f_lo ? f_lo - p : 0u;
            const uint32_t s1 = dh_min<uint32_t>(p_next - p, nf > p ? nf - p : 0u);
            float* orow = P.filt_out + (size_t) ch * P.filt_stride + ((ptrdiff_t) p + (ptrdiff_t) NZ - (ptrdiff_t) tc);
            DH_FOR_LANES_FRESH(lane) {
                for (uint32_t e = s0 + (uint32_t) lane; e < s1; e += DH_WAVE) orow[e] = fbuf[e];
            }
        }
        auto issue_next_window = [&]() __attribute__((always_inline)) {
        if (DH_LIKELY(pf_reg)) {
            // The split-f16 FIR leaves registers free where the packed-FMA FIR had none: the next window's five 16-byte loads
            // per lane are issued here, straight from HBM, land while P4 - P6 run, and P7 turns them into the two arrays of
            // halves -- the next iteration starts at the matrix cores.
            constexpr uint32_t LAST_LANES = (DH_FTILE + NZ - 4u * DH_WAVE * (DH_PF_N - 1)) / 4u;
            const float* src = in + (p_next - tc);
            DH_FOR_LANES_FRESH(lane) {
                const float* lsrc = src + 4u * (uint32_t) lane;
                const bool in_last = (uint32_t) lane < LAST_LANES;
#pragma unroll
                for (int r = 0; r < DH_PF_N - 1; r++) DH_LA(pfr, lane)[PF_REG ? r : 0] = dh_load4_stream(lsrc + 4 * DH_WAVE * r);
                if constexpr (LAST_LANES > 0) DH_LA(pfr, lane)[PF_REG ? DH_PF_N - 1 : 0] = dh_load4_stream(lsrc + (in_last ? 4 * DH_WAVE * (DH_PF_N - 1) : 0));
                else (void) in_last;
            }
        } else {
        // Register-free variant: one dword per 128-byte line of the next window is requested now, which pulls the
        // lines into L2; the next iteration's P1 then stages from L2 instead of HBM.  The dwords themselves are not
        // wanted: global_load_lds_dword drops them (lane l -> m0 + 4 l) into a part of the window block that is dead
        // until the next staging, so no vector register is tied to a load the compiler does not know about.  They
        // have landed before the next P1 stores anything there (its own, younger loads are waited for first), and
        // an explicit s_waitcnt follows the loop for the last one.
        if (pf_plain) {
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
            const uint32_t pf_lane = (uint32_t) dh_fresh_lane_id_();       // not loop-invariant: a hoisted address is spilled
            const float* line = in + (p_next - tc) + 32u * pf_lane;
            const uint32_t sink = (uint32_t) (uintptr_t) (__attribute__((address_space(3))) float*) (S.xf + DH_PF_SINK);
            uint32_t keep_m0;
            if (32u * pf_lane < DH_FTILE + NZ)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep_m0) : "s"(sink), "v"(line) : "memory");
#endif
        }
        }
        };

        DH_COMPILER_FENCE();                            // the bookkeeping in LDS is read here, not carried across the FIR
        float e_blk = 0.0f;
        if (BOUNDED) { e_blk = dh_uniform_f(__builtin_fmaxf(BS->e_blk, e_run)); BS->e_blk = e_blk; }
        const float e_eff = BOUNDED ? dh_uniform_f(__builtin_fmaxf(e_run, __builtin_fmaxf(BS->e_cur, BS->e_prev))) : 0.0f;
        const float T = DH_BOUND_T_FACTOR * e_eff;
        uint64_t vote_a = 0, vote_b = 0;                // doubtful symbols of the run: bit l of vote_a = symbol 2 l, of vote_b = symbol 2 l + 1
        // Straight-line per half of the run (m <= 100 symbols: lanes 0..63, then 0..35): every lane computes, a predicate
        // guards the store -- written as a loop over the lane's symbols this compiled to a real loop with exec-mask
        // bookkeeping (half of this phase's instructions were scalar).  The wave-uniform choices (4 / 2 levels, invert,
        // exact_mode) are selects, not branches.
        const bool four_levels = LV ? LV == 4 : P.levels == 4;
        // (DH_FLAG_EXACT_SYMBOLS -- every symbol decided exactly -- is an infinite threshold, not another mask)
        const bool force_doubt = BOUNDED && P.exact_mode == 1;
        const bool e_pos = e_eff > 0.0f || force_doubt;
        const float T_eff = force_doubt ? __builtin_inff() : T;
        const float inv_width = 1.0f / (float) (ev_hi - ev_lo);                       // (a power of two for sps 10: the product below is the division)
        const bool width_pow2 = ((ev_hi - ev_lo) & (ev_hi - ev_lo - 1u)) == 0u;
        const bool pair_store = ((uint32_t) (uintptr_t) syms + nsym) % 2u == 0u;      // (paired slicing phases: symbols 2 l, 2 l + 1 of the run as one 16-bit store)

        // ---- P3: symbol windows (gfsk_demodulator.cpp:28-35, 82-83): dh_symbol_windows
        dh_symbol_windows<SPS>(S, fbuf, k0, m, step_off, sps, ev_lo, ev_hi, sps_rcp);
        DH_BARRIER();
        DH_PHASE_MARK(2);

        issue_next_window();

        // ---- P4: sliding AGC min/max as two wave scans
        DhAgcPair agc = { 0.0f, 0.0f, 0.0f, 0.0f };
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
        if (DH_STOP_AFTER >= 4) agc = dh_agc_scan(S, k0, k0 + m);
#else
        if (DH_STOP_AFTER >= 4) dh_agc_scan(S, k0, k0 + m);
#endif
        DH_BARRIER();
        DH_PHASE_MARK(3);

        // ---- P5: thresholds + slice (gfsk_demodulator.cpp:88-106 / fsk_demodulator.cpp:89-99): dh_slice_symbols; what it leaves in doubt: dh_settle_doubts
        if (DH_STOP_AFTER >= 5) {
            const DhSliceSetup U = { four_levels, e_pos, width_pow2, pair_store, T_eff, inv_width };
            dh_slice_symbols<SPS, LV, BOUNDED>(P, S, agc, syms + nsym, k0, m, ev_lo, ev_hi, U, vote_a, vote_b);
        }
        if (BOUNDED && DH_UNLIKELY((vote_a | vote_b) != 0)) {
            const DhStreamView V = { tail, tc, in, nv, sps_rcp };
            dh_settle_doubts<NZ>(P, S, BS, V, syms + nsym, k0, e_eff, sps, ev_lo, ev_hi, vote_a, vote_b);
        }

        DH_PHASE_MARK(4);
        // ---- P6: end of a variance block -> timing decision (gfsk_demodulator.cpp:41-80): dh_timing_decision
        int32_t new_off = 0;
        const bool block_done = (k0 + m == DH_VARIANCE_SYMBOLS);
        if (block_done && DH_STOP_AFTER >= 6) {
            const DhStreamView V = { tail, tc, in, nv, sps_rcp };
            new_off = dh_timing_decision<NZ, SPS, BOUNDED>(P, S, BS, V, sps, k0, e_blk);
        }

        DH_PHASE_MARK(5);
        // ---- P7: commit the run (wave-uniform bookkeeping) and fold new volumes into the ring
        DH_FOR_LANES_FRESH(lane) {
            // a run has at most 100 symbols: two predicated copies per lane, both loads in flight together (as a loop this was
            // two trips of exec-mask bookkeeping with a wait in each)
            const uint32_t ka = k0 + (uint32_t) lane, kb = ka + DH_WAVE, ke = k0 + m;
            const float va_ = S.vol_new[ka < ke ? ka : k0], vb_ = S.vol_new[kb < ke ? kb : k0];
            if (ka < ke) S.vol_old[ka] = va_;
            if (kb < ke) S.vol_old[kb] = vb_;
        }
        staged = false;
        if (DH_LIKELY(pf_reg)) {
            if constexpr (PF_REG && MF16) { staged = stage_f16(pfr, pf_have, st_e_run, st_k1, st_k2); staged_p = p_next; }
            else if constexpr (PF_PLAIN) {
                // no RRC stage: the window is the 1 024 samples themselves (four groups of four per lane, unpadded), zeros beyond the stream's end
                static_assert(!PF_PLAIN || (DH_FTILE == 4u * DH_WAVE * (DH_PF_N - 1)), "four full groups cover the window");
                DH_FOR_LANES_FRESH(lane) {
                    const uint32_t l4 = 4u * (uint32_t) lane;
                    float* ldst = &S.xf[l4];
#pragma unroll
                    for (int r = 0; r < DH_PF_N - 1; r++) {
                        dh_f4 w = DH_LA(pfr, lane)[PF_REG ? r : 0];
                        if (DH_UNLIKELY(pf_have < DH_FTILE)) {
                            const uint32_t e = l4 + 4u * DH_WAVE * (uint32_t) r;
                            w.x = e + 0u < pf_have ? w.x : 0.0f; w.y = e + 1u < pf_have ? w.y : 0.0f;
                            w.z = e + 2u < pf_have ? w.z : 0.0f; w.w = e + 3u < pf_have ? w.w : 0.0f;
                        }
                        dh_store4(ldst + 4 * DH_WAVE * r, w);
                    }
                }
                staged = true; staged_p = p_next;
            }
        }
        DH_BARRIER();
        p = p_next;
        nsym += m;
        if (k0 == 0) off = 0;                           // the pending step has been consumed (:36-38)
        k0 += m;
        if (block_done) { k0 = 0; off = new_off; }
        if (BOUNDED) {
            // radii of the ring entries: the last 100 symbols always lie inside the current bucket + the one before it
            const float ec = __builtin_fmaxf(BS->e_cur, e_run);
            const uint32_t cnt = dh_uniform(BS->e_count) + m;
            if (cnt >= DH_VOLUME_RB_SIZE) { BS->e_prev = ec; BS->e_cur = 0.0f; BS->e_count = 0; }
            else { BS->e_cur = ec; BS->e_count = cnt; }
            if (block_done) { BS->prev_start = BS->cur_start; BS->prev_off = BS->cur_off; BS->blk_flags = (BS->blk_flags & 1u) ? 2u : 0u; }
        }
        DH_TAPFRAG_LOAD();                              // for the next run
        DH_PHASE_MARK(6);
    }

#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // a last L2-touch load may still be writing into the window block
#endif
    if constexpr (KEEPF) {
        // the outputs behind the last run (less than a symbol's worth, unless the symbol buffer is full): the reference's arithmetic,
        // sample by sample, while the tail it reads is still the old one
        const uint32_t f_lo = tc > (uint32_t) NZ ? tc - (uint32_t) NZ : 0u;
        float* orow = P.filt_out + (size_t) ch * P.filt_stride;
        DH_FOR_LANES(lane) {
            for (uint32_t f = dh_max<uint32_t>(p, f_lo) + (uint32_t) lane; f < nf; f += DH_WAVE)
                orow[f + (uint32_t) NZ - tc] = dh_exact_filtered<NZ>(tail, tc, in, nv, S.tapsf, P.gain, P.rgain, (int32_t) f);
        }
    }
    // ---- write back state: rings, header, and the raw tail V[base .. nv): the unread samples (from p) and, in the
    // error-bounded kernels, up to dh_history(sps) samples behind them
    const uint32_t tail_max = dh_tail_max(sps);
    const uint32_t keep = BOUNDED ? dh_min<uint32_t>(dh_history(sps), p) : 0u;
    const uint32_t base = p - keep;
    const uint32_t new_tc = nv - base;                 // = history + unread filtered samples + NZ
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < DH_VOLUME_RB_SIZE; j += DH_WAVE) st[DH_ST_VOL + j] = S.vol_old[j];
        for (uint32_t j = lane; j < DH_VARIANCE_SYMBOLS * sps; j += DH_WAVE) st[DH_ST_VAR + j] = S.var_rb[j];
    }
    // the tail may overlap its own source when base < tc: in chunks through LDS (a chunk is read completely before any of
    // it is written, and later chunks only read further ahead)
    for (uint32_t c0 = 0; c0 < new_tc && c0 < tail_max; c0 += DH_FTILE) {
        DH_BARRIER();
        DH_FOR_LANES(lane) {
            for (uint32_t j = c0 + lane; j < new_tc && j < tail_max && j < c0 + DH_FTILE; j += DH_WAVE) S.xf[j - c0] = dh_virtual_sample(tail, tc, in, base + j);
        }
        DH_BARRIER();
        DH_FOR_LANES(lane) {
            for (uint32_t j = c0 + lane; j < new_tc && j < tail_max && j < c0 + DH_FTILE; j += DH_WAVE) tail[j] = S.xf[j - c0];
        }
    }
    DH_FOR_LANES(lane) {
        if (DH_IS_LANE0(lane)) {
            // (timing blocks completed by this call: the symbol counter of the block wraps at 100 exactly when one ends -- counted here,
            // from the counter as the call found it, not with a read-modify-write of LDS by one lane in every run)
            // (sym_base is not kept through the loop: a later part finds it where it read it, until P.sym_count is stored below)
            const uint32_t produced = nsym - (part_lo ? P.sym_count[ch] : 0u);
            sth[DH_ST_BLOCKS] += (sth[DH_ST_K] + produced) / DH_VARIANCE_SYMBOLS;
            sth[DH_ST_K] = k0;
            sth[DH_ST_OFF] = (uint32_t) off;
            sth[DH_ST_TAIL] = new_tc < tail_max ? new_tc : tail_max;
            sth[DH_ST_NSYM] += produced;
            if (BOUNDED) {
                sth[DH_ST_P0] = keep;
                sth[DH_ST_CUR_START] = (uint32_t) (BS->cur_start - (int32_t) base); sth[DH_ST_CUR_OFF] = (uint32_t) BS->cur_off;
                sth[DH_ST_PREV_START] = (uint32_t) (BS->prev_start - (int32_t) base); sth[DH_ST_PREV_OFF] = (uint32_t) BS->prev_off;
                sth[DH_ST_BLOCK_FLAGS] = BS->blk_flags; sth[DH_ST_E_COUNT] = BS->e_count;
                st[DH_ST_E_CUR] = BS->e_cur; st[DH_ST_E_PREV] = BS->e_prev; st[DH_ST_E_BLOCK] = BS->e_blk;
                sth[DH_ST_UNCERTAIN] += BS->n_uncertain; sth[DH_ST_EXACT_RUNS] += BS->n_exact_runs; sth[DH_ST_EXACT_BLOCKS] += BS->n_exact_blocks;
            }

            sth[DH_ST_ORDERED] += S.stats[1];
            P.sym_count[ch] = nsym;
            if ((overflow || new_tc > tail_max) && P.overflow) *P.overflow = 1u;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stand-alone RRC stage (materialised output, BASELINE config 2): one 1024-sample tile of one
// channel per wavefront.  `hist` holds the last NZ inputs of the previous push ([B][NZ]).
struct DhRrcParams {
    const float* in; size_t in_stride;
    float* out; size_t out_stride;
    const float* hist;                                  // [B][nz] previous inputs (zeros after reset)
    const uint32_t* n_per;                              // ragged pushes: [B] samples of each channel (<= n), or null
    uint32_t n, n_channels, nz;
    int32_t fast;
    double gain, rgain; float inv_gain;
    float taps[DH_MAX_NZ / 2 + 1];
};

template <int NZ, bool FAST>
DH_HD void dh_rrc_tile(const DhRrcParams& R, uint32_t ch, uint32_t tile, DhDspShared& S) {
    const float* in = R.in + (size_t) ch * R.in_stride;
    const float* hist = R.hist + (size_t) ch * NZ;
    float* out = R.out + (size_t) ch * R.out_stride;
    const uint32_t t0 = tile * DH_FTILE;
    const uint32_t n_ch = R.n_per ? dh_min<uint32_t>(dh_uniform(R.n_per[ch]), R.n) : R.n;
    if (t0 >= n_ch) return;                             // (ragged pushes: the grid covers the longest row)
    const uint32_t cnt = dh_min<uint32_t>(DH_FTILE, n_ch - t0);
    // virtual stream = hist (NZ samples) ++ in; output t needs virtual [t, t+NZ]
    if (t0 >= (uint32_t) NZ) {
        // every tile but the first reads `in` only: 16 bytes per lane per load
        const float* src = in + (t0 - NZ);
        const uint32_t have = cnt + NZ;
        if (have >= DH_FTILE + NZ) {
            // a full tile: the loads are issued unconditionally (clamped addresses) so that all of them are in flight
            // together -- under a branch each would be waited for before the next starts
            constexpr int NV = (DH_FTILE + NZ + 4 * DH_WAVE - 1) / (4 * DH_WAVE);
            DH_FOR_LANES(lane) {
                dh_f4 v[NV];
#pragma unroll
                for (int r = 0; r < NV; r++) {
                    const uint32_t e = 4u * (uint32_t) lane + 4u * DH_WAVE * (uint32_t) r;
                    v[r] = dh_load4_unaligned(src + dh_min<uint32_t>(e, DH_FTILE + NZ - 4u));
                }
#pragma unroll
                for (int r = 0; r < NV; r++) {
                    const uint32_t e = 4u * (uint32_t) lane + 4u * DH_WAVE * (uint32_t) r;
                    if (e < DH_FTILE + NZ) dh_store4(&S.xf[DH_XPAD(e)], v[r]);
                }
            }
        } else {
            DH_FOR_LANES(lane) {
                for (uint32_t e = 4u * (uint32_t) lane; e < DH_FTILE + NZ; e += 4u * DH_WAVE) {
                    dh_f4 v; v.x = 0.0f; v.y = 0.0f; v.z = 0.0f; v.w = 0.0f;
                    if (e + 4u <= have) v = dh_load4_unaligned(src + e);
                    else if (e < have) {
                        v.x = src[e];
                        if (e + 1u < have) v.y = src[e + 1u];
                        if (e + 2u < have) v.z = src[e + 2u];
                    }
                    dh_store4(&S.xf[DH_XPAD(e)], v);
                }
            }
        }
    } else {
        DH_FOR_LANES(lane) {
            for (uint32_t e = lane; e < DH_FTILE + NZ; e += DH_WAVE) {
                const uint32_t v = t0 + e;
                float x = 0.0f;
                if (e < cnt + NZ) x = v < (uint32_t) NZ ? hist[v] : in[v - NZ];
                S.xf[DH_XPAD(e)] = x;
            }
        }
    }
    DH_BARRIER();
    DH_COMPILER_FENCE();                                // the taps become live after the staging registers are dead
    float tv[NZ / 2 + 1];
#pragma unroll
    for (int i = 0; i <= NZ / 2; i++) tv[i] = R.taps[i];       // harness only: the GPU reads R.taps as scalar pairs (DhFirBatch SG)
    // Each lane produces 16 consecutive outputs.  Stored straight from registers, one store instruction would touch
    // 64 separate 64-byte pieces of the row; instead the tile goes through the (now idle) window block and leaves
    // as four fully coalesced 1 KB stores.
    DH_LANE_ARRAY(float, fo, DH_FIR_L);
    DH_FOR_LANES(lane) {
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
        if ((uint32_t) lane * DH_FIR_L < cnt) dh_fir_lane<NZ, FAST, true>(R.taps, R.gain, R.rgain, R.inv_gain, S.xf, lane, DH_LA(fo, lane));
#else
        if ((uint32_t) lane * DH_FIR_L < cnt) dh_fir_lane<NZ, FAST>(tv, R.gain, R.rgain, R.inv_gain, S.xf, lane, DH_LA(fo, lane));
#endif
    }
    DH_BARRIER();                                       // every lane has read its window
    DH_FOR_LANES(lane) {
        if ((uint32_t) lane * DH_FIR_L < cnt) {
            dh_f4a* dst = reinterpret_cast<dh_f4a*>(S.xf + DH_FIR_L * lane);
#pragma unroll
            for (int q = 0; q < DH_FIR_L / 4; q++) {
                dh_f4a v; v.x = DH_LA(fo, lane)[4 * q]; v.y = DH_LA(fo, lane)[4 * q + 1];
                v.z = DH_LA(fo, lane)[4 * q + 2]; v.w = DH_LA(fo, lane)[4 * q + 3];
                dst[q] = v;
            }
        }
    }
    DH_BARRIER();
    DH_FOR_LANES(lane) {
        float* dst = out + t0;
#pragma unroll
        for (int r = 0; r < DH_FTILE / (4 * DH_WAVE); r++) {
            const uint32_t e = 4u * (uint32_t) lane + 4u * DH_WAVE * (uint32_t) r;
            const dh_f4a v = *reinterpret_cast<const dh_f4a*>(S.xf + e);
            if (e + 4u <= cnt) { dh_f4 w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w; dh_store4_unaligned(dst + e, w); }
            else {
                if (e + 0u < cnt) dst[e + 0u] = v.x;
                if (e + 1u < cnt) dst[e + 1u] = v.y;
                if (e + 2u < cnt) dst[e + 2u] = v.z;
            }
        }
    }
    DH_BARRIER();
}
