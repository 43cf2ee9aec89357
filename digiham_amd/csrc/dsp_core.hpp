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
//   min_k = min( old ring entries outside [k0..k], new entries k0..k )
// and the per-phase variance runs on `sps` lanes at the end of the block.
#pragma once

#include "dh_portable.hpp"

#define DH_VARIANCE_SYMBOLS 100      // include/gfsk_demodulator.hpp:5
#define DH_VOLUME_RB_SIZE 100        // include/gfsk_demodulator.hpp:6
#define DH_FTILE 1024                // filtered samples produced per FIR pass (64 lanes x 16)
#define DH_FIR_L 16                  // consecutive outputs per lane
#define DH_MAX_NZ 160
#define DH_MAX_SPS 40
#define DH_TAIL_MAX 256              // raw samples carried between pushes (>= nz + sps + 2)
#define DH_STATE_HDR 16              // u32 words of per-channel header

// per-channel state block in HBM (floats / u32 words, AoS, `state_stride` words apart):
//   [0..15]                      header: k, pending offset, tail count, symbols produced (lo), ...
//   [16 .. 16+100)               volume ring  (volume_rb)
//   [116 .. 116+100*sps)         variance ring (variance_rb)
//   [.. + DH_TAIL_MAX)           raw-sample tail: the last nz inputs + not yet consumed samples
enum { DH_ST_K = 0, DH_ST_OFF = 1, DH_ST_TAIL = 2, DH_ST_NSYM = 3 };
#define DH_ST_VOL DH_STATE_HDR
#define DH_ST_VAR (DH_STATE_HDR + DH_VOLUME_RB_SIZE)

struct DhDspParams {
    const float* in; size_t in_stride; uint32_t n;     // n new samples per channel
    float* state; size_t state_stride;                 // per-channel state (see above), in 4-byte words
    uint8_t* syms; size_t sym_stride;                  // symbol output [B][sym_stride]
    uint32_t* sym_count;                               // [B] symbols produced by this push
    uint32_t sym_cap;                                  // max symbols a push may append per channel
    uint32_t* overflow;                                // set to 1 if sym_cap was hit
    uint32_t n_channels;
    uint32_t sps, lo, hi;                              // samples/symbol, [lo,hi) = mid-symbol evaluation window
    int32_t levels, invert;                            // 4 = GFSK, 2 = FSK
    uint32_t nz;                                       // FIR order (0 = no RRC stage)
    int32_t fast;                                      // 1 = FMA FIR
    double gain; float inv_gain;
    float taps[DH_MAX_NZ / 2 + 1];                     // first half + centre of the symmetric response
};

DH_HD uint32_t dh_state_words(uint32_t sps) { return DH_ST_VAR + DH_VARIANCE_SYMBOLS * sps + DH_TAIL_MAX; }

// LDS block of one wavefront.  xbuf is padded one word per 16 so that the FIR's per-lane
// sliding windows (stride 16 words between lanes) hit 32 different banks.
#define DH_XPAD(i) ((i) + ((i) >> 4))
struct DhDspShared {
    float xbuf[DH_XPAD(DH_FTILE + DH_MAX_NZ) + 1];
    float fbuf[DH_FTILE + 8];
    float vol_old[DH_VOLUME_RB_SIZE];                  // ring content before the current run
    float vol_new[DH_VOLUME_RB_SIZE];                  // entries written by the current run
    float mn_pre[DH_VOLUME_RB_SIZE], mx_pre[DH_VOLUME_RB_SIZE];   // running min/max over new[k0..k]
    float mn_suf[DH_VOLUME_RB_SIZE + 1], mx_suf[DH_VOLUME_RB_SIZE + 1];   // min/max over old(k..99]
    float sum[DH_VOLUME_RB_SIZE];                      // mid-symbol window sums of the current run
    double variance[DH_MAX_SPS];
    // variance ring follows (100 * sps floats), sized at launch
    float var_rb[1];
};

DH_HD size_t dh_dsp_shared_bytes(uint32_t sps) {
    return sizeof(DhDspShared) + sizeof(float) * (size_t) (DH_VARIANCE_SYMBOLS * sps);
}

// ---------------------------------------------------------------------------------------------
// FIR pass: lane computes outputs [lane*16, lane*16+16) of the tile from the padded LDS window.
//   x[e] = S.xbuf[DH_XPAD(e)] holds input sample (tile_start - NZ + e);  out[j] = sum_i c[i]*x[j+i]
template <int NZ, bool FAST>
DH_HD void dh_fir_lane(const DhDspParams& P, const DhDspShared& S, int lane, float* out16) {
    float acc[DH_FIR_L];
#pragma unroll
    for (int j = 0; j < DH_FIR_L; j++) acc[j] = 0.0f;
    const int base = lane * DH_FIR_L;
#pragma unroll
    for (int t = 0; t < NZ + DH_FIR_L; t++) {
        const float x = S.xbuf[DH_XPAD(base + t)];
#pragma unroll
        for (int j = 0; j < DH_FIR_L; j++) {
            const int i = t - j;                       // tap index
            if (i >= 0 && i <= NZ) {
                const float c = P.taps[i <= NZ / 2 ? i : NZ - i];
                if (FAST) acc[j] = __builtin_fmaf(c, x, acc[j]);
                else acc[j] = acc[j] + c * x;          // separately rounded (-ffp-contract=off)
            }
        }
    }
#pragma unroll
    for (int j = 0; j < DH_FIR_L; j++) {
        if (FAST) out16[j] = acc[j] * P.inv_gain;
        else out16[j] = (float) ((double) acc[j] / P.gain);
    }
}

// virtual input stream of a channel for this push: carried tail followed by the new samples
DH_HD float dh_virtual_sample(const float* tail, uint32_t tc, const float* in, uint32_t idx) {
    return idx < tc ? tail[idx] : in[idx - tc];
}

// ---------------------------------------------------------------------------------------------
// One channel, one push.  `S` is this wavefront's LDS block.  Called by all 64 lanes (device) or
// once (host harness; the DH_FOR_LANES loops then iterate the lanes).
template <int NZ, bool FAST>
DH_HD void dh_rrc_demod_channel(const DhDspParams& P, uint32_t ch, DhDspShared& S) {
    float* st = P.state + (size_t) ch * P.state_stride;
    uint32_t* sth = (uint32_t*) st;
    const uint32_t sps = P.sps;
    float* tail = st + DH_ST_VAR + DH_VARIANCE_SYMBOLS * sps;
    const float* in = P.in + (size_t) ch * P.in_stride;
    uint8_t* syms = P.syms + (size_t) ch * P.sym_stride;

    // ---- load carried state
    uint32_t k0 = sth[DH_ST_K];
    int32_t off = (int32_t) sth[DH_ST_OFF];
    const uint32_t tc = sth[DH_ST_TAIL];
    const uint32_t nv = tc + P.n;                       // length of the virtual input stream
    const uint32_t nf = nv >= NZ ? nv - NZ : 0u;        // filtered samples available this push
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < DH_VOLUME_RB_SIZE; j += DH_WAVE) S.vol_old[j] = st[DH_ST_VOL + j];
        for (uint32_t j = lane; j < DH_VARIANCE_SYMBOLS * sps; j += DH_WAVE) S.var_rb[j] = st[DH_ST_VAR + j];
    }
    DH_BARRIER();

    uint32_t p = 0;                                     // read position in the filtered stream
    uint32_t nsym = 0;
    bool overflow = false;
    const uint32_t max_run = (DH_FTILE - 2) / sps;      // symbols whose windows fit one FIR pass

    for (;;) {
        // ---- run planning (wave-uniform): symbols k0 .. k0+m-1 of the current variance block.
        // A symbol is produced iff available > sps + 1 (gfsk_demodulator.cpp:18-22).
        const int32_t step_off = (k0 == 0) ? off : 0;   // applied after the first symbol of a block
        uint32_t m = 0;
        {
            uint32_t lim = dh_min<uint32_t>(DH_VARIANCE_SYMBOLS - k0, max_run);
            if (lim > P.sym_cap - nsym) { lim = P.sym_cap - nsym; }
            for (uint32_t q = 0; q < lim; q++) {
                const uint32_t s = p + q * sps + (q > 0 ? (uint32_t) step_off : 0u);
                if (nf > s && nf - s > sps + 1) m = q + 1; else break;
            }
            if (m == 0) {
                if (lim == 0 && nf > p && nf - p > sps + 1) overflow = true;
                break;
            }
        }
        const uint32_t last_start = p + (m - 1) * sps + (m > 1 ? (uint32_t) step_off : 0u);
        const uint32_t need = last_start + sps - p;     // filtered samples [p, p+need) feed this run

        // ---- P1: stage raw samples V[p .. p+need+NZ) into the padded LDS window
        DH_FOR_LANES(lane) {
            for (uint32_t e = lane; e < need + NZ; e += DH_WAVE)
                S.xbuf[DH_XPAD(e)] = dh_virtual_sample(tail, tc, in, p + e);
            if (NZ > 0) {
                // lanes past the end of the run still read their whole window: keep it defined
                for (uint32_t e = need + NZ + lane; e < DH_FTILE + NZ; e += DH_WAVE) S.xbuf[DH_XPAD(e)] = 0.0f;
            }
        }
        DH_BARRIER();

        // ---- P2: FIR (or pass-through when there is no RRC stage)
        DH_FOR_LANES(lane) {
            if (NZ > 0) {
                if ((uint32_t) (lane * DH_FIR_L) < need) {
                    float o[DH_FIR_L];
                    dh_fir_lane<NZ, FAST>(P, S, lane, o);
#pragma unroll
                    for (int j = 0; j < DH_FIR_L; j++) S.fbuf[lane * DH_FIR_L + j] = o[j];
                }
            } else {
                for (uint32_t e = lane; e < need; e += DH_WAVE) S.fbuf[e] = S.xbuf[DH_XPAD(e)];
            }
        }
        DH_BARRIER();

        // ---- P3: symbol windows (gfsk_demodulator.cpp:28-35, 82-83)
        DH_FOR_LANES(lane) {
            for (uint32_t q = lane; q < m; q += DH_WAVE) {
                const uint32_t k = k0 + q;
                const uint32_t s = q * sps + (q > 0 ? (uint32_t) step_off : 0u);   // relative to p
                float sum = 0.0f, volume_sum = 0.0f;
                for (uint32_t i = 0; i < sps; i++) {
                    const float value = S.fbuf[s + i];
                    if (i >= P.lo && i < P.hi) sum += value;
                    volume_sum += value;
                    S.var_rb[k * sps + i] = value;
                }
                S.sum[q] = sum;
                S.vol_new[k] = volume_sum / (float) sps;
            }
        }
        DH_BARRIER();

        // ---- P4: sliding AGC min/max (calibrateAudio, gfsk_demodulator.cpp:109-116).
        // Four lanes run the four order-independent scans; min/max are exact in any order.
        const float FLT_MAX_ = 3.402823466e+38f, FLT_MIN_ = 1.175494351e-38f;   // sic: max seeds with FLT_MIN
        DH_FOR_LANES(lane) {
            const uint32_t k1 = k0 + m;                // one past the last symbol of the run
            if (lane == 0) {
                float c = FLT_MAX_;                    // old entries below k0 are in every window
                for (uint32_t j = 0; j < k0; j++) c = S.vol_old[j] < c ? S.vol_old[j] : c;
                for (uint32_t j = k0; j < k1; j++) { c = S.vol_new[j] < c ? S.vol_new[j] : c; S.mn_pre[j] = c; }
            } else if (lane == 1) {
                float c = FLT_MIN_;
                for (uint32_t j = 0; j < k0; j++) c = S.vol_old[j] > c ? S.vol_old[j] : c;
                for (uint32_t j = k0; j < k1; j++) { c = S.vol_new[j] > c ? S.vol_new[j] : c; S.mx_pre[j] = c; }
            } else if (lane == 2) {
                float c = FLT_MAX_;
                S.mn_suf[DH_VOLUME_RB_SIZE] = c;
                for (int j = DH_VOLUME_RB_SIZE - 1; j > (int) k0; j--) { c = S.vol_old[j] < c ? S.vol_old[j] : c; S.mn_suf[j] = c; }
            } else if (lane == 3) {
                float c = FLT_MIN_;
                S.mx_suf[DH_VOLUME_RB_SIZE] = c;
                for (int j = DH_VOLUME_RB_SIZE - 1; j > (int) k0; j--) { c = S.vol_old[j] > c ? S.vol_old[j] : c; S.mx_suf[j] = c; }
            }
        }
        DH_BARRIER();

        // ---- P5: thresholds + slice (gfsk_demodulator.cpp:88-106 / fsk_demodulator.cpp:89-99)
        DH_FOR_LANES(lane) {
            for (uint32_t q = lane; q < m; q += DH_WAVE) {
                const uint32_t k = k0 + q;
                const float a = S.mn_pre[k], b = S.mn_suf[k + 1];
                const float mn = b < a ? b : a;
                const float c = S.mx_pre[k], d = S.mx_suf[k + 1];
                const float mx = d > c ? d : c;
                const float center = (mx + mn) / 2.0f;
                const float average = S.sum[q] / (float) (P.hi - P.lo);
                uint8_t sym;
                if (P.levels == 4) {
                    const float umid = (float) ((double) (mx - center) * 0.625 + (double) center);
                    const float lmid = (float) ((double) (mn - center) * 0.625 + (double) center);
                    if (average > center) sym = average > umid ? 1 : 0;
                    else sym = average < lmid ? 3 : 2;
                } else {
                    sym = average > center ? (uint8_t) !P.invert : (uint8_t) (P.invert != 0);
                }
                syms[nsym + q] = sym;
            }
        }
        DH_BARRIER();

        // ---- P6: end of a variance block -> timing decision (gfsk_demodulator.cpp:41-80)
        int32_t new_off = 0;
        const bool block_done = (k0 + m == DH_VARIANCE_SYMBOLS);
        if (block_done) {
            DH_FOR_LANES(lane) {
                if ((uint32_t) lane < sps) {
                    float total = 0.0f;
                    for (int k = 0; k < DH_VARIANCE_SYMBOLS; k++) total += S.var_rb[k * sps + lane];
                    const double mean = (double) (total / (float) DH_VARIANCE_SYMBOLS);
                    double dsum = 0.0;
                    for (int k = 0; k < DH_VARIANCE_SYMBOLS; k++) {
                        const double diff = mean - (double) S.var_rb[k * sps + lane];
                        dsum += diff * diff;
                    }
                    S.variance[lane] = dsum / (double) DH_VARIANCE_SYMBOLS;
                }
            }
            DH_BARRIER();
            double vmin = S.variance[0]; uint32_t vmin_pos = 0;
            for (uint32_t i = 1; i < sps; i++) if (S.variance[i] < vmin) { vmin = S.variance[i]; vmin_pos = i; }
            if (vmin <= 0 || vmin > 5000000) {
            } else if (vmin_pos > 0 && vmin_pos < sps / 2) new_off = +1;
            else if (vmin_pos >= sps / 2 && vmin_pos < sps - 1) new_off = -1;
        }

        // ---- P7: commit the run (wave-uniform bookkeeping) and fold new volumes into the ring
        DH_FOR_LANES(lane) {
            for (uint32_t k = k0 + lane; k < k0 + m; k += DH_WAVE) S.vol_old[k] = S.vol_new[k];
        }
        DH_BARRIER();
        p = last_start + sps + ((k0 == 0 && m == 1) ? (uint32_t) step_off : 0u);
        nsym += m;
        if (k0 == 0) off = 0;                           // the pending step has been consumed (:36-38)
        k0 += m;
        if (block_done) { k0 = 0; off = new_off; }
    }

    // ---- write back state: rings, header, and the raw tail V[p .. nv)
    const uint32_t new_tc = nv - p;                    // = unread filtered samples + NZ
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < DH_VOLUME_RB_SIZE; j += DH_WAVE) st[DH_ST_VOL + j] = S.vol_old[j];
        for (uint32_t j = lane; j < DH_VARIANCE_SYMBOLS * sps; j += DH_WAVE) st[DH_ST_VAR + j] = S.var_rb[j];
        // tail may overlap its own source when p < tc: go through LDS
        for (uint32_t j = lane; j < new_tc && j < DH_TAIL_MAX; j += DH_WAVE) S.xbuf[j] = dh_virtual_sample(tail, tc, in, p + j);
    }
    DH_BARRIER();
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < new_tc && j < DH_TAIL_MAX; j += DH_WAVE) tail[j] = S.xbuf[j];
        if (DH_IS_LANE0(lane)) {
            sth[DH_ST_K] = k0;
            sth[DH_ST_OFF] = (uint32_t) off;
            sth[DH_ST_TAIL] = new_tc < DH_TAIL_MAX ? new_tc : DH_TAIL_MAX;
            sth[DH_ST_NSYM] += nsym;
            P.sym_count[ch] = nsym;
            if ((overflow || new_tc > DH_TAIL_MAX) && P.overflow) *P.overflow = 1u;
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
    uint32_t n, n_channels, nz;
    int32_t fast;
    double gain; float inv_gain;
    float taps[DH_MAX_NZ / 2 + 1];
};

template <int NZ, bool FAST>
DH_HD void dh_rrc_tile(const DhRrcParams& R, uint32_t ch, uint32_t tile, DhDspShared& S) {
    const float* in = R.in + (size_t) ch * R.in_stride;
    const float* hist = R.hist + (size_t) ch * NZ;
    float* out = R.out + (size_t) ch * R.out_stride;
    const uint32_t t0 = tile * DH_FTILE;
    const uint32_t cnt = dh_min<uint32_t>(DH_FTILE, R.n - t0);
    DH_FOR_LANES(lane) {
        // virtual stream = hist (NZ samples) ++ in; output t needs virtual [t, t+NZ]
        for (uint32_t e = lane; e < DH_FTILE + NZ; e += DH_WAVE) {
            const uint32_t v = t0 + e;
            float x = 0.0f;
            if (e < cnt + NZ) x = v < (uint32_t) NZ ? hist[v] : in[v - NZ];
            S.xbuf[DH_XPAD(e)] = x;
        }
    }
    DH_BARRIER();
    // FIR parameters live in a DhDspParams-shaped view for dh_fir_lane
    DH_FOR_LANES(lane) {
        if ((uint32_t) (lane * DH_FIR_L) < cnt) {
            float acc[DH_FIR_L];
#pragma unroll
            for (int j = 0; j < DH_FIR_L; j++) acc[j] = 0.0f;
            const int base = lane * DH_FIR_L;
#pragma unroll
            for (int t = 0; t < NZ + DH_FIR_L; t++) {
                const float x = S.xbuf[DH_XPAD(base + t)];
#pragma unroll
                for (int j = 0; j < DH_FIR_L; j++) {
                    const int i = t - j;
                    if (i >= 0 && i <= NZ) {
                        const float c = R.taps[i <= NZ / 2 ? i : NZ - i];
                        if (FAST) acc[j] = __builtin_fmaf(c, x, acc[j]);
                        else acc[j] = acc[j] + c * x;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < DH_FIR_L; j++) {
                const float y = FAST ? acc[j] * R.inv_gain : (float) ((double) acc[j] / R.gain);
                S.fbuf[base + j] = y;
            }
        }
    }
    DH_BARRIER();
    DH_FOR_LANES(lane) {
        for (uint32_t e = lane; e < cnt; e += DH_WAVE) out[t0 + e] = S.fbuf[e];     // coalesced store
    }
}
