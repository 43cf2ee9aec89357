// kernels_core.hpp -- bodies of the small kernels (state init, RRC history carry, batch FEC items,
// batch Viterbi, digital-voice filter), shared by the gfx950 kernels in engine.hip and by the CPU
// test harness.
#pragma once

#include "dsp_core.hpp"
#include "frontend_core.hpp"
#include "decoder_core.hpp"

// ---- engine state initialisation: what the reference's constructors leave behind ------------
DH_HD void dh_init_state_channel(uint32_t* dsp_state, size_t state_words, uint32_t tail0,
                                 uint32_t* dec_state, uint32_t slot_filter, uint32_t ch) {
    if (dsp_state) {
        uint32_t* s = dsp_state + (size_t) ch * state_words;
        s[DH_ST_K] = 0; s[DH_ST_OFF] = 0; s[DH_ST_NSYM] = 0;
        s[DH_ST_TAIL] = tail0;       // RrcFilter's delay line: nz zero samples (canonical zero heap)
    }
    if (dec_state) {
        uint32_t* d = dec_state + (size_t) ch * DH_DEC_STATE_WORDS;
        for (int i = 0; i < DH_DEC_STATE_WORDS; i++) d[i] = 0;
        d[DS_SLOT_FILTER_DECODER] = slot_filter;     // Dmr::Decoder::slotFilter = 3 by default (dmr_decoder.hpp:16)
        d[DS_SLOT_FILTER] = slot_filter;
    }
}

// Dmr::Decoder::setSlotFilter + FramePhase::setSlotFilter (dmr_decoder.cpp:9-15, dmr_phase.cpp:341-345)
DH_HD void dh_set_slot_filter_channel(uint32_t* dec_state, uint32_t filter, uint32_t ch) {
    uint32_t* d = dec_state + (size_t) ch * DH_DEC_STATE_WORDS;
    d[DS_SLOT_FILTER_DECODER] = filter;
    if (d[DS_PHASE] == 1) {
        d[DS_SLOT_FILTER] = filter;
        if ((((int) d[DS_ACTIVE_SLOT] + 1) & (int) filter) == 0) d[DS_ACTIVE_SLOT] = (uint32_t) -1;
    }
}

// history of the stand-alone RRC stage: the last nz samples of (hist ++ in[0..n))
// (one channel per wavefront; staged through LDS because old and new history overlap when n < nz)
DH_HD void dh_rrc_hist_channel(float* hist, const float* in, size_t in_stride, uint32_t n_max, const uint32_t* n_per, uint32_t nz, uint32_t ch, float* sh) {
    const uint32_t n = n_per ? dh_min<uint32_t>(n_per[ch], n_max) : n_max;
    float* h = hist + (size_t) ch * nz;
    const float* x = in + (size_t) ch * in_stride;
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < nz; j += DH_WAVE) {
            const uint32_t v = n + j;                   // index into (old hist ++ in)
            sh[j] = v < nz ? h[v] : x[v - nz];
        }
    }
    DH_BARRIER();
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < nz; j += DH_WAVE) h[j] = sh[j];
    }
    DH_BARRIER();
}

// Stand-alone FIR with the caller's coefficient table (DH_RRC_CUSTOM): any shape, 1 <= nz <= 160, exact arithmetic
// only.  One 1024-output tile of one channel per wavefront; lane l produces outputs l, l + 64, ... (LDS reads of a
// tap step are then 64 consecutive words, so the window needs no padding) and every tap is an LDS broadcast.
// A compatibility path -- the reference only ever instantiates its two built-in designs -- not a tuned one.
struct DhRrcGenParams {
    const float* in; size_t in_stride;
    float* out; size_t out_stride;
    const float* hist;                                  // [B][nz] previous inputs (zeros after reset)
    const float* taps;                                  // [nz + 1] device copy of the caller's table
    uint32_t n, n_channels, nz;
    double gain;
};
#define DH_GEN_WINDOW (DH_FTILE + DH_MAX_NZ)
DH_HD void dh_rrc_generic_tile(const DhRrcGenParams& G, uint32_t ch, uint32_t tile, float* win /* [DH_GEN_WINDOW] */, float* taps /* [DH_MAX_NZ + 1] */) {
    const float* in = G.in + (size_t) ch * G.in_stride;
    const float* hist = G.hist + (size_t) ch * G.nz;
    float* out = G.out + (size_t) ch * G.out_stride;
    const uint32_t t0 = tile * DH_FTILE, nz = G.nz;
    const uint32_t cnt = dh_min<uint32_t>(DH_FTILE, G.n - t0);
    DH_FOR_LANES(lane) {
        for (uint32_t e = lane; e < cnt + nz; e += DH_WAVE) {       // virtual stream = hist (nz samples) ++ in
            const uint32_t v = t0 + e;
            win[e] = v < nz ? hist[v] : in[v - nz];
        }
        for (uint32_t i = lane; i <= nz; i += DH_WAVE) taps[i] = G.taps[i];
    }
    DH_BARRIER();
    DH_FOR_LANES(lane) {
        float acc[DH_FTILE / DH_WAVE];
        for (int j = 0; j < DH_FTILE / DH_WAVE; j++) acc[j] = 0.0f;
        for (uint32_t i = 0; i <= nz; i++) {
            const float c = taps[i];
            for (int j = 0; j < DH_FTILE / DH_WAVE; j++) {
                const uint32_t o = (uint32_t) lane + DH_WAVE * (uint32_t) j;
                if (o < cnt) { const float p = c * win[o + i]; acc[j] = acc[j] + p; }      // rounded product, rounded sum (-ffp-contract=off)
            }
        }
        for (int j = 0; j < DH_FTILE / DH_WAVE; j++) {
            const uint32_t o = (uint32_t) lane + DH_WAVE * (uint32_t) j;
            if (o < cnt) out[t0 + o] = dh_div_gain_exact(acc[j], G.gain);
        }
    }
    DH_BARRIER();
}

// ---- batch FEC: one word per lane --------------------------------------------------------------
DH_HD void dh_fec_block_item(const DhFecTables& T, int code, void* words, uint8_t* ok, size_t i) {
    bool r = false;
    switch (code) {
        case 0: { uint32_t w = ((uint8_t*) words)[i]; r = dh_block_decode(T.h74, T.lut_h74, w); ((uint8_t*) words)[i] = (uint8_t) w; break; }
        case 1: { uint32_t w = ((uint16_t*) words)[i]; r = dh_block_decode(T.h139, T.lut_h139, w); ((uint16_t*) words)[i] = (uint16_t) w; break; }
        case 2: { uint32_t w = ((uint16_t*) words)[i]; r = dh_block_decode(T.h1511, T.lut_h1511, w); ((uint16_t*) words)[i] = (uint16_t) w; break; }
        case 3: { uint32_t w = ((uint16_t*) words)[i]; r = dh_block_decode(T.h1611, T.lut_h1611, w); ((uint16_t*) words)[i] = (uint16_t) w; break; }
        case 4: { uint32_t w = ((uint16_t*) words)[i]; r = dh_block_decode(T.qr, T.lut_qr, w); ((uint16_t*) words)[i] = (uint16_t) w; break; }
        case 5: { uint32_t w = ((uint32_t*) words)[i]; r = dh_block_decode(T.g208, T.lut_g208, w); ((uint32_t*) words)[i] = w; break; }
        case 6: { uint32_t w = ((uint32_t*) words)[i]; r = dh_block_decode(T.g2412, T.lut_g2412, w); ((uint32_t*) words)[i] = w; break; }
        case 7: { uint32_t w = ((uint32_t*) words)[i]; r = dh_block_decode(T.bch3121, T.lut_bch3121, w); ((uint32_t*) words)[i] = w; break; }
    }
    ok[i] = r ? 1 : 0;
}

// (the lane-local decoder of the DMR chain, dh_dmr_bptc_lane -- columns bit-sliced over row words: the batch entry runs the same code, so
// the reference's golden vectors pin it directly)
DH_HD void dh_bptc_item(const DhFecTables& T, const uint8_t* in, uint8_t* out, uint8_t* ok, size_t i) {
    uint32_t w[7] = { 0, 0, 0, 0, 0, 0, 0 };                      // the 196 received bits, first bit on top of w[0]
    for (int b = 0; b < 25; b++) w[b >> 2] |= (uint32_t) in[i * 25 + b] << (24 - 8 * (b & 3));
    uint32_t o[3];
    const uint32_t* wp = w;
    const bool r = dh_dmr_bptc_lane(T, [wp](int k) -> uint32_t { return (wp[k >> 5] >> (31 - (k & 31))) & 1u; }, o);
    for (int b = 0; b < 12; b++) out[i * 12 + b] = r ? (uint8_t) (o[b >> 2] >> (8 * (b & 3))) : (uint8_t) 0;
    ok[i] = r ? 1 : 0;
}

DH_HD void dh_crc16_item(const uint8_t* in, size_t stride, int count, uint16_t* out, size_t i) {
    out[i] = dh_crc16(in + i * stride, count);
}

DH_HD void dh_whitening_item(const uint8_t* in, uint8_t* out, size_t stride, int n_bits, size_t i) {
    uint8_t a[32], b[32];
    const int nb = (n_bits + 7) / 8;
    for (int k = 0; k < nb; k++) a[k] = in[i * stride + k];
    dh_whiten(a, b, n_bits);
    for (int k = 0; k < nb; k++) out[i * stride + k] = b[k];
}

// batch Viterbi: wavefront `wave` decodes codewords 4*wave .. 4*wave+3
DH_HD void dh_trellis_wave(const uint8_t* in, size_t in_stride, int n_dibits, uint8_t* out, size_t out_stride,
                           uint8_t* metric, size_t n, size_t wave, DhDecShared& S) {
    const int nin = (n_dibits + 3) / 4, nout = (n_dibits + 7) / 8;
    int sizes[4];
    for (int g = 0; g < 4; g++) sizes[g] = (wave * 4 + g < n) ? n_dibits : 0;
    DH_FOR_LANES(lane) {
        for (int e = lane; e < 4 * nin; e += DH_WAVE) {
            const int g = e / nin, b = e % nin;
            const size_t cw = wave * 4 + g;
            S.vit_in[g][b] = dh_vit_word(cw < n ? in[cw * in_stride + b] : (uint8_t) 0);
        }
    }
    DH_BARRIER();
    // 100-dibit codewords (FICH, V/D2 DCH): the syndrome check of the YSF decoder (dh_ysf_clean100), one codeword per lane -- a codeword
    // with a zero syndrome, or with one wrong dibit away from the block's ends, leaves with its message and metric; the full decoder only
    // sees the others
    uint64_t clean = 0;
    if (n_dibits == 100) {
        DH_FOR_LANES(lane) {
            bool ok = false;
            if (lane < 4 && wave * 4 + (size_t) lane < n) {
                uint32_t h[4] = { 0, 0, 0, 0 }, l[4] = { 0, 0, 0, 0 }, o[4];
                for (int i = 0; i < 100; i++) {
                    const uint32_t d = (S.vit_in[lane][i >> 2] >> (8 * (i & 3))) & 3u;     // one dibit per byte (dh_vit_word)
                    h[i >> 5] |= (d >> 1) << (i & 31); l[i >> 5] |= (d & 1u) << (i & 31);
                }
                uint32_t mt = 0;
                ok = !dh_ysf_clean100(h, l, o, &mt);
                if (ok) {
                    const size_t cw = wave * 4 + (size_t) lane;
                    for (int b = 0; b < 13; b++) out[cw * out_stride + b] = (uint8_t) (o[b >> 2] >> (8 * (b & 3)));
                    metric[cw] = (uint8_t) mt;
                }
            }
            DH_BALLOT_ACC(clean, ok, lane);
        }
        for (int g = 0; g < 4; g++) if ((clean >> g) & 1ull) sizes[g] = 0;
    }
    if (sizes[0] | sizes[1] | sizes[2] | sizes[3]) dh_viterbi_wave(S, sizes);
    DH_FOR_LANES(lane) {
        for (int e = lane; e < 4 * nout; e += DH_WAVE) {
            const int g = e / nout, b = e % nout;
            const size_t cw = wave * 4 + g;
            if (cw < n && !((clean >> g) & 1ull)) out[cw * out_stride + b] = S.vit_out[g][b];
        }
        if (lane < 4 && wave * 4 + lane < n && !((clean >> lane) & 1ull)) metric[wave * 4 + lane] = S.vit_best_metric[lane];
    }
    DH_BARRIER();
}

// ---- digital voice filter: one channel per lane (strictly sequential recurrence) ------------
// reference: src/digitalvoice_filter/digitalvoice_filter.cpp:6-10,34-45 (GAIN 5)
DH_HD void dh_dvfilter_channel(const int16_t* in, int16_t* out, float* st, size_t n) {
    float xv[11], yv[11];
    for (int i = 0; i < 11; i++) { xv[i] = st[i]; yv[i] = st[11 + i]; }
    for (size_t t = 0; t < n; t++) {
        const float sample = (float) in[t] / 32767.0f;
        for (int i = 0; i < 10; i++) xv[i] = xv[i + 1];
        xv[10] = sample / 5.0f;
        for (int i = 0; i < 10; i++) yv[i] = yv[i + 1];
        const float ff = (xv[10] - xv[0]) + 5.0f * (xv[2] - xv[8]) + 10.0f * (xv[6] - xv[4]);
        double acc = (double) ff;
        acc = acc + (0.1254306222 * (double) yv[0]);
        acc = acc + (0.1285714097 * (double) yv[1]);
        acc = acc + (-0.8106454980 * (double) yv[2]);
        acc = acc + (-0.7664515771 * (double) yv[3]);
        acc = acc + (2.1846187758 * (double) yv[4]);
        acc = acc + (1.8106678608 * (double) yv[5]);
        acc = acc + (-3.1465011600 * (double) yv[6]);
        acc = acc + (-2.0391991609 * (double) yv[7]);
        acc = acc + (2.4873968618 * (double) yv[8]);
        acc = acc + (1.0249072542 * (double) yv[9]);
        yv[10] = (float) acc;
        const float v = yv[10] * 32767.0f;
        // (short) of a float on x86-64: cvttss2si to int32 ("integer indefinite" 0x80000000 when out of
        // range or NaN), then the low 16 bits
        int32_t iv;
        if (!(v >= -2147483648.0f && v < 2147483648.0f)) iv = (int32_t) 0x80000000u;
        else iv = (int32_t) v;
        out[t] = (int16_t) (uint16_t) ((uint32_t) iv & 0xFFFFu);
    }
    for (int i = 0; i < 11; i++) { st[i] = xv[i]; st[11 + i] = yv[i]; }
}
