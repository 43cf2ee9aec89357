// harness.cpp -- CPU wave-emulation build of the kernel bodies (TEST INFRASTRUCTURE ONLY).
//
// Compiles digiham_amd/csrc/*_core.hpp with a plain C++ compiler: DH_FOR_LANES becomes a
// 64-iteration loop, DH_BARRIER a no-op, "device memory" is the host heap and a "kernel launch"
// is a loop over workgroups.  It exports the same C ABI as libdigiham_amd.so so the CPU-only test
// tier can run the engine's orchestration and the exact wave algorithms against the oracle.
// It is built into tests/host_harness/libdh_hostemu.so, is never installed, and the digiham_amd
// package has no code path that loads it: the product fails loudly without the gfx950 library.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/digiham_amd.h"
#include "../../digiham_amd/csrc/kernels_core.hpp"
#include "../../digiham_amd/csrc/fec_tables.hpp"
#include "../../digiham_amd/csrc/rrc_taps.h"

namespace {

const DhFecTables& host_tables() {
    static DhFecTables* T = [] { auto* t = new DhFecTables; dh::build_fec_tables(*t); return t; }();
    return *T;
}

struct HostBackend {
    struct Scope {};
    Scope scope() const { return Scope(); }
    void close() {}
    int open(int, void*) { return 0; }
    void* alloc(size_t bytes) { return calloc(1, bytes ? bytes : 1); }
    void free(void* p) { ::free(p); }
    int zero(void* p, size_t bytes) { memset(p, 0, bytes); return 0; }
    int upload(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
    int copy_device(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
    int download2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows) {
        return upload2d(dst, dpitch, src, spitch, width, rows);
    }
    int upload2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows) {
        for (size_t r = 0; r < rows; r++) memcpy((char*) dst + r * dpitch, (const char*) src + r * spitch, width);
        return 0;
    }
    int download(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
    int sync() { return 0; }
    void timing_mark(int) {}
    void timing_next() {}
    int timing_enable(uint32_t) { return 0; }
    int timing_read(float*, float*, float*, uint32_t* n) { *n = 0; return 0; }
    int timing_read_split(float*, uint32_t*, uint32_t* n) { *n = 0; return 0; }
    bool overlap_pushes = false;                     // (a property of the GPU dispatcher; nothing to emulate)

    template <int NZ, bool FAST, int SPS, int KEEPF = 0> static void run_rrc_demod(const DhDspParams& P) {
        std::vector<float> lds(dh_dsp_shared_bytes(P.sps, NZ) / sizeof(float));     // exactly the device allocation
        DhDspShared S = dh_dsp_carve(lds.data(), P.sps, NZ);
        for (uint32_t ch = 0; ch < P.n_channels; ch++) dh_rrc_demod_channel<NZ, FAST, SPS, 0, KEEPF>(P, ch, S);
    }
    int launch_rrc_demod(const DhDspParams& P, uint32_t nz, bool fast) {
        if (P.filt_out) { if (P.sps == 10 && nz == 80) { if (fast) run_rrc_demod<80, false, 10, 2>(P); else run_rrc_demod<80, false, 10, 1>(P); return 0; } return -1; }
        if (P.sps == 10 && nz == 0) run_rrc_demod<0, false, 10>(P);
        else if (P.sps == 10 && nz == 80) { if (fast) run_rrc_demod<80, true, 10>(P); else run_rrc_demod<80, false, 10>(P); }
        else if (nz == 0 && P.sps == 40) run_rrc_demod<0, false, 40>(P);                 // the same instantiations as engine.hip
        else if (nz == 160 && P.sps == 20 && !fast) run_rrc_demod<160, false, 20>(P);
        else if (nz == 0) run_rrc_demod<0, false, 0>(P);
        else if (nz == 80) { if (fast) run_rrc_demod<80, true, 0>(P); else run_rrc_demod<80, false, 0>(P); }
        else if (nz == 160) { if (fast) run_rrc_demod<160, true, 0>(P); else run_rrc_demod<160, false, 0>(P); }
        else return -1;
        return 0;
    }
    template <int NZ, bool FAST> static void run_rrc_tiles(const DhRrcParams& R) {
        std::vector<float> lds(dh_dsp_shared_bytes(0, NZ) / sizeof(float));
        DhDspShared S = dh_dsp_carve(lds.data(), 0, NZ);
        const uint32_t tiles = (R.n + DH_FTILE - 1) / DH_FTILE;
        for (uint32_t ch = 0; ch < R.n_channels; ch++)
            for (uint32_t t = 0; t < tiles; t++) dh_rrc_tile<NZ, FAST>(R, ch, t, S);
    }
    int launch_rrc_tiles(const DhRrcParams& R, uint32_t nz, bool fast) {
        if (nz == 80) { if (fast) run_rrc_tiles<80, true>(R); else run_rrc_tiles<80, false>(R); }
        else if (nz == 160) { if (fast) run_rrc_tiles<160, true>(R); else run_rrc_tiles<160, false>(R); }
        else return -1;
        return 0;
    }
    int launch_rrc_generic(const DhRrcGenParams& G) {
        static float win[DH_GEN_WINDOW], taps[DH_MAX_NZ + 1];
        for (uint32_t ch = 0; ch < G.n_channels; ch++)
            for (uint32_t t = 0; t * DH_FTILE < G.n; t++) dh_rrc_generic_tile(G, ch, t, win, taps);
        return 0;
    }
    int launch_rrc_hist(float* hist, const float* in, size_t in_stride, uint32_t n, const uint32_t* n_per, uint32_t nz, uint32_t B) {
        float sh[DH_MAX_NZ];
        for (uint32_t ch = 0; ch < B; ch++) dh_rrc_hist_channel(hist, in, in_stride, n, n_per, nz, ch, sh);
        return 0;
    }
    int launch_chain(const DhDspParams& P, const DhDecParams& D, uint32_t nz, bool fast, int proto) {
        // same order of work as the device kernel: channel by channel, slicer then decoder
        if (P.sps != 10 || (nz != 0 && nz != 80) || (proto != DH_PROTO_DMR && proto != DH_PROTO_YSF)) return 1;
        // the tail split of the device's chain launches (engine.hip, k_chain): DH_TAIL_SPLIT = percent of a push the first
        // part takes.  Here the two parts of a channel simply run one after the other -- what is exercised is the part
        // arithmetic of the kernel bodies (offsets into the rows, appended symbols / frames / events).
        uint32_t pct = 0, pct2 = 0;
        if (const char* e = getenv("DH_TAIL_SPLIT")) {
            char* end = nullptr;
            const long v = strtol(e, &end, 10), w = end && *end == ',' ? strtol(end + 1, nullptr, 10) : 0;
            pct = v > 0 && v < 100 ? (uint32_t) v : 0u; pct2 = pct && w > v && w < 100 ? (uint32_t) w : 0u;
        }
        if (pct && nz == 80 && !fast && P.n >= 2) {
            const uint32_t b0 = std::max<uint32_t>(1u, (uint32_t) ((uint64_t) P.n * pct / 100u));
            const uint32_t b1 = pct2 ? std::max<uint32_t>(b0, (uint32_t) ((uint64_t) P.n * pct2 / 100u)) : 0u;
            const uint32_t lo[3] = { 0u, b0, b1 }, hi[3] = { b0, b1 ? b1 : 0xFFFFFFFFu, 0xFFFFFFFFu };
            std::vector<float> lds(dh_dsp_shared_bytes(P.sps, 80) / sizeof(float));
            DhDspShared S = dh_dsp_carve(lds.data(), P.sps, 80);
            DhDecShared* DS = new DhDecShared;
            // DH_TAIL_SPLIT_FORCE_FAIL = k: the later parts of the channels with ch % k == 1 "fail" their hand-over, and what the
            // device's fix-up launch does is done here -- the rest of the row in one piece behind the first part
            const uint32_t force = getenv("DH_TAIL_SPLIT_FORCE_FAIL") ? (uint32_t) strtoul(getenv("DH_TAIL_SPLIT_FORCE_FAIL"), nullptr, 10) : 0u;
            for (uint32_t ch = 0; ch < P.n_channels; ch++)
                for (uint32_t part = 0; part < (b1 ? 3u : 2u); part++) {
                    const uint32_t sym_base = part ? P.sym_count[ch] : 0u;
                    if (force && ch % force == 1u && part) {
                        dh_rrc_demod_channel<80, false, 10>(P, ch, S, lo[1], 0xFFFFFFFFu, sym_base);
                        if (proto == DH_PROTO_DMR) dh_dmr_channel(D, ch, *DS, sym_base, true); else dh_ysf_channel(D, ch, *DS, sym_base, true);
                        break;
                    }
                    dh_rrc_demod_channel<80, false, 10>(P, ch, S, lo[part], hi[part], sym_base);
                    if (proto == DH_PROTO_DMR) dh_dmr_channel(D, ch, *DS, sym_base, part != 0); else dh_ysf_channel(D, ch, *DS, sym_base, part != 0);
                }
            delete DS;
            return 0;
        }
        if (launch_rrc_demod(P, nz, fast)) return -1;
        return launch_decoder(D, proto) ? -1 : 0;
    }
    int launch_decoder(const DhDecParams& P, int proto) {
        DhDecShared* S = new DhDecShared;
        for (uint32_t ch = 0; ch < P.n_channels; ch++) {
            if (proto == DH_PROTO_DMR) dh_dmr_channel(P, ch, *S); else if (proto == DH_PROTO_YSF) dh_ysf_channel(P, ch, *S); else if (proto == DH_PROTO_NXDN) dh_nxdn_channel(P, ch, *S);
            else if (proto == DH_PROTO_POCSAG) dh_pocsag_channel(P, ch, *S); else dh_dstar_channel(P, ch, *S);
        }
        delete S;
        return 0;
    }
    int launch_init_state(uint32_t* dsp_state, size_t state_words, uint32_t tail0, uint32_t* dec_state, uint32_t slot_filter, uint32_t B) {
        for (uint32_t ch = 0; ch < B; ch++) dh_init_state_channel(dsp_state, state_words, tail0, dec_state, slot_filter, ch);
        return 0;
    }
    int launch_set_slot_filter(uint32_t* dec_state, uint32_t filter, uint32_t B) {
        for (uint32_t ch = 0; ch < B; ch++) dh_set_slot_filter_channel(dec_state, filter, ch);
        return 0;
    }
};

}  // namespace

static int dh_be_device_count() { return 0; }
static const char* dh_be_last_error() { return "host emulation"; }
static int dh_be_alloc(int, size_t bytes, void** out) { *out = calloc(1, bytes ? bytes : 1); return *out ? 0 : DH_ENOMEM; }
static int dh_be_free(void* p) { free(p); return 0; }
static int dh_be_copy(void* dst, const void* src, size_t bytes, int) { if (bytes) memcpy(dst, src, bytes); return 0; }
static int dh_be_fec_block(int code, void* words, uint8_t* ok, size_t n, void*) {
    for (size_t i = 0; i < n; i++) dh_fec_block_item(host_tables(), code, words, ok, i);
    return 0;
}
static int dh_be_bptc(const uint8_t* in, uint8_t* out, uint8_t* ok, size_t n, void*) {
    for (size_t i = 0; i < n; i++) dh_bptc_item(host_tables(), in, out, ok, i);
    return 0;
}
static int dh_be_trellis(const uint8_t* in, size_t in_stride, int n_dibits, uint8_t* out, size_t out_stride, uint8_t* metric, size_t n, void*) {
    DhDecShared* S = new DhDecShared;
    for (size_t w = 0; w < (n + 3) / 4; w++) dh_trellis_wave(in, in_stride, n_dibits, out, out_stride, metric, n, w, *S);
    delete S;
    return 0;
}
static int dh_be_crc16(const uint8_t* in, size_t stride, int count, uint16_t* out, size_t n, void*) {
    for (size_t i = 0; i < n; i++) dh_crc16_item(in, stride, count, out, i);
    return 0;
}
static int dh_be_whitening(const uint8_t* in, uint8_t* out, size_t stride, int n_bits, size_t n, void*) {
    for (size_t i = 0; i < n; i++) dh_whitening_item(in, out, stride, n_bits, i);
    return 0;
}
static int dh_be_dvfilter(const int16_t* in, int16_t* out, float* state, size_t B, size_t stride, size_t n, void*) {
    for (size_t ch = 0; ch < B; ch++) dh_dvfilter_channel(in + ch * stride, out + ch * stride, state + ch * 22, n);
    return 0;
}

static int dh_be_div_gain(const float* in, float* out, size_t n, int narrow, void*) {
    const double gain = narrow ? DH_RRC_NARROW_GAIN : DH_RRC_WIDE_GAIN, rgain = 1.0 / gain;
    for (size_t i = 0; i < n; i++) out[i] = dh_div_gain(in[i], gain, rgain);
    return 0;
}

static int dh_be_frontend(const int16_t* in, size_t in_stride, float* out, size_t out_stride, float* state, size_t B, size_t n, int mode, int dcblock, void*) {
    for (size_t ch = 0; ch < B; ch++) dh_frontend_channel(in + ch * in_stride, out + ch * out_stride, state + ch * DH_FE_STATE_WORDS, n, mode, dcblock);
    return 0;
}

// (harness: products of halves are exact in f32; a k-ordered chain of rounded additions is one of the behaviours (H1) allows)
static int dh_be_mfma_f16(const uint16_t* a, const uint16_t* b, const float* c, float* d, size_t tiles, void*) {
    for (size_t t = 0; t < tiles; t++) for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
        float acc = c[t * 256 + m * 16 + n];
        for (int k = 0; k < 32; k++) acc = __builtin_fmaf(dh_f16_value(a[t * 512 + m * 32 + k]), dh_f16_value(b[t * 512 + k * 16 + n]), acc);
        d[t * 256 + m * 16 + n] = acc;
    }
    return 0;
}
static int dh_be_f16_split(const float* in, uint16_t* h1, uint16_t* h2, size_t n, float scale, void*) {
    for (size_t i = 0; i < n; i += 4) {
        dh_f4 v; v.x = in[i]; v.y = i + 1 < n ? in[i + 1] : 0.0f; v.z = i + 2 < n ? in[i + 2] : 0.0f; v.w = i + 3 < n ? in[i + 3] : 0.0f;
        uint16_t a[4], b[4];
        dh_f16_split4(v, scale, a, b);
        for (int j = 0; j < 4 && i + j < n; j++) { h1[i + j] = a[j]; h2[i + j] = b[j]; }
    }
    return 0;
}
static int dh_be_copy_kernel(const void* src, void* dst, size_t n_bytes, void*) { if (dst) __builtin_memcpy(dst, src, n_bytes); return 0; }
static int dh_be_div_const(const float* in, float* out, size_t n, unsigned divisor, void*) {
    const float d = (float) divisor, r = 1.0f / d;
    for (size_t i = 0; i < n; i++) out[i] = dh_div_const(in[i], d, r);
    return 0;
}

#define DH_BACKEND HostBackend
#include "../../digiham_amd/csrc/abi_impl.hpp"
