// engine_impl.hpp -- the streaming engine behind the C ABI: buffer layout in HBM, per-push kernel
// sequence, output views.  Written against a small backend interface so that the same
// orchestration is exercised by the gfx950 build (engine.hip: HIP allocations + kernel launches)
// and by the CPU-only test harness (tests/host_harness: malloc + lane loops).
//
// HBM layout per engine (B = n_channels, all channel-major, one contiguous row per channel):
//   in            [B][stride] f32      caller's buffer (device-resident), read once per push
//   dsp_state     [B][state_words] u32 slicer rings + timing state + raw tail (dsp_core.hpp)
//   syms          [B][sym_stride] u8   this push's symbols (dibits)
//   sym_carry     [B][512] u8          symbols the decoder left unread (< one frame)
//   dec_state     [B][64] u32          decoder phase state (DH_DEC_STATE_WORDS, decoder_core.hpp)
//   frames        [B][out_cap] u8      decoder output bytes of this push
//   events        [B][ev_cap] dh_event decoder events of this push
//   filtered      [B][max_samples] f32 only with DH_FLAG_KEEP_FILTERED (unfused path)
//   rrc_hist      [B][nz] f32          only with DH_FLAG_KEEP_FILTERED
#pragma once

#include <string.h>
#include <new>

#include "../../include/digiham_amd.h"
#include "dsp_core.hpp"
#include "decoder_core.hpp"
#include "fec_tables.hpp"
#include "rrc_taps.h"

namespace dh {

struct Layout {
    uint32_t B, max_samples, sps, nz, lo, hi;
    int rrc, demod, proto;
    uint32_t flags;
    uint32_t sym_cap, out_cap, ev_cap;
    size_t sym_stride, state_words;
    bool fused;          // RRC folded into the slicer kernel (no filtered signal in HBM)
    int fused_keep;      // ... which also delivers the filtered samples (DH_FLAG_KEEP_FILTERED | DH_FLAG_ONE_LAUNCH, wide filter, sps 10): one launch; 2 = with DH_FLAG_FAST_FIR
};

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

inline int make_layout(const dh_engine_config& c, Layout& L) {
    if (c.struct_size != sizeof(dh_engine_config) && c.struct_size != DH_ENGINE_CONFIG_V1_SIZE) return DH_EINVAL;
    if (c.n_channels == 0 || c.max_samples == 0) return DH_EINVAL;
    if (c.rrc < DH_RRC_NONE || c.rrc > DH_RRC_CUSTOM) return DH_EINVAL;
    if (c.rrc == DH_RRC_CUSTOM) {
        if (c.struct_size != sizeof(dh_engine_config) || !c.rrc_taps || c.rrc_nzeros < 1 || c.rrc_nzeros > DH_MAX_NZ) return DH_EINVAL;
        if (!(c.rrc_gain == c.rrc_gain) || c.rrc_gain == 0.0 || (c.flags & DH_FLAG_FAST_FIR)) return DH_EINVAL;      // exact arithmetic only
    }
    if (c.demod != DH_DEMOD_NONE && c.demod != DH_DEMOD_FSK2 && c.demod != DH_DEMOD_GFSK4) return DH_EINVAL;
    if (c.proto < DH_PROTO_NONE || c.proto > DH_PROTO_DSTAR) return DH_EINVAL;
    if (c.demod != DH_DEMOD_NONE && (c.sps < 3 || c.sps > DH_MAX_SPS)) return DH_EINVAL;
    if (c.demod == DH_DEMOD_NONE && c.rrc == DH_RRC_NONE && c.proto == DH_PROTO_NONE) return DH_EINVAL;
    L.B = c.n_channels; L.max_samples = c.max_samples; L.sps = c.demod ? c.sps : 1;
    L.rrc = c.rrc; L.demod = c.demod; L.proto = c.proto; L.flags = c.flags;
    L.nz = c.rrc == DH_RRC_WIDE ? DH_RRC_WIDE_NZEROS : c.rrc == DH_RRC_NARROW ? DH_RRC_NARROW_NZEROS : c.rrc == DH_RRC_CUSTOM ? c.rrc_nzeros : 0;
    // GfskDemodulator ctor: lowestEval = round(sps/3), highestEval = round(2*sps/3)  (gfsk_demodulator.cpp:8-9)
    L.lo = (uint32_t) (int) __builtin_roundf((float) L.sps / 3);
    L.hi = (uint32_t) (int) __builtin_roundf((float) L.sps * 2 / 3);
    // DH_FLAG_ONE_LAUNCH: the error-bounded slicer kernel filters on the matrix cores, decides every dibit as the reference does and
    // stores the filtered samples it holds in LDS -- one kernel instead of RRC tiles + slicer (8.1 instead of 12.1 bytes per sample
    // through HBM).  Its floats are the split-f16 FIR's: 1.1e-6 of max(|ref|, rms) measured, outside the 1e-6 of BASELINE configs[1].
    // With DH_FLAG_FAST_FIR (floats within 1e-6 asked for) the same kernel filters with the f32 FMA chain on the matrix cores instead
    // (v_mfma_f32_16x16x4_f32, the arithmetic of the two-kernel FMA pair: 5e-7 measured) and still decides every dibit as the reference does:
    // BASELINE configs[1] in one launch (round 6; before, the flag fell back to the two-kernel pair).
    L.fused_keep = ((L.flags & DH_FLAG_KEEP_FILTERED) && (L.flags & DH_FLAG_ONE_LAUNCH) && L.rrc == DH_RRC_WIDE && L.demod != DH_DEMOD_NONE && L.sps == 10
                    && !(L.flags & DH_FLAG_EXACT_FIR)) ? ((L.flags & DH_FLAG_FAST_FIR) ? 2 : 1) : 0;
    L.fused = L.rrc != DH_RRC_NONE && L.rrc != DH_RRC_CUSTOM && L.demod != DH_DEMOD_NONE && (!(L.flags & DH_FLAG_KEEP_FILTERED) || L.fused_keep);
    // a symbol consumes at least sps-1 samples
    L.sym_cap = L.demod ? (L.max_samples + dh_tail_max(L.sps)) / (L.sps - 1) + 4 : L.max_samples;
    L.sym_stride = round_up(L.sym_cap, 64);
    L.state_words = round_up(dh_state_words(L.sps), 16);
    const uint32_t max_syms = dh_carry_max(L.proto) + L.sym_cap;
    if (L.proto == DH_PROTO_DMR) { L.out_cap = (max_syms / 144 + 1) * 27; L.ev_cap = (max_syms / 144 + 1) * 4 + 8; }
    else if (L.proto == DH_PROTO_YSF) { L.out_cap = (max_syms / 480 + 1) * 95; L.ev_cap = (max_syms / 480 + 1) * 5 + 8; }
    else if (L.proto == DH_PROTO_NXDN) { L.out_cap = (max_syms / 192 + 1) * 36; L.ev_cap = (max_syms / 192 + 1) * 7 + 8; }
    else if (L.proto == DH_PROTO_POCSAG) { L.out_cap = max_syms / 2 + 256; L.ev_cap = max_syms / 32 + 8; }
    else if (L.proto == DH_PROTO_DSTAR) { L.out_cap = (max_syms / 96 + 1) * 9; L.ev_cap = max_syms / 32 + 16; }
    else { L.out_cap = 0; L.ev_cap = 0; }
    L.out_cap = round_up(L.out_cap, 64);
    return DH_OK;
}

inline void fill_taps(int rrc, float* half, double* gain) {
    float full[DH_RRC_MAX_TAPS];
    const int narrow = rrc == DH_RRC_NARROW;
    dh_rrc_expand_taps(narrow, full);
    const int nz = narrow ? DH_RRC_NARROW_NZEROS : DH_RRC_WIDE_NZEROS;
    for (int i = 0; i <= nz / 2; i++) half[i] = full[i];
    *gain = narrow ? DH_RRC_NARROW_GAIN : DH_RRC_WIDE_GAIN;
}

// Backend interface (duck-typed):
//   void* alloc(size_t bytes); void free(void*);
//   int zero(void* p, size_t bytes);                       // async on the engine stream
//   int upload(void* dst, const void* src, size_t bytes);  // host -> device, async
//   int copy_device(void* dst, const void* src, size_t bytes);  // device -> device, async on the engine stream
//   int download2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t rows);  // synchronous
//   int launch_chain(const DhDspParams&, const DhDecParams&, uint32_t nz, bool fast, int proto);   // 1 = unavailable
//   int upload2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, size_t rows);
//   int download(void* dst, const void* src, size_t bytes);// device -> host, synchronous
//   int sync();
//   int launch_rrc_demod(const DhDspParams&, uint32_t nz, bool fast);
//   int launch_rrc_tiles(const DhRrcParams&, uint32_t nz, bool fast);
//   int launch_rrc_hist(float* hist, const float* in, size_t in_stride, uint32_t n, uint32_t nz, uint32_t B);
//   int launch_decoder(const DhDecParams&, int proto);
//   int launch_init_state(uint32_t* dsp_state, size_t state_words, uint32_t tail0, uint32_t* dec_state, uint32_t slot_filter, uint32_t B);
//   int launch_set_slot_filter(uint32_t* dec_state, uint32_t filter, uint32_t B);
//   void timing_mark(int k); void timing_next();            // optional per-push stage timestamps (k = 0..3)
//   int timing_enable(uint32_t max_pushes); int timing_read(float* rrc, float* slicer, float* decoder, uint32_t* n);
//   int timing_read_split(float* first_ms, uint32_t* first_channels, uint32_t* n); bool overlap_pushes;
template <class BE>
struct Engine {
    BE be;
    Layout L;
    uint32_t slot_filter;
    // device buffers
    uint32_t* dsp_state = nullptr;
    uint8_t* syms = nullptr;
    uint32_t* sym_count = nullptr;
    uint8_t* sym_carry = nullptr;        // [B][DH_SYM_CARRY_MAX] symbols the decoder has not consumed yet
    uint32_t* dec_state = nullptr;
    uint8_t* frames = nullptr; uint32_t* frame_count = nullptr;
    dh_event* events = nullptr; uint32_t* ev_count = nullptr;
    uint32_t* overflow = nullptr;
    float* filtered = nullptr; float* rrc_hist = nullptr;
    float* staging = nullptr;            // push_host staging [B][max_samples]
    uint32_t* staging_counts = nullptr;  // ... and the per-channel counts of a ragged host push
    uint32_t* counts_copy = nullptr;     // the counts of the last ragged push of an engine that keeps its filtered samples
    const uint32_t* last_counts = nullptr;       // per-channel sample counts of the last push (device), or null
    DhFecTables* tables = nullptr;
    uint32_t* zero_counts = nullptr;
    uint32_t last_n = 0;
    DhDspParams dsp{}; DhRrcParams rrcp{}; DhDecParams dec{};
    float* custom_taps = nullptr; double custom_gain = 0.0;      // DH_RRC_CUSTOM: device copy of the caller's table
    uint32_t* tapfrag = nullptr; float err_coef_f16 = 0.0f;      // split-f16 FIR of the wide filter: per-lane tap fragments (dh_f16_tap_fragments)

    int init(const dh_engine_config& c) {
        int rc = make_layout(c, L);
        if (rc) return rc;
        be.overlap_pushes = (L.flags & DH_FLAG_OVERLAP_PUSHES) != 0;
        slot_filter = c.slot_filter;
        const size_t B = L.B;
#define DH_ALLOC(ptr, type, count) do { ptr = (type*) be.alloc(sizeof(type) * (size_t) (count)); if (!ptr) return DH_ENOMEM; } while (0)
        DH_ALLOC(overflow, uint32_t, 16);
        DH_ALLOC(sym_count, uint32_t, B);
        if (L.demod) DH_ALLOC(dsp_state, uint32_t, B * L.state_words);
        if (L.demod) DH_ALLOC(syms, uint8_t, B * L.sym_stride);
        if (L.proto) {
            DH_ALLOC(dec_state, uint32_t, B * DH_DEC_STATE_WORDS);
            DH_ALLOC(sym_carry, uint8_t, B * dh_carry_max(L.proto));
            DH_ALLOC(frames, uint8_t, B * (size_t) L.out_cap);
            DH_ALLOC(frame_count, uint32_t, B);
            DH_ALLOC(ev_count, uint32_t, B);
            if (!(L.flags & DH_FLAG_NO_EVENTS)) DH_ALLOC(events, dh_event, B * (size_t) L.ev_cap);
            DH_ALLOC(tables, DhFecTables, 1);
            DhFecTables* host = new (std::nothrow) DhFecTables;
            if (!host) return DH_ENOMEM;
            build_fec_tables(*host);
            rc = be.upload(tables, host, sizeof(DhFecTables));
            be.sync();
            delete host;
            if (rc) return rc;
        }
        if (L.rrc && (!L.fused || L.fused_keep)) DH_ALLOC(filtered, float, B * (size_t) L.max_samples);
        if (L.rrc && !L.fused) DH_ALLOC(rrc_hist, float, B * (size_t) L.nz);
        if (L.rrc == DH_RRC_CUSTOM) {
            DH_ALLOC(custom_taps, float, L.nz + 1);
            custom_gain = c.rrc_gain;
            if (be.upload(custom_taps, c.rrc_taps, sizeof(float) * (L.nz + 1)) || be.sync()) return DH_EDEVICE;   // the caller's table may go away
        }
        if (L.fused && (L.nz == 80 || L.nz == 160)) {
            DhF16Taps* F = new (std::nothrow) DhF16Taps;
            if (!F) return DH_ENOMEM;
            float half[DH_MAX_NZ / 2 + 1]; double gain;
            fill_taps(L.rrc, half, &gain);
            dh_f16_tap_fragments(half, L.nz, *F);
            err_coef_f16 = dh_f16_error_coefficient(*F, L.nz, gain);
            tapfrag = (uint32_t*) be.alloc(sizeof(F->frag));
            rc = tapfrag ? be.upload(tapfrag, F->frag, sizeof(F->frag)) : DH_ENOMEM;       // (the first 2 KS fragments are the ones a kernel reads)
            if (!rc) rc = be.sync();
            delete F;
            if (rc) return rc == DH_ENOMEM ? rc : DH_EDEVICE;
        }
#undef DH_ALLOC
        return reset();
    }

    void destroy() {
        void* ptrs[] = { dsp_state, syms, sym_count, sym_carry, dec_state, frames, frame_count, events, ev_count,
                         overflow, filtered, rrc_hist, staging, staging_counts, counts_copy, tables, custom_taps, tapfrag };
        for (void* p : ptrs) if (p) be.free(p);
    }

    int reset() {
        int rc = be.zero(overflow, sizeof(uint32_t) * 16);
        rc |= be.zero(sym_count, sizeof(uint32_t) * L.B);
        if (sym_carry) rc |= be.zero(sym_carry, (size_t) L.B * dh_carry_max(L.proto));
        if (dsp_state) rc |= be.zero(dsp_state, sizeof(uint32_t) * L.B * L.state_words);
        if (syms) rc |= be.zero(syms, L.B * L.sym_stride);
        if (dec_state) rc |= be.zero(dec_state, sizeof(uint32_t) * L.B * DH_DEC_STATE_WORDS);
        if (frame_count) rc |= be.zero(frame_count, sizeof(uint32_t) * L.B);
        if (ev_count) rc |= be.zero(ev_count, sizeof(uint32_t) * L.B);
        if (rrc_hist) rc |= be.zero(rrc_hist, sizeof(float) * L.B * L.nz);
        // the fused slicer starts with nz zero samples of history in its tail (RrcFilter's delay line)
        rc |= be.launch_init_state(dsp_state, L.state_words, L.fused ? L.nz : 0u, dec_state, slot_filter, L.B);
        last_n = 0;
        return rc ? DH_EDEVICE : DH_OK;
    }

    int set_slot_filter(uint32_t f) {
        if (L.proto != DH_PROTO_DMR) return DH_EINVAL;
        slot_filter = f;
        return be.launch_set_slot_filter(dec_state, f, L.B) ? DH_EDEVICE : DH_OK;
    }
    // one channel only (a module instance attached to a shared engine: include/digiham/shared_engine.hpp)
    int set_slot_filter_channel(uint32_t ch, uint32_t f) {
        if (L.proto != DH_PROTO_DMR || ch >= L.B) return DH_EINVAL;
        return be.launch_set_slot_filter(dec_state + (size_t) ch * DH_DEC_STATE_WORDS, f, 1) ? DH_EDEVICE : DH_OK;
    }
    // back to the freshly-constructed state of ONE channel (the others keep streaming)
    int reset_channel(uint32_t ch) {
        if (ch >= L.B) return DH_EINVAL;
        int rc = be.zero(sym_count + ch, sizeof(uint32_t));
        if (sym_carry) rc |= be.zero(sym_carry + (size_t) ch * dh_carry_max(L.proto), dh_carry_max(L.proto));
        if (dsp_state) rc |= be.zero(dsp_state + (size_t) ch * L.state_words, sizeof(uint32_t) * L.state_words);
        if (syms) rc |= be.zero(syms + (size_t) ch * L.sym_stride, L.sym_stride);
        if (dec_state) rc |= be.zero(dec_state + (size_t) ch * DH_DEC_STATE_WORDS, sizeof(uint32_t) * DH_DEC_STATE_WORDS);
        if (frame_count) rc |= be.zero(frame_count + ch, sizeof(uint32_t));
        if (ev_count) rc |= be.zero(ev_count + ch, sizeof(uint32_t));
        if (rrc_hist) rc |= be.zero(rrc_hist + (size_t) ch * L.nz, sizeof(float) * L.nz);
        rc |= be.launch_init_state(dsp_state ? dsp_state + (size_t) ch * L.state_words : nullptr, L.state_words, L.fused ? L.nz : 0u,
                                   dec_state ? dec_state + (size_t) ch * DH_DEC_STATE_WORDS : nullptr, slot_filter, 1);
        return rc ? DH_EDEVICE : DH_OK;
    }

    void fill_dec_params(const uint8_t* d_syms, size_t stride, const uint32_t* d_count) {
        dec.syms = d_syms; dec.sym_stride = stride; dec.sym_count = d_count;
        dec.carry = sym_carry; dec.carry_stride = dh_carry_max(L.proto);
        dec.state = dec_state; dec.state_stride = DH_DEC_STATE_WORDS;
        dec.out = frames; dec.out_stride = L.out_cap; dec.out_cap = L.out_cap; dec.out_count = frame_count;
        dec.events = events; dec.ev_stride = L.ev_cap; dec.ev_cap = L.ev_cap; dec.ev_count = ev_count;
        dec.overflow = overflow; dec.T = tables; dec.n_channels = L.B;
    }

    // d_counts (device, [B], or null): ragged push -- channel b brings d_counts[b] <= n samples of its row
    int push(const float* d_in, size_t stride, size_t n, const uint32_t* d_counts = nullptr) {
        if ((!d_in && n) || n > L.max_samples || stride < n) return DH_EINVAL;
        if (d_counts && L.rrc == DH_RRC_CUSTOM) return DH_EINVAL;          // (the generic FIR takes whole pushes only)
        if (!L.rrc && !L.demod) return DH_EINVAL;
        // what dh_engine_read_filtered needs to know later is kept in a buffer of the engine's own: the caller's counts only
        // have to live as long as the sample rows (until the push has run)
        last_counts = nullptr;
        if (d_counts && filtered) {
            if (!counts_copy) { counts_copy = (uint32_t*) be.alloc(sizeof(uint32_t) * L.B); if (!counts_copy) return DH_ENOMEM; }
            if (be.copy_device(counts_copy, d_counts, sizeof(uint32_t) * L.B)) return DH_EDEVICE;
            last_counts = counts_copy;
        }
        // an empty push is a module call with nothing readable: it runs (zero outputs, state untouched) and never
        // dereferences the sample pointer, which may then be null
        if (!d_in) d_in = reinterpret_cast<const float*>(overflow);
        last_n = (uint32_t) n;
        int rc = 0;
        const float* demod_in = d_in; size_t demod_stride = stride;
        const bool fast = (L.flags & DH_FLAG_FAST_FIR) != 0;                        // (with fused_keep: which floats the one launch delivers)
        bool decoder_done = false;
        be.timing_mark(0);
        if (L.rrc == DH_RRC_CUSTOM) {
            DhRrcGenParams G{};
            G.in = d_in; G.in_stride = stride; G.out = filtered; G.out_stride = L.max_samples; G.hist = rrc_hist; G.taps = custom_taps;
            G.n = (uint32_t) n; G.n_channels = L.B; G.nz = L.nz; G.gain = custom_gain;
            if (n) rc |= be.launch_rrc_generic(G);
            if (n) rc |= be.launch_rrc_hist(rrc_hist, d_in, stride, (uint32_t) n, nullptr, L.nz, L.B);
            demod_in = filtered; demod_stride = L.max_samples;
        } else if (L.rrc && !L.fused) {
            rrcp.in = d_in; rrcp.in_stride = stride; rrcp.out = filtered; rrcp.out_stride = L.max_samples;
            rrcp.hist = rrc_hist; rrcp.n = (uint32_t) n; rrcp.n_per = d_counts; rrcp.n_channels = L.B; rrcp.nz = L.nz; rrcp.fast = fast;
            fill_taps(L.rrc, rrcp.taps, &rrcp.gain); rrcp.rgain = 1.0 / rrcp.gain; rrcp.inv_gain = (float) rrcp.rgain;
            if (n) rc |= be.launch_rrc_tiles(rrcp, L.nz, fast);
            if (n) rc |= be.launch_rrc_hist(rrc_hist, d_in, stride, (uint32_t) n, d_counts, L.nz, L.B);
            demod_in = filtered; demod_stride = L.max_samples;
        }
        be.timing_mark(1);
        if (L.demod) {
            dsp.in = demod_in; dsp.in_stride = demod_stride; dsp.n = (uint32_t) n; dsp.n_per = d_counts;
            dsp.state = (float*) dsp_state; dsp.state_stride = L.state_words;
            dsp.syms = syms; dsp.sym_stride = L.sym_stride;
            dsp.sym_count = sym_count; dsp.sym_cap = L.sym_cap; dsp.overflow = overflow; dsp.n_channels = L.B;
            dsp.sps = L.sps; dsp.lo = L.lo; dsp.hi = L.hi;
            dsp.levels = L.demod; dsp.invert = (L.flags & DH_FLAG_FSK_INVERT) ? 1 : 0;
            dsp.nz = L.fused ? L.nz : 0; dsp.fast = fast && !L.fused_keep;       // (fused_keep = 2 is an error-bounded kernel: exact dibits)
            dsp.ordered_timing = (L.flags & DH_FLAG_ORDERED_TIMING) ? 1 : 0;
            dsp.exact_mode = (L.flags & DH_FLAG_EXACT_FIR) ? 2 : (L.flags & DH_FLAG_EXACT_SYMBOLS) ? 1 : 0;
            if (L.fused) {
                fill_taps(L.rrc, dsp.taps, &dsp.gain); dsp.rgain = 1.0 / dsp.gain; dsp.inv_gain = (float) dsp.rgain;
                dsp.err_coef = dh_fir_error_coefficient(dsp.taps, L.nz, dsp.gain);
                dsp.tapfrag = tapfrag; dsp.err_coef_f16 = err_coef_f16;
            }
            dsp.filt_out = L.fused_keep ? filtered : nullptr; dsp.filt_stride = L.max_samples;
            // slicer and decoder of a channel in one wavefront where the backend has that kernel (sps 10, wide or
            // no RRC); DH_FLAG_SPLIT_STAGES keeps the two launches (per-stage timing, A/B measurements)
            int chained = 1;
            if (L.proto && !(L.flags & DH_FLAG_SPLIT_STAGES) && !L.fused_keep) {
                fill_dec_params(syms, L.sym_stride, sym_count);
                chained = be.launch_chain(dsp, dec, dsp.nz, fast, L.proto);
                if (chained < 0) rc |= chained;
            }
            if (chained == 1) rc |= be.launch_rrc_demod(dsp, dsp.nz, fast);
            decoder_done = chained == 0;
        }
        be.timing_mark(2);
        if (L.proto && L.demod && !decoder_done) {
            fill_dec_params(syms, L.sym_stride, sym_count);
            rc |= be.launch_decoder(dec, L.proto);
        }
        be.timing_mark(3);
        be.timing_next();
        return rc ? DH_EDEVICE : DH_OK;
    }

    int push_host(const float* h_in, size_t stride, size_t n, const uint32_t* h_counts = nullptr) {
        if (!h_in || n > L.max_samples || stride < n) return DH_EINVAL;
        if (!staging) { staging = (float*) be.alloc(sizeof(float) * (size_t) L.B * L.max_samples); if (!staging) return DH_ENOMEM; }
        // one strided copy for the whole batch (rows of n floats, host pitch `stride`, device pitch max_samples)
        if (be.upload2d(staging, sizeof(float) * L.max_samples, h_in, sizeof(float) * stride, sizeof(float) * n, L.B)) return DH_EDEVICE;
        if (h_counts) {
            if (!staging_counts) { staging_counts = (uint32_t*) be.alloc(sizeof(uint32_t) * L.B); if (!staging_counts) return DH_ENOMEM; }
            for (uint32_t b = 0; b < L.B; b++) if (h_counts[b] > n) return DH_EINVAL;
            if (be.upload(staging_counts, h_counts, sizeof(uint32_t) * L.B)) return DH_EDEVICE;
        }
        return push(staging, L.max_samples, n, h_counts ? staging_counts : nullptr);
    }

    // decoder-only engines: the caller's symbol rows are read in place
    int push_symbols(const uint8_t* d_syms, size_t stride, const uint32_t* d_count) {
        if (!L.proto || L.demod || !d_syms || !d_count) return DH_EINVAL;
        last_counts = nullptr;
        fill_dec_params(d_syms, stride, d_count);
        return be.launch_decoder(dec, L.proto) ? DH_EDEVICE : DH_OK;
    }

    int check_overflow() {
        // [0] a capacity was hit.  ([2] counts hand-overs of split launches that did not come -- not an error: the fix-up
        // launch behind every split launch finished those rows; dh_engine_debug_header(e, 202, ...) reads it.)
        uint32_t flag = 0;
        if (be.download(&flag, overflow, sizeof(flag))) return DH_EDEVICE;
        return flag ? DH_ECAPACITY : DH_OK;
    }

    // per-channel header words of the slicer state: timing blocks evaluated / decided by the ordered chain
    int timing_stats(uint32_t* h_blocks, uint32_t* h_ordered) {
        if (!dsp_state) return DH_EINVAL;
        const size_t pitch = sizeof(uint32_t) * L.state_words;
        if (h_blocks && be.download2d(h_blocks, sizeof(uint32_t), dsp_state + DH_ST_BLOCKS, pitch, sizeof(uint32_t), L.B)) return DH_EDEVICE;
        if (h_ordered && be.download2d(h_ordered, sizeof(uint32_t), dsp_state + DH_ST_ORDERED, pitch, sizeof(uint32_t), L.B)) return DH_EDEVICE;
        return DH_OK;
    }

    int debug_header(uint32_t word, uint32_t* h_out) {
        if (word >= 200u) {                                  // 200 + w: word w of the engine's flag block, the same for every channel
            if (word - 200u >= 16u || !h_out) return DH_EINVAL;
            uint32_t v = 0;
            if (be.download(&v, overflow + (word - 200u), sizeof(v))) return DH_EDEVICE;
            for (uint32_t b = 0; b < L.B; b++) h_out[b] = v;
            return DH_OK;
        }
        if (word >= 100u) {                                  // 100 + w: word w of the decoder state
            if (!dec_state || word - 100u >= DH_DEC_STATE_WORDS || !h_out) return DH_EINVAL;
            return be.download2d(h_out, sizeof(uint32_t), dec_state + (word - 100u), sizeof(uint32_t) * DH_DEC_STATE_WORDS, sizeof(uint32_t), L.B) ? DH_EDEVICE : DH_OK;
        }
        if (!dsp_state || word >= DH_STATE_HDR || !h_out) return DH_EINVAL;
        return be.download2d(h_out, sizeof(uint32_t), dsp_state + word, sizeof(uint32_t) * L.state_words, sizeof(uint32_t), L.B) ? DH_EDEVICE : DH_OK;
    }

    int read_row(const void* base, size_t row_bytes, uint32_t channel, const uint32_t* counts, size_t elem, void* h_out, size_t* n) {
        if (channel >= L.B || !n || !base) return DH_EINVAL;
        uint32_t cnt = 0; const uint32_t off = 0;
        if (be.download(&cnt, counts + channel, sizeof(cnt))) return DH_EDEVICE;
        const size_t cap = *n;
        *n = cnt;
        if (cnt > cap) return DH_ECAPACITY;
        if (cnt && h_out && be.download(h_out, (const char*) base + row_bytes * channel + (size_t) off * elem, cnt * elem)) return DH_EDEVICE;
        return DH_OK;
    }
};

}  // namespace dh
