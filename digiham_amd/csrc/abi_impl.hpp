// abi_impl.hpp -- the extern "C" surface of include/digiham_amd.h over dh::Engine<DH_BACKEND>.
// Included once by engine.hip (DH_BACKEND = HIP backend -> libdigiham_amd.so) and once by the
// CPU test harness (DH_BACKEND = lane-loop backend -> tests/host_harness/libdh_hostemu.so, never
// shipped and never loaded by the digiham_amd package).
//
// Before inclusion the includer defines DH_BACKEND and these free functions:
//   int  dh_be_device_count();
//   const char* dh_be_last_error();
//   int  dh_be_alloc(int device, size_t bytes, void** out); int dh_be_free(void*);
//   int  dh_be_copy(void* dst, const void* src, size_t bytes, int to_host);
//   int  dh_be_fec_block(int code, void* words, uint8_t* ok, size_t n, void* stream);
//   int  dh_be_bptc(const uint8_t* in, uint8_t* out, uint8_t* ok, size_t n, void* stream);
//   int  dh_be_trellis(const uint8_t* in, size_t in_stride, int n_dibits, uint8_t* out, size_t out_stride, uint8_t* metric, size_t n, void* stream);
//   int  dh_be_crc16(const uint8_t* in, size_t stride, int count, uint16_t* out, size_t n, void* stream);
//   int  dh_be_whitening(const uint8_t* in, uint8_t* out, size_t stride, int n_bits, size_t n, void* stream);
//   int  dh_be_dvfilter(const int16_t* in, int16_t* out, float* state, size_t B, size_t stride, size_t n, void* stream);
#pragma once

#include "engine_impl.hpp"

static_assert(sizeof(dh_event) == 32, "dh_event layout");

struct dh_engine { dh::Engine<DH_BACKEND> impl; };

enum { DH_CODE_H74 = 0, DH_CODE_H139, DH_CODE_H1511, DH_CODE_H1611, DH_CODE_QR, DH_CODE_G208, DH_CODE_G2412, DH_CODE_BCH3121 };

extern "C" {

const char* dh_version(void) { return "digiham_amd 0.1.0 (gfx950)"; }
const char* dh_last_error(void) { return dh_be_last_error(); }
int dh_device_count(void) { return dh_be_device_count(); }

int dh_device_alloc(int device, size_t bytes, void** out) { return out ? dh_be_alloc(device, bytes, out) : DH_EINVAL; }
int dh_device_free(void* p) { return dh_be_free(p); }
int dh_copy_to_host(void* dst, const void* src, size_t bytes) { return (dst && src) || !bytes ? dh_be_copy(dst, src, bytes, 1) : DH_EINVAL; }
int dh_copy_to_device(void* dst, const void* src, size_t bytes) { return (dst && src) || !bytes ? dh_be_copy(dst, src, bytes, 0) : DH_EINVAL; }

int dh_hamming_7_4(uint8_t* w, uint8_t* ok, size_t n, void* s) { return dh_be_fec_block(DH_CODE_H74, w, ok, n, s); }
int dh_hamming_13_9(uint16_t* w, uint8_t* ok, size_t n, void* s) { return dh_be_fec_block(DH_CODE_H139, w, ok, n, s); }
int dh_hamming_15_11(uint16_t* w, uint8_t* ok, size_t n, void* s) { return dh_be_fec_block(DH_CODE_H1511, w, ok, n, s); }
int dh_hamming_16_11(uint16_t* w, uint8_t* ok, size_t n, void* s) { return dh_be_fec_block(DH_CODE_H1611, w, ok, n, s); }
int dh_quadratic_residue(uint16_t* w, uint8_t* ok, size_t n, void* s) { return dh_be_fec_block(DH_CODE_QR, w, ok, n, s); }
int dh_golay_20_8(uint32_t* w, uint8_t* ok, size_t n, void* s) { return dh_be_fec_block(DH_CODE_G208, w, ok, n, s); }
int dh_golay_24_12(uint32_t* w, uint8_t* ok, size_t n, void* s) { return dh_be_fec_block(DH_CODE_G2412, w, ok, n, s); }
int dh_bch_31_21(uint32_t* w, uint8_t* ok, size_t n, void* s) { return dh_be_fec_block(DH_CODE_BCH3121, w, ok, n, s); }
int dh_bptc_196_96(const uint8_t* in, uint8_t* out, uint8_t* ok, size_t n, void* s) {
    if ((!in || !out || !ok) && n) return DH_EINVAL;
    return dh_be_bptc(in, out, ok, n, s);
}
int dh_trellis(const uint8_t* in, size_t in_stride, int n_dibits, uint8_t* out, size_t out_stride, uint8_t* metric, size_t n, void* s) {
    if (n_dibits < 1 || n_dibits > 192 || in_stride < (size_t) (n_dibits + 3) / 4 || out_stride < (size_t) (n_dibits + 7) / 8) return DH_EINVAL;
    if ((!in || !out || !metric) && n) return DH_EINVAL;
    return dh_be_trellis(in, in_stride, n_dibits, out, out_stride, metric, n, s);
}
int dh_crc16(const uint8_t* in, size_t stride, int count, uint16_t* out, size_t n, void* s) {
    if (count < 0 || stride < (size_t) count || ((!in || !out) && n)) return DH_EINVAL;
    return dh_be_crc16(in, stride, count, out, n, s);
}
int dh_whitening(const uint8_t* in, uint8_t* out, size_t stride, int n_bits, size_t n, void* s) {
    if (n_bits < 0 || n_bits > 255 || stride < (size_t) (n_bits + 7) / 8 || ((!in || !out) && n)) return DH_EINVAL;
    return dh_be_whitening(in, out, stride, n_bits, n, s);
}
int dh_dvfilter_s16(const int16_t* in, int16_t* out, float* state, size_t B, size_t stride, size_t n, void* s) {
    if (!in || !out || !state || stride < n) return DH_EINVAL;
    return dh_be_dvfilter(in, out, state, B, stride, n, s);
}

int dh_debug_div_gain(const float* in, float* out, size_t n, int narrow, void* s) {
    if ((!in || !out) && n) return DH_EINVAL;
    return dh_be_div_gain(in, out, n, narrow, s);
}

int dh_frontend_s16(const int16_t* in, size_t in_stride, float* out, size_t out_stride, float* state, size_t B, size_t n, int mode, int dcblock, void* s) {
    if (mode != DH_FE_AUDIO_S16 && mode != DH_FE_IQ_S16) return DH_EINVAL;
    if (((!in || !out) && n) || !state || out_stride < n || in_stride < n * (mode == DH_FE_IQ_S16 ? 2u : 1u)) return DH_EINVAL;
    return dh_be_frontend(in, in_stride, out, out_stride, state, B, n, mode, dcblock, s);
}

int dh_debug_div_const(const float* in, float* out, size_t n, unsigned divisor, void* s) {
    if (((!in || !out) && n) || divisor == 0) return DH_EINVAL;
    return dh_be_div_const(in, out, n, divisor, s);
}

int dh_debug_mfma_f16(const uint16_t* a, const uint16_t* b, const float* c, float* d, size_t tiles, void* s) {
    if ((!a || !b || !c || !d) && tiles) return DH_EINVAL;
    return dh_be_mfma_f16(a, b, c, d, tiles, s);
}
int dh_debug_f16_split(const float* in, uint16_t* h1, uint16_t* h2, size_t n, float scale, void* s) {
    if ((!in || !h1 || !h2) && n) return DH_EINVAL;
    return dh_be_f16_split(in, h1, h2, n, scale, s);
}

int dh_debug_copy(const void* src, void* dst, size_t n_bytes, void* s) {
    if ((!src && n_bytes) || (n_bytes & 15u) || (((uintptr_t) src | (uintptr_t) dst) & 15u)) return DH_EINVAL;      // (dst == null: read only)
    return dh_be_copy_kernel(src, dst, n_bytes, s);
}

int dh_engine_create(const dh_engine_config* cfg, dh_engine** out) {
    if (!cfg || !out) return DH_EINVAL;
    *out = nullptr;
    dh_engine* e = new (std::nothrow) dh_engine;
    if (!e) return DH_ENOMEM;
    int rc = e->impl.be.open(cfg->device, cfg->stream);
    if (rc == DH_OK) { auto on_device = e->impl.be.scope(); (void) on_device; rc = e->impl.init(*cfg); if (rc != DH_OK) e->impl.destroy(); }
    if (rc != DH_OK) { delete e; return rc; }
    *out = e;
    return DH_OK;
}

void dh_engine_destroy(dh_engine* e) {
    if (!e) return;
    {
        auto on_device = e->impl.be.scope(); (void) on_device;
        e->impl.be.sync();
        e->impl.be.close();
        e->impl.destroy();
    }
    delete e;
}

// every entry below runs on the engine's device whatever the calling thread's current device is (HipBackend::Scope)
#define DH_ON_DEVICE(e) auto dh_on_device_ = (e)->impl.be.scope(); (void) dh_on_device_
int dh_engine_reset(dh_engine* e) { if (!e) return DH_EINVAL; DH_ON_DEVICE(e); return e->impl.reset(); }
int dh_engine_set_slot_filter(dh_engine* e, uint32_t f) { if (!e) return DH_EINVAL; DH_ON_DEVICE(e); return e->impl.set_slot_filter(f); }
int dh_engine_reset_channel(dh_engine* e, uint32_t ch) { if (!e) return DH_EINVAL; DH_ON_DEVICE(e); return e->impl.reset_channel(ch); }
int dh_engine_set_slot_filter_channel(dh_engine* e, uint32_t ch, uint32_t f) { if (!e) return DH_EINVAL; DH_ON_DEVICE(e); return e->impl.set_slot_filter_channel(ch, f); }
int dh_engine_push(dh_engine* e, const float* d, size_t stride, size_t n) { if (!e) return DH_EINVAL; DH_ON_DEVICE(e); return e->impl.push(d, stride, n); }
int dh_engine_push_host(dh_engine* e, const float* h, size_t stride, size_t n) { if (!e) return DH_EINVAL; DH_ON_DEVICE(e); return e->impl.push_host(h, stride, n); }
int dh_engine_push_ragged(dh_engine* e, const float* d, size_t stride, const uint32_t* d_counts, size_t max_n) {
    if (!e || !d_counts) return DH_EINVAL;
    DH_ON_DEVICE(e);
    return e->impl.push(d, stride, max_n, d_counts);
}
int dh_engine_push_host_ragged(dh_engine* e, const float* h, size_t stride, const uint32_t* h_counts, size_t max_n) {
    if (!e || !h_counts) return DH_EINVAL;
    DH_ON_DEVICE(e);
    return e->impl.push_host(h, stride, max_n, h_counts);
}
int dh_engine_push_symbols(dh_engine* e, const uint8_t* d, size_t stride, const uint32_t* cnt) { if (!e) return DH_EINVAL; DH_ON_DEVICE(e); return e->impl.push_symbols(d, stride, cnt); }

int dh_engine_filtered(dh_engine* e, const float** d, size_t* stride) {
    if (!e || !e->impl.filtered) return DH_EINVAL;
    if (d) *d = e->impl.filtered;
    if (stride) *stride = e->impl.L.max_samples;
    return DH_OK;
}
int dh_engine_symbols(dh_engine* e, const uint8_t** d, size_t* stride, const uint32_t** cnt) {
    if (!e || !e->impl.syms) return DH_EINVAL;
    if (d) *d = e->impl.syms;
    if (stride) *stride = e->impl.L.sym_stride;
    if (cnt) *cnt = e->impl.sym_count;
    return DH_OK;
}
int dh_engine_frames(dh_engine* e, const uint8_t** d, size_t* stride, const uint32_t** cnt) {
    if (!e || !e->impl.frames) return DH_EINVAL;
    if (d) *d = e->impl.frames;
    if (stride) *stride = e->impl.L.out_cap;
    if (cnt) *cnt = e->impl.frame_count;
    return DH_OK;
}
int dh_engine_events(dh_engine* e, const dh_event** d, size_t* stride, const uint32_t** cnt) {
    if (!e || !e->impl.events) return DH_EINVAL;
    if (d) *d = e->impl.events;
    if (stride) *stride = e->impl.L.ev_cap;
    if (cnt) *cnt = e->impl.ev_count;
    return DH_OK;
}
int dh_engine_debug_header(dh_engine* e, uint32_t word, uint32_t* h_out) {
    if (!e) return DH_EINVAL;
    DH_ON_DEVICE(e);
    return e->impl.debug_header(word, h_out);
}
int dh_engine_timing_stats(dh_engine* e, uint32_t* h_blocks, uint32_t* h_ordered) {
    if (!e) return DH_EINVAL;
    DH_ON_DEVICE(e);
    return e->impl.timing_stats(h_blocks, h_ordered);
}
int dh_engine_read_symbols(dh_engine* e, uint32_t ch, uint8_t* h, size_t* n) {
    if (!e || !e->impl.syms) return DH_EINVAL;
    DH_ON_DEVICE(e);
    return e->impl.read_row(e->impl.syms, e->impl.L.sym_stride, ch, e->impl.sym_count, 1, h, n);
}
int dh_engine_read_frames(dh_engine* e, uint32_t ch, uint8_t* h, size_t* n) {
    if (!e || !e->impl.frames) return DH_EINVAL;
    DH_ON_DEVICE(e);
    return e->impl.read_row(e->impl.frames, e->impl.L.out_cap, ch, e->impl.frame_count, 1, h, n);
}
int dh_engine_read_events(dh_engine* e, uint32_t ch, dh_event* h, size_t* n) {
    if (!e || !e->impl.events) return DH_EINVAL;
    DH_ON_DEVICE(e);
    return e->impl.read_row(e->impl.events, sizeof(dh_event) * e->impl.L.ev_cap, ch, e->impl.ev_count, sizeof(dh_event), h, n);
}
int dh_engine_read_filtered(dh_engine* e, uint32_t ch, float* h, size_t* n) {
    if (!e || !e->impl.filtered || !n || ch >= e->impl.L.B) return DH_EINVAL;
    DH_ON_DEVICE(e);
    size_t cnt = e->impl.last_n;
    const size_t cap = *n;
    if (e->impl.last_counts) {                   // a ragged push: this channel's own count
        uint32_t c = 0;
        if (e->impl.be.download(&c, e->impl.last_counts + ch, sizeof(c))) return DH_EDEVICE;
        if (c < cnt) cnt = c;
    }
    *n = cnt;
    if (cnt > cap) return DH_ECAPACITY;
    if (cnt && h && e->impl.be.download(h, e->impl.filtered + (size_t) ch * e->impl.L.max_samples, sizeof(float) * cnt)) return DH_EDEVICE;
    return DH_OK;
}
int dh_engine_timing_enable(dh_engine* e, uint32_t max_pushes) { if (!e) return DH_EINVAL; DH_ON_DEVICE(e); return e->impl.be.timing_enable(max_pushes); }
int dh_engine_timing_read_split(dh_engine* e, float* first_ms, uint32_t* first_channels, uint32_t* n) {
    if (!e || !n) return DH_EINVAL;
    DH_ON_DEVICE(e);
    return e->impl.be.timing_read_split(first_ms, first_channels, n);
}
int dh_engine_timing_read(dh_engine* e, float* rrc_ms, float* slicer_ms, float* decoder_ms, uint32_t* n) {
    if (!e || !n) return DH_EINVAL;
    DH_ON_DEVICE(e);
    return e->impl.be.timing_read(rrc_ms, slicer_ms, decoder_ms, n);
}
int dh_engine_sync(dh_engine* e) {
    if (!e) return DH_EINVAL;
    DH_ON_DEVICE(e);
    if (e->impl.be.sync()) return DH_EDEVICE;
    return e->impl.check_overflow();
}

}  // extern "C"
