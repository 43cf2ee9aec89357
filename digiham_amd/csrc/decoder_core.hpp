// decoder_core.hpp -- Digiham::Decoder shell + DMR / YSF frame state machines, one channel per
// wavefront.
//
// The reference walks byte-per-dibit buffers with scalar loops.  Here a frame's dibits are
// loaded once (one symbol per lane, three loads), turned into two 192-bit BITPLANES with
// wave-wide votes (bit1 plane H, bit0 plane L), and every field the protocol needs is then a
// shift/mask/bit-reverse/interleave of those planes: wave-uniform integer work that stays off
// the vector memory path.  Sync correlation is XOR + popcount of the planes against the sync
// pattern planes; while unsynchronised, each lane tests one candidate offset and a vote picks
// the first hit (what the reference does one symbol per call: dmr_phase.cpp:39-47).
// Lane-parallel parts: plane construction, sync search, BPTC column decode, payload packing,
// Viterbi add-compare-select (16 states on 16 lanes), event / output stores.
#pragma once

#include "dh_portable.hpp"
#include <stddef.h>
#include "fec_core.hpp"

#include "../../include/digiham_amd.h"     // dh_event, DH_EV_*

// one value per lane that outlives a lane loop: a register on the GPU, a [lane] array in the harness; DH_LV_READ
// fetches the value of lane k (wave-uniform k) -- v_readlane
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
#define DH_LANE_VALUE(type, name) type name
#define DH_LV(name, lane) name
#define DH_LV_READ(name, k) ((uint32_t) __builtin_amdgcn_readlane((int) (name), (int) (k)))
#define DH_LV_DOWN(name, lane, d) ((uint32_t) __shfl_down((int) (name), (unsigned) (d), DH_WAVE))      // the value of lane + d
#else
#define DH_LANE_VALUE(type, name) type name[DH_WAVE]
#define DH_LV(name, lane) name[lane]
#define DH_LV_READ(name, k) ((uint32_t) name[k])
#define DH_LV_DOWN(name, lane, d) ((uint32_t) name[((lane) + (d)) < DH_WAVE ? (lane) + (d) : (lane)])
#endif
// the same for a struct of per-lane values (registers on the GPU, an array of structs in the harness); DH_LS_READ fetches a
// field of lane k (wave-uniform k: v_readlane), DH_LS_WRITE sets it (v_writelane)
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
#define DH_LANE_STRUCT(type, name) type name
#define DH_LANE_STRUCT_REF(type, name) type& name
#define DH_LS(name, lane) name
#define DH_LS_READ(name, field, k) ((uint32_t) __builtin_amdgcn_readlane((int) (name).field, (int) (k)))
// (v_writelane takes value and lane from scalar registers, a VOP3 of gfx9 reads at most one: the lane select goes through m0,
// which belongs to the compiler -- saved and restored around the write, as in DhState::set)
static __device__ __forceinline__ uint32_t dh_writelane(uint32_t reg, uint32_t lane, uint32_t v) {
    const uint32_t sv = (uint32_t) __builtin_amdgcn_readfirstlane((int) v), si = (uint32_t) __builtin_amdgcn_readfirstlane((int) lane);
    uint32_t keep;
    asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(reg), "=&s"(keep) : "s"(sv), "s"(si));
    return reg;
}
#define DH_LS_WRITE(name, field, k, v) (name).field = dh_writelane((name).field, (uint32_t) (k), (uint32_t) (v))
#else
#define DH_LANE_STRUCT(type, name) type name[DH_WAVE]
#define DH_LANE_STRUCT_REF(type, name) type* name
#define DH_LS(name, lane) name[lane]
#define DH_LS_READ(name, field, k) ((uint32_t) (name)[k].field)
#define DH_LS_WRITE(name, field, k, v) (name)[k].field = (uint32_t) (v)
#endif

#define DH_SYM_CARRY_MAX 512          // symbols a decoder may leave unread between pushes (<= 480)
#define DH_DSTAR_CARRY_MAX 704        // D-Star: the header phase waits for more than 660 bits (dstar_phase.hpp:52)
DH_HD uint32_t dh_carry_max(int proto) { return proto == DH_PROTO_DSTAR ? DH_DSTAR_CARRY_MAX : DH_SYM_CARRY_MAX; }
#define DH_DEC_STATE_WORDS 64

enum {
    DS_PHASE = 0, DS_SYNC_COUNT = 1, DS_SLOT = 2, DS_SLOT_STABILITY = 3, DS_SYNC_TYPE0 = 4, DS_SYNC_TYPE1 = 5,
    DS_SLOT_SYNC0 = 6, DS_SLOT_SYNC1 = 7, DS_ACTIVE_SLOT = 8, DS_SLOT_FILTER = 9, DS_SUPERFRAME0 = 10,
    DS_SUPERFRAME1 = 11, DS_EMB_OFF0 = 12, DS_EMB_OFF1 = 13, DS_EMB_DATA0 = 14 /*4 words*/, DS_EMB_DATA1 = 18,
    DS_CONSUMED = 22, DS_CARRY = 23, DS_SLOT_FILTER_DECODER = 24, DS_HAS_FICH = 25, DS_FICH = 26, DS_EXPECT_SUB = 27,
    // NXDN (Nxdn::FramedPhase, nxdn_phase.hpp:31-40): LICH + 1 (0 = none yet), collected SACCH fragments (bit i),
    // their bytes 1..4 as big-endian words
    DS_NX_LICH = 2, DS_NX_HAVE = 3, DS_NX_SACCH0 = 4,
    // POCSAG (Pocsag::CodewordPhase, pocsag_phase.hpp:27-36, and its Message, message.hpp:11-23): codeword counter,
    // message present, address, type, bit / char position, 80 content bytes (little-endian words)
    DS_PC_COUNTER = 2, DS_PC_HAS = 3, DS_PC_ADDR = 4, DS_PC_TYPE = 5, DS_PC_POS = 6, DS_PC_CONTENT = 32,
    // D-Star (DStar::VoicePhase, dstar_phase.hpp:67-77): frameCount, collected_data (2 x 3 bytes), messageBlocks,
    // headerCount, message (20 bytes, little-endian words), header (41 bytes)
    DS_DT_FRAME = 2, DS_DT_COLLECT0 = 3, DS_DT_COLLECT1 = 4, DS_DT_BLOCKS = 5, DS_DT_HCOUNT = 6, DS_DT_MESSAGE = 8,
    DS_DT_HEADER = 32
};

struct DhDecParams {
    const uint8_t* syms; size_t sym_stride;    // [B][sym_stride] this push's symbols (dibits, one per byte)
    const uint32_t* sym_count;                 // [B] symbols in this push
    uint8_t* carry; size_t carry_stride;       // [B][DH_SYM_CARRY_MAX] symbols left unread by the previous push
    uint32_t* state; size_t state_stride;      // [B][DH_DEC_STATE_WORDS]
    uint8_t* out; size_t out_stride; uint32_t out_cap; uint32_t* out_count;
    dh_event* events; size_t ev_stride; uint32_t ev_cap; uint32_t* ev_count;   // events may be null
    uint32_t* overflow;
    const DhFecTables* T;
    uint32_t n_channels;
};

#define DH_PLANE_WORDS 8            // 512 symbols: a YSF frame is 480, a DMR burst + search window 154
struct DhPlanes { uint64_t h[DH_PLANE_WORDS], l[DH_PLANE_WORDS]; };

// bits [start, start+cnt) of a plane, symbol `start` in bit 0 (cnt <= 32)
DH_HD uint32_t dh_plane_range(const uint64_t* w, int start, int cnt) {
    const int i = start >> 6, sh = start & 63;
    uint64_t v = w[i] >> sh;
    if (sh && i < DH_PLANE_WORDS - 1) v |= w[i + 1] << (64 - sh);
    return (uint32_t) (v & ((cnt >= 32) ? 0xFFFFFFFFull : ((1ull << cnt) - 1)));
}

DH_HD uint32_t dh_spread16(uint32_t x) {      // abcd -> 0a0b0c0d
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

DH_HD uint32_t dh_sym_at(const DhPlanes& p, int j) {
    return (uint32_t) ((((p.h[j >> 6] >> (j & 63)) & 1ull) << 1) | ((p.l[j >> 6] >> (j & 63)) & 1ull));
}

// sync words as planes (symbol i in bit i), derived from the ETSI TS 102 361-1 table 9.2 / YSF spec hex
// words (the dibit arrays at dmr_phase.hpp:25-28 and ysf_phase.hpp:21 are the same words)
constexpr uint32_t dh_sync_plane(uint64_t word, int ndibits, int bit) {
    uint32_t r = 0;
    for (int i = 0; i < ndibits; i++) {
        const uint32_t dibit = (uint32_t) ((word >> (2 * (ndibits - 1 - i))) & 3u);
        r |= ((dibit >> bit) & 1u) << i;
    }
    return r;
}
#define DH_DMR_BS_DATA_H  dh_sync_plane(0xDFF57D75DF5Dull, 24, 1)
#define DH_DMR_BS_VOICE_H dh_sync_plane(0x755FD7DF75F7ull, 24, 1)
#define DH_DMR_MS_DATA_H  dh_sync_plane(0xD5D7F77FD757ull, 24, 1)
#define DH_DMR_MS_VOICE_H dh_sync_plane(0x7F7D5DD57DFDull, 24, 1)
#define DH_DMR_SYNC_L     dh_sync_plane(0xDFF57D75DF5Dull, 24, 0)     /* every DMR sync dibit is 01 or 11: all ones */
#define DH_YSF_SYNC_H     dh_sync_plane(0xD471C9634Dull, 20, 1)
#define DH_YSF_SYNC_L     dh_sync_plane(0xD471C9634Dull, 20, 0)

#define DH_SYNCTYPE_DATA 1
#define DH_SYNCTYPE_VOICE 2

// dmr_phase.cpp:18-33
DH_HD int dh_dmr_sync_type(const DhPlanes& p, int start) {
    const uint32_t h = dh_plane_range(p.h, start, 24), l = dh_plane_range(p.l, start, 24);
    constexpr uint32_t SL = DH_DMR_SYNC_L, BD = DH_DMR_BS_DATA_H, BV = DH_DMR_BS_VOICE_H,
                       MD = DH_DMR_MS_DATA_H, MV = DH_DMR_MS_VOICE_H;
    static_assert(SL == 0xFFFFFFu, "DMR sync L plane");
    const int dl = dh_popc32(l ^ SL);
    if (dh_popc32(h ^ BD) + dl <= 3) return DH_SYNCTYPE_DATA;
    if (dh_popc32(h ^ BV) + dl <= 3) return DH_SYNCTYPE_VOICE;
    if (dh_popc32(h ^ MD) + dl <= 3) return DH_SYNCTYPE_DATA;
    if (dh_popc32(h ^ MV) + dl <= 3) return DH_SYNCTYPE_VOICE;
    return -1;
}

// ---------------------------------------------------------------------------------------------
// The 32 state words of a channel while its wavefront runs: ONE vector register, word i in lane i.  The frame
// state machines index them with run-time (wave-uniform) indices such as DS_SYNC_TYPE0 + slot; v_readlane /
// v_writelane take the index from a scalar register, so a read or write is a single instruction with no memory
// round trip (an LDS copy costs ~100 cycles of latency per access on this serial path, a register array would
// go to scratch).  s[i] reads, s[i] = v writes; the harness build is a plain array.
struct DhState {
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    uint32_t reg;
    __device__ __forceinline__ uint32_t get(uint32_t i) const { return (uint32_t) __builtin_amdgcn_readlane((int) reg, (int) i); }
    __device__ __forceinline__ void set(uint32_t i, uint32_t v) {
        const uint32_t sv = dh_uniform(v), si = dh_uniform(i);         // v_writelane takes value and lane from scalar registers
        // gfx9 VOP3 reads at most one SGPR, so the lane select goes through m0; m0 is reserved for the compiler
        // (a clobber would be ignored), hence saved and restored around the write
        uint32_t keep;
        asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
            : "+v"(reg), "=&s"(keep) : "s"(sv), "s"(si));
    }
    __device__ __forceinline__ void load(const uint32_t* g) { reg = threadIdx.x < DH_DEC_STATE_WORDS ? g[threadIdx.x] : 0u; }
    __device__ __forceinline__ void store(uint32_t* g) const { if (threadIdx.x < DH_DEC_STATE_WORDS) g[threadIdx.x] = reg; }
#else
    uint32_t w[DH_DEC_STATE_WORDS];
    uint32_t get(uint32_t i) const { return w[i]; }
    void set(uint32_t i, uint32_t v) { w[i] = v; }
    void load(const uint32_t* g) { for (int i = 0; i < DH_DEC_STATE_WORDS; i++) w[i] = g[i]; }
    void store(uint32_t* g) const { for (int i = 0; i < DH_DEC_STATE_WORDS; i++) g[i] = w[i]; }
#endif
    struct Ref {
        DhState* st; uint32_t i;
        DH_HD operator uint32_t() const { return st->get(i); }
        DH_HD Ref& operator=(uint32_t v) { st->set(i, v); return *this; }
        DH_HD Ref& operator=(const Ref& o) { st->set(i, (uint32_t) o); return *this; }
        DH_HD uint32_t operator++(int) { const uint32_t v = st->get(i); st->set(i, v + 1u); return v; }
    };
    DH_HD Ref operator[](uint32_t i) { return Ref{this, i}; }
    DH_HD Ref operator[](int i) { return Ref{this, (uint32_t) i}; }
};

struct DhDecCtx {
    const DhDecParams* P;
    const DhFecTables* T;              // LDS-resident prefix of the tables
    DhState* st;
    uint8_t* out; dh_event* ev;
    uint32_t nout, nev, consumed;
    bool overflow;
    bool writer;                       // this thread performs the (wave-uniform) global stores
};

DH_HD void dh_emit(DhDecCtx& c, uint8_t type, uint8_t a, uint8_t b, const uint8_t* payload, int len) {
    if (c.ev == nullptr) return;
    if (c.nev >= c.P->ev_cap) { c.overflow = true; return; }
    if (c.writer) {
        dh_event e;
        e.sym_index = c.consumed; e.type = type; e.a = a; e.b = b; e.len = (uint8_t) len;
        for (int i = 0; i < 24; i++) e.payload[i] = i < len ? payload[i] : (uint8_t) 0;
        c.ev[c.nev] = e;
    }
    c.nev++;
}

// the same with the payload (up to 12 bytes, memory order) in three words: eight word stores, no byte shuffling (the frame loops' hot events)
DH_HD void dh_emit_w(DhDecCtx& c, uint32_t type, uint32_t a, uint32_t b, uint32_t len, uint32_t w0 = 0, uint32_t w1 = 0, uint32_t w2 = 0) {
    if (c.ev == nullptr) return;
    if (c.nev >= c.P->ev_cap) { c.overflow = true; return; }
    if (c.writer) {
        uint32_t* q = reinterpret_cast<uint32_t*>(c.ev + c.nev);
        q[0] = c.consumed; q[1] = type | a << 8 | b << 16 | len << 24;
        q[2] = w0; q[3] = w1; q[4] = w2; q[5] = 0u; q[6] = 0u; q[7] = 0u;
    }
    c.nev++;
}

// FramePhase::FramePhase() + Decoder::setPhase (dmr_phase.cpp:49-52, dmr_phase.hpp:51-60, dmr_decoder.cpp:17-23)
DH_HD void dh_dmr_enter_frame_phase(DhState& s) {
    s[DS_SYNC_COUNT] = 0; s[DS_SLOT] = (uint32_t) -1; s[DS_SLOT_STABILITY] = 0;
    s[DS_SYNC_TYPE0] = s[DS_SYNC_TYPE1] = (uint32_t) -1;
    s[DS_SLOT_SYNC0] = s[DS_SLOT_SYNC1] = 0;
    s[DS_ACTIVE_SLOT] = (uint32_t) -1;
    s[DS_SUPERFRAME0] = s[DS_SUPERFRAME1] = 0;
    s[DS_EMB_OFF0] = s[DS_EMB_OFF1] = 0;
    for (int i = 0; i < 8; i++) s[DS_EMB_DATA0 + i] = 0;
    s[DS_SLOT_FILTER] = s[DS_SLOT_FILTER_DECODER];
}

// ------------------------------------------------------------------------------------------
#define DH_SYMWIN 1024               // fresh symbols staged in LDS per refill (a DMR burst is 144, a YSF frame 480)
// Frame-parallel DMR (dh_dmr_channel): the bit planes of a chunk of up to 64 bursts (dibit j of the chunk: bit j & 31 of
// word j >> 5; plane_h bit 1, plane_l bit 0 of the dibit), and a block that first holds the collected embedded-signalling words of
// the bursts that close an embedded LC (pass B -> pass C), then the voice payloads on their way out (pass C).
#define DH_DMR_CHUNK 64
#define DH_DMR_PLANE_GROUPS (DH_DMR_CHUNK * 9)                  /* 16-dibit groups: 64 bursts x 144 dibits */
#define DH_DMR_PLANE_WORDS (DH_DMR_PLANE_GROUPS / 2 + 4)        /* + slack: a lane reads six consecutive words */
struct DhDmrChunkShared {
    uint32_t plane_h[DH_DMR_PLANE_WORDS], plane_l[DH_DMR_PLANE_WORDS];
    union { uint32_t emb_words[DH_DMR_CHUNK][4]; uint32_t voice[DH_DMR_CHUNK][7]; };
};

// YSF (dh_ysf_channel): the decoded FICH and V/D2 DCH codewords of the next frames (13 bytes each, in 4 words), their V/D2 voice blocks
// (decoded as if every frame were V/D mode 2: the frame loop copies them out when it is), and a block that holds
// the bit planes of those frames while the codewords are picked out of them, then the dibit streams of the codewords that need
// the full Viterbi decoder.
#define DH_YSF_CHUNK 16
#define DH_YSF_PLANE_WORDS (DH_YSF_CHUNK * 15 + 4)              /* 480 dibits a frame; + slack: a lane reads consecutive words */
struct DhYsfChunkShared {
    uint32_t res[DH_YSF_CHUNK][2][4];
    uint32_t voice[DH_YSF_CHUNK][10];                           // the five V/D2 voice blocks of a frame, 7 bytes in two words each (dh_ysf_v2_block)
    union {
        struct { uint32_t plane_h[DH_YSF_PLANE_WORDS], plane_l[DH_YSF_PLANE_WORDS]; };
        uint32_t dirty[2 * DH_YSF_CHUNK][8];                    // bit 1 / bit 0 streams (100 bits each) of a codeword
    };
};

struct DhDecShared {
    uint8_t  carry[DH_SYM_CARRY_MAX];     // symbols carried from the previous push
    uint8_t  symwin[DH_SYMWIN];           // window of this push's symbols (refilled with 16-byte-per-lane loads)
    // codes + small syndrome LUTs (everything of DhFecTables in front of the two 16 KiB Golay LUTs): table lookups
    // on the frame path are LDS reads instead of dependent global loads
    uint32_t fec_small[(offsetof(DhFecTables, lut_g208) + 3) / 4];
    DhPlanes planes;                      // bit planes of the current frame (wave-uniform; LDS broadcast reads)
    uint32_t colword[16];
    union {
        struct {
            uint32_t vit_metric[2][64];
            uint64_t vit_dec[192];
            uint32_t vit_in[4][48];           // up to 4 concurrent codewords of 192 dibits, one dibit per byte (dh_vit_word)
            uint8_t  vit_out[4][24];
            uint8_t  vit_best_metric[4];
            DhYsfChunkShared ysf;
        };
        DhDmrChunkShared dmr;                 // the frame-parallel DMR decoder's chunk (no Viterbi in DMR)
    };
};

// virtual symbol stream of a channel for this push: carried symbols, then the fresh ones
// (both parts are read through LDS: `carry` is S.carry, fresh symbols come through the S.symwin window)
struct DhSymView { const uint8_t* carry; uint32_t nc; const uint8_t* fresh; uint32_t nfresh; uint8_t* win; uint32_t wbase, wlen; };
DH_HD uint32_t dh_view_at(const DhSymView& v, uint32_t j) { return j < v.nc ? v.carry[j] : v.win[(j - v.nc) - v.wbase]; }

// make the symbols [pos, pos+need) of the virtual stream readable (wave-uniform; refills the LDS window when the
// range runs past it).  need <= DH_SYMWIN - 4.
DH_HD void dh_view_ensure(DhSymView& v, uint32_t pos, uint32_t need) {
    const uint32_t end = pos + need;
    if (end <= v.nc) return;                                   // entirely inside the carried part
    const uint32_t f0 = pos > v.nc ? pos - v.nc : 0u;
    uint32_t f1 = end - v.nc; if (f1 > v.nfresh) f1 = v.nfresh;
    if (f0 >= v.wbase && f1 <= v.wbase + v.wlen) return;
    const uint32_t mis = (uint32_t) ((uintptr_t) (v.fresh + f0) & 3u);
    const uint32_t wb = f0 >= mis ? f0 - mis : f0;             // start the window on a 4-byte aligned address when possible
    DH_BARRIER();                                              // earlier readers of the window are done
    // four 32-bit words per lane.  The loads are issued unconditionally (a lane past the end of the push reads the
    // window's first word instead) so that all four are in flight together; a load under a branch would be waited
    // for before the next one starts.  Words that straddle the end, or an unaligned row, take the byte loop.
    const bool aligned = (((uintptr_t) (v.fresh + wb)) & 3u) == 0 && wb + 4u <= v.nfresh;
    DH_FOR_LANES(lane) {
        constexpr int NW = DH_SYMWIN / (4 * DH_WAVE);
        uint32_t word[NW];
        for (int k = 0; k < NW; k++) {
            const uint32_t f = wb + 4u * (uint32_t) lane + (uint32_t) k * 4u * DH_WAVE;
            const bool whole = aligned && f + 4u <= v.nfresh;
            word[k] = aligned ? *reinterpret_cast<const uint32_t*>(v.fresh + (whole ? f : wb)) : 0u;
        }
        for (int k = 0; k < NW; k++) {
            const uint32_t idx = 4u * (uint32_t) lane + (uint32_t) k * 4u * DH_WAVE, f = wb + idx;
            uint32_t w = word[k];
            if (!(aligned && f + 4u <= v.nfresh)) {
                w = 0;
                for (uint32_t b = 0; b < 4u; b++) if (f + b < v.nfresh) w |= (uint32_t) v.fresh[f + b] << (8u * b);
            }
            *reinterpret_cast<uint32_t*>(v.win + idx) = w;
        }
    }
    DH_BARRIER();
    v.wbase = wb; v.wlen = DH_SYMWIN;
}


// `cnt` (<= 16) symbols given as two bit masks (symbol i in bit i), packed MSB-first 2 bits per symbol -- what the
// reference's `(v << 2) | raw[i]` loops build (e.g. dmr_phase.cpp:123-132, :236-245)
DH_HD uint32_t dh_pack_msb(uint32_t h, uint32_t l, int cnt) {
    return (dh_spread16(dh_brev32(h) >> (32 - cnt)) << 1) | dh_spread16(dh_brev32(l) >> (32 - cnt));
}

// dmr_phase.cpp:18-33 on the sync slot given as bit masks (dibit 66+i in bit i)
DH_HD int dh_dmr_sync_type_bits(uint32_t h, uint32_t l) {
    constexpr uint32_t SL = DH_DMR_SYNC_L, BD = DH_DMR_BS_DATA_H, BV = DH_DMR_BS_VOICE_H,
                       MD = DH_DMR_MS_DATA_H, MV = DH_DMR_MS_VOICE_H;
    const int dl = dh_popc32(l ^ SL);
    if (dh_popc32(h ^ BD) + dl <= 3) return DH_SYNCTYPE_DATA;
    if (dh_popc32(h ^ BV) + dl <= 3) return DH_SYNCTYPE_VOICE;
    if (dh_popc32(h ^ MD) + dl <= 3) return DH_SYNCTYPE_DATA;
    if (dh_popc32(h ^ MV) + dl <= 3) return DH_SYNCTYPE_VOICE;
    return -1;
}

DH_HD void dh_load_planes(const DhSymView& syms, uint32_t pos, uint32_t total, DhPlanes& pl, int nwords) {
    for (int w = 0; w < DH_PLANE_WORDS; w++) { pl.h[w] = 0; pl.l[w] = 0; }
    for (int w = 0; w < nwords; w++) {
        uint64_t mh = 0, ml = 0;
        DH_FOR_LANES(lane) {
            const uint32_t j = pos + (uint32_t) (w * 64 + lane);
            const uint32_t v = j < total ? dh_view_at(syms, j) : 0u;
            DH_BALLOT_ACC(mh, (v >> 1) & 1u, lane);
            DH_BALLOT_ACC(ml, v & 1u, lane);
        }
        pl.h[w] = mh; pl.l[w] = ml;
    }
}

// kernel prologue: carried symbols and the small FEC tables into LDS
DH_HD void dh_stage_decoder_lds(const DhDecParams& P, DhDecShared& S, const uint8_t* carry_buf, uint32_t nc) {
    const uint32_t* tsrc = reinterpret_cast<const uint32_t*>(P.T);
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < nc; j += DH_WAVE) S.carry[j] = carry_buf[j];
        for (uint32_t j = lane; j < sizeof(S.fec_small) / 4; j += DH_WAVE) S.fec_small[j] = tsrc[j];
    }
    DH_BARRIER();
}
// the LDS copy viewed as a DhFecTables: valid for the codes and the small LUTs only (not lut_g208 / lut_g2412)
DH_HD const DhFecTables& dh_lds_tables(const DhDecShared& S) { return *reinterpret_cast<const DhFecTables*>(S.fec_small); }

// =============================================================================================
// Frame-parallel DMR decoder (round 5).
//
// The reference handles one 144-dibit burst per call (dmr_phase.cpp:61-302).  While a channel is in its FramePhase the burst
// grid of a push is known up front (pos + 144 k), and almost everything a burst needs is a function of its own dibits alone:
// sync correlation, TACT Hamming(7,4), EMB QR(16,7), slot-type Golay(20,8), BPTC(196,96), the 27 voice bytes.  Only the slot /
// superframe / embedded-LC bookkeeping (:65-204) chains the bursts together, and it needs a dozen bits of each.  So a push is
// taken in CHUNKS of up to 64 bursts:
//   planes  the chunk's dibits -> two bit planes in LDS (16 dibits per lane and load, packed in registers)
//   pass A  one burst per LANE: its 144 dibits as 2 x 5 words from the planes, every block code of the burst decoded lane-locally
//           (bit-sliced Hamming(13,9) over the BPTC columns), results condensed into a summary word
//   pass B  the reference's state machine, burst after burst, on the summaries: scalar code only (v_readlane in, v_writelane
//           out), leaves one word of event / output flags per burst; a burst that sends the decoder back to its SyncPhase
//           ends the chunk
//   pass C  one burst per lane again: embedded LCs, event records (offsets by a vote-based prefix sum), voice payloads
//           (staged in LDS, stored coalesced)
// The SyncPhase search (dmr_phase.cpp:35-47) is unchanged.
// =============================================================================================
#if defined(DH_ASM_MARKERS) && DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
#define DH_DMARK(text) asm volatile("; DH_DMARK " text ::: "memory")      // (tools/asm_census.py: the decoder's regions in the assembly)
#else
#define DH_DMARK(text) ((void) 0)
#endif
struct DhDmrLane {
    uint32_t hw[5], lw[5];        // the burst's dibits: dibit i = bit i & 31 of word i >> 5 (hw: bit 1, lw: bit 0)
    uint32_t summary;             // pass A -> B, C (DH_DS_*)
    uint32_t frag;                // the 32 embedded-signalling bits (dmr_phase.cpp:141-145)
    uint32_t bptc[3];             // BPTC(196,96) payload, 12 bytes in memory order
    uint32_t flags;               // pass B -> C (DH_DF_*)
};
enum {   // summary word
    DH_DS_HAS_TACT = 1u << 0, DH_DS_TACT_SLOT_SHIFT = 1, DH_DS_SYNC_SHIFT = 2 /*2 bits: 0 none, DATA, VOICE*/, DH_DS_EMB_OK = 1u << 4,
    DH_DS_LCSS_SHIFT = 5 /*2*/, DH_DS_ST_OK = 1u << 7, DH_DS_DT_SHIFT = 8 /*4*/, DH_DS_BPTC_OK = 1u << 12,
    DH_DS_EMB_CC_SHIFT = 16 /*4*/, DH_DS_ST_CC_SHIFT = 20 /*4*/,
    DH_DS_DFLAGS_SHIFT = 24 /*4: what a data burst emits, DH_DF_SLOTTYPE .. DH_DF_BPTC_TAIL >> 6*/
};
enum {   // flag word, events in the order they are emitted
    DH_DF_RESET_OTHER = 1u << 0, DH_DF_SYNC = 1u << 1, DH_DF_EMB = 1u << 2, DH_DF_EMB_LC = 1u << 3, DH_DF_SLOT_RESET = 1u << 4,
    DH_DF_META_RESET = 1u << 5, DH_DF_SLOTTYPE = 1u << 6, DH_DF_SLOT_RESET2 = 1u << 7, DH_DF_BPTC = 1u << 8, DH_DF_BPTC_TAIL = 1u << 9,
    DH_DF_EVENTS = 0x3FFu,
    DH_DF_VOICE = 1u << 10, DH_DF_SLOT = 1u << 11, DH_DF_SOFT = 1u << 12
};

struct alignas(16) DhU4 { uint32_t x, y, z, w; };

// 16 dibits, one per byte in four words -> their bit-1 mask | bit-0 mask << 16
DH_HD uint32_t dh_pack16_dibits(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    const uint32_t t = w0 | (w1 << 2) | (w2 << 4) | (w3 << 6);      // byte q: dibits q, 4 + q, 8 + q, 12 + q
    uint32_t r[2];
    for (int b = 0; b < 2; b++) {
        uint32_t x = (t >> b) & 0x55555555u;                       // bit 8 q + 2 g = dibit 4 g + q
        x = (x | (x >> 7)) & 0x00FF00FFu;                          // byte 0: q = 0, 1 interleaved; byte 2: q = 2, 3
        x = (x | (x << 4)) & 0x0F0F0F0Fu;
        x = (x | (x << 2)) & 0x33333333u;                          // pair (q, q + 1) of group g at bit 4 g (+ 16)
        r[b] = ((x & 0xFFFFu) | (x >> 14)) & 0xFFFFu;
    }
    return r[1] | (r[0] << 16);
}

// dibits [origin, origin + 16 ngroups) of the virtual stream -> the two bit planes of a chunk, 16 dibits per lane and step.  `origin`
// (it may lie in front of the stream) is chosen by the caller so that every group inside the fresh part is a 16-byte aligned piece
// of the row; groups that touch the carried part or the end of the stream are gathered dibit by dibit.
DH_HD void dh_build_chunk_planes(const DhSymView& syms, int32_t origin, uint32_t ngroups, uint32_t total, uint32_t* plane_h, uint32_t* plane_l) {
    // a 16-byte piece of the row that may be loaded whatever a lane's own group is (loads are issued unconditionally: a load under a
    // branch is waited for on the spot)
    const int64_t first_fresh = (int64_t) syms.nc;
    int64_t safe = origin;
    while (safe < first_fresh) safe += 16;
    const bool have_safe = safe + 16 <= (int64_t) total;
    uint16_t* const ph = reinterpret_cast<uint16_t*>(plane_h);
    uint16_t* const pl = reinterpret_cast<uint16_t*>(plane_l);
    constexpr int BATCH = 5;
    for (uint32_t g0 = 0; g0 < ngroups; g0 += BATCH * DH_WAVE) {
        DH_FOR_LANES(lane) {
            uint32_t w[BATCH][4];
            bool fast[BATCH];
#pragma unroll
            for (int b = 0; b < BATCH; b++) {
                const uint32_t g = g0 + (uint32_t) b * DH_WAVE + (uint32_t) lane;
                const int64_t j0 = (int64_t) origin + 16 * (int64_t) g;
                fast[b] = have_safe && g < ngroups && j0 >= first_fresh && j0 + 16 <= (int64_t) total;
                if (have_safe) {
                    const DhU4 v = *reinterpret_cast<const DhU4*>(syms.fresh + ((fast[b] ? j0 : safe) - first_fresh));
                    w[b][0] = v.x; w[b][1] = v.y; w[b][2] = v.z; w[b][3] = v.w;
                } else { w[b][0] = w[b][1] = w[b][2] = w[b][3] = 0u; }
            }
#pragma unroll
            for (int b = 0; b < BATCH; b++) {
                const uint32_t g = g0 + (uint32_t) b * DH_WAVE + (uint32_t) lane;
                if (g < ngroups) {
                    if (DH_UNLIKELY(!fast[b])) {
                        const int64_t j0 = (int64_t) origin + 16 * (int64_t) g;
                        for (int q = 0; q < 4; q++) {
                            uint32_t v = 0;
                            for (int e = 0; e < 4; e++) {
                                const int64_t j = j0 + 4 * q + e;
                                uint32_t d = 0;
                                if (j >= 0 && j < (int64_t) total) d = j < first_fresh ? (uint32_t) syms.carry[j] : (uint32_t) syms.fresh[j - first_fresh];
                                v |= (d & 3u) << (8 * e);
                            }
                            w[b][q] = v;
                        }
                    }
                    const uint32_t hl = dh_pack16_dibits(w[b][0] & 0x03030303u, w[b][1] & 0x03030303u, w[b][2] & 0x03030303u, w[b][3] & 0x03030303u);
                    ph[g] = (uint16_t) hl; pl[g] = (uint16_t) (hl >> 16);
                }
            }
        }
    }
    DH_BARRIER();
}

// CNT (<= 32) bits from bit START of a little-endian bit string in words
template <int START, int CNT> DH_HD uint32_t dh_fld(const uint32_t* w) {
    constexpr int i = START >> 5, sh = START & 31;
    uint32_t v = w[i] >> sh;
    if (sh + CNT > 32) v |= w[i + 1] << (32 - sh);
    return CNT >= 32 ? v : v & ((1u << (CNT & 31)) - 1u);
}

// BPTC(196,96) (bptc_196_96.c:5-59), one block per lane.  get(r) = received bit r (0..195).  The de-interleave and the 13 x 15
// pivot are one gather into ROW words (bit 14 - c of R[k] = column c, row k); the fifteen Hamming(13,9) column decodes run
// bit-sliced over the row words: syndrome bit j of every column at once is the XOR of the rows in parity check j, a column's
// error sits in the row whose check pattern equals its syndrome (no two rows share one; a non-zero syndrome that matches no row
// is the reference's "syndrome not found": block rejected, hamming_13_9.c:70-84).  Rows: Hamming(15,11), every syndrome
// correctable.  Output: the 96 payload bits as 12 bytes in memory order (3 words).
#define DH_H139_SYN(k) ((k) == 0 ? 0xFu : (k) == 1 ? 0xEu : (k) == 2 ? 0x7u : (k) == 3 ? 0xAu : (k) == 4 ? 0x5u : (k) == 5 ? 0xBu : \
                        (k) == 6 ? 0xCu : (k) == 7 ? 0x6u : (k) == 8 ? 0x3u : (1u << (12 - (k))))      /* parity part of the generator rows (hamming_13_9.c:5-13), then the identity */
template <typename Get>
DH_HD bool dh_dmr_bptc_lane(const DhFecTables& T, Get get, uint32_t* out3) {
    uint32_t R[13];
#pragma unroll
    for (int k = 0; k < 13; k++) {
        uint32_t w = 0;
#pragma unroll
        for (int c = 0; c < 15; c++) w |= get(((k * 15 + c + 1) * 181) % 196) << (14 - c);
        R[k] = w;
    }
    uint32_t Sy[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < 13; k++) if ((DH_H139_SYN(k) >> j) & 1u) x ^= R[k];
        Sy[j] = x;
    }
    const uint32_t nz = Sy[0] | Sy[1] | Sy[2] | Sy[3];
    uint32_t matched = 0;
#pragma unroll
    for (int k = 0; k < 13; k++) {
        uint32_t f = nz;
#pragma unroll
        for (int j = 0; j < 4; j++) f &= ((DH_H139_SYN(k) >> j) & 1u) ? Sy[j] : ~Sy[j];
        R[k] ^= f; matched |= f;
    }
    bool ok = (nz & ~matched) == 0u;
#pragma unroll
    for (int i = 0; i < 9; i++) ok &= dh_block_decode_rows<4>(T.h1511, T.lut_h1511, R[i]);
    // 96 information bits: row 0 carries 3 reserved + 8, rows 1..8 carry 11 each (:45-56)
    uint64_t acc = 0; int nacc = 0, ob = 0;
    out3[0] = out3[1] = out3[2] = 0u;
#pragma unroll
    for (int r = 0; r < 9; r++) {
        const int nb = r == 0 ? 8 : 11;
        acc = (acc << nb) | ((R[r] >> 4) & ((1u << nb) - 1u)); nacc += nb;
        while (nacc >= 8) { out3[ob >> 2] |= (uint32_t) ((acc >> (nacc - 8)) & 0xFFu) << (8 * (ob & 3)); ob++; nacc -= 8; }
    }
    return ok;
}

// EmbeddedCollector::getLc (embedded.cpp:32-94), lane-local: the 16 collected bytes as 4 big-endian words -> 9 LC bytes
DH_HD bool dh_dmr_embedded_lc_lane(const DhFecTables& T, const uint32_t* data, uint8_t* lc) {
    uint32_t m[8];
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        uint32_t row = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = (data[j] >> (7 - r)) & 0x01010101u;
            row |= ((x * 0x10204080u) >> 28) << (12 - 4 * j);
        }
        if (r < 7) ok &= dh_block_decode_rows<5>(T.h1611, T.lut_h1611, row);
        m[r] = row;
    }
    uint32_t parity = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) parity ^= m[i];
    ok &= parity == 0u;
    uint64_t acc = 0; int nacc = 0, ob = 0; uint32_t received = 0, sum = 0;
#pragma unroll
    for (int r = 0; r < 7; r++) {
        const int nb = r < 2 ? 11 : 10;
        acc = (acc << nb) | ((m[r] >> (16 - nb)) & ((1u << nb) - 1)); nacc += nb;
        while (nacc >= 8) { lc[ob] = (uint8_t) (acc >> (nacc - 8)); sum += lc[ob]; ob++; nacc -= 8; }
        if (r >= 2) received |= ((m[r] >> 5) & 1u) << (4 - (r - 2));
    }
    return ok && (sum % 31u) == received;
}

// pass A: lane k takes the burst at bit `off0 + 144 k` of the planes
DH_HD void dh_dmr_pass_a(const DhDecParams& P, const DhFecTables& T, const DhDecShared& S, uint32_t off0, DH_LANE_STRUCT_REF(DhDmrLane, L)) {
    DH_FOR_LANES(lane) {
        DhDmrLane& me = DH_LS(L, lane);
        const uint32_t o = off0 + 144u * (uint32_t) lane, w0 = o >> 5, sh = o & 31u;
        uint32_t a[6], b[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { a[i] = S.dmr.plane_h[w0 + i]; b[i] = S.dmr.plane_l[w0 + i]; }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            me.hw[i] = (uint32_t) ((((uint64_t) a[i + 1] << 32) | a[i]) >> sh);
            me.lw[i] = (uint32_t) ((((uint64_t) b[i + 1] << 32) | b[i]) >> sh);
        }
        const uint32_t* hw = me.hw; const uint32_t* lw = me.lw;
        uint32_t sm = 0;
        // CACH: TACT bits = bit 1 of dibits 0, 2, 4, 6, 7, 9, 11, first one in the MSB (cach.cpp:7,11-19); Hamming(7,4)
        const uint32_t c12 = hw[0];
        uint32_t tact = ((c12 & 1u) << 6) | (((c12 >> 2) & 1u) << 5) | (((c12 >> 4) & 1u) << 4) | (((c12 >> 6) & 1u) << 3) |
                        (((c12 >> 7) & 1u) << 2) | (((c12 >> 9) & 1u) << 1) | ((c12 >> 11) & 1u);
        if (dh_block_decode_rows<3>(T.h74, T.lut_h74, tact)) sm |= DH_DS_HAS_TACT | ((tact >> 5) & 1u) << DH_DS_TACT_SLOT_SHIFT;
        // sync slot: dibits 66..89 (dmr_phase.cpp:18-33)
        const uint32_t sync_h = dh_fld<66, 24>(hw), sync_l = dh_fld<66, 24>(lw);
        const int sync_type = dh_dmr_sync_type_bits(sync_h, sync_l);
        if (sync_type > 0) sm |= (uint32_t) sync_type << DH_DS_SYNC_SHIFT;
        // EMB: dibits 66..69 and 86..89 (:123-132), QR(16,7); the 32 embedded bits between them (:141-145)
        uint32_t emb = (dh_pack_msb(sync_h & 15u, sync_l & 15u, 4) << 8) | dh_pack_msb(sync_h >> 20, sync_l >> 20, 4);
        if (dh_block_decode_rows<9>(T.qr, T.lut_qr, emb)) sm |= DH_DS_EMB_OK | ((emb >> 9) & 3u) << DH_DS_LCSS_SHIFT | ((emb >> 12) & 15u) << DH_DS_EMB_CC_SHIFT;
        me.frag = dh_pack_msb((sync_h >> 4) & 0xFFFFu, (sync_l >> 4) & 0xFFFFu, 16);
        // slot type: dibits 61..65 and 90..94 (:236-245), Golay(20,8)
        uint32_t slot_type = (dh_pack_msb(dh_fld<61, 5>(hw), dh_fld<61, 5>(lw), 5) << 10) | dh_pack_msb(dh_fld<90, 5>(hw), dh_fld<90, 5>(lw), 5);
        if (dh_block_decode_rows<12>(T.g208, P.T->lut_g208, slot_type))
            sm |= DH_DS_ST_OK | ((slot_type >> 12) & 15u) << DH_DS_DT_SHIFT | ((slot_type >> 16) & 15u) << DH_DS_ST_CC_SHIFT;
        // BPTC(196,96) of the burst taken as a data burst: received bit r = dibit r / 2 of the 98 info dibits (12..60, 95..143), its
        // bit 1 first (:256-269)
        const bool bok = dh_dmr_bptc_lane(T, [hw, lw](int r) -> uint32_t {
            const int d = r >> 1, pos = d < 49 ? 12 + d : 46 + d;
            return (((r & 1) ? lw : hw)[pos >> 5] >> (pos & 31)) & 1u; }, me.bptc);
        if (bok) sm |= DH_DS_BPTC_OK;
        if (sm & DH_DS_ST_OK) {                                   // a data burst's events (:246-300); rate 3/4 data is not decoded (:251-253)
            const uint32_t dt = (sm >> DH_DS_DT_SHIFT) & 15u;
            uint32_t df = DH_DF_SLOTTYPE;
            if (dt != 8u && bok) df |= DH_DF_BPTC | ((dt == 1u || dt == 2u || dt == 9u) ? (uint32_t) DH_DF_BPTC_TAIL : 0u);      // LC (:283-285) / soft reset (:286-293)
            sm |= (df >> 6) << DH_DS_DFLAGS_SHIFT;
        }
        me.summary = sm;
        me.flags = 0u;
    }
}

// the FramePhase's members (dmr_phase.hpp:51-60) while a chunk is walked: scalar registers
struct DhDmrMachine {
    int slot, stab, sync_count, st0, st1, ss0, ss1, active, filter, sf0, sf1, eo0, eo1;
    DH_HD void load(DhState& s) {
        slot = (int) s[DS_SLOT]; stab = (int) s[DS_SLOT_STABILITY]; sync_count = (int) s[DS_SYNC_COUNT];
        st0 = (int) s[DS_SYNC_TYPE0]; st1 = (int) s[DS_SYNC_TYPE1]; ss0 = (int) s[DS_SLOT_SYNC0]; ss1 = (int) s[DS_SLOT_SYNC1];
        active = (int) s[DS_ACTIVE_SLOT]; filter = (int) s[DS_SLOT_FILTER]; sf0 = (int) s[DS_SUPERFRAME0]; sf1 = (int) s[DS_SUPERFRAME1];
        eo0 = (int) s[DS_EMB_OFF0]; eo1 = (int) s[DS_EMB_OFF1];
    }
    DH_HD void store(DhState& s) const {
        s[DS_SLOT] = (uint32_t) slot; s[DS_SLOT_STABILITY] = (uint32_t) stab; s[DS_SYNC_COUNT] = (uint32_t) sync_count;
        s[DS_SYNC_TYPE0] = (uint32_t) st0; s[DS_SYNC_TYPE1] = (uint32_t) st1; s[DS_SLOT_SYNC0] = (uint32_t) ss0; s[DS_SLOT_SYNC1] = (uint32_t) ss1;
        s[DS_ACTIVE_SLOT] = (uint32_t) active; s[DS_SUPERFRAME0] = (uint32_t) sf0; s[DS_SUPERFRAME1] = (uint32_t) sf1;
        s[DS_EMB_OFF0] = (uint32_t) eo0; s[DS_EMB_OFF1] = (uint32_t) eo1;
    }
};

// pass B: FramePhase::process (dmr_phase.cpp:65-254) for bursts 0 .. n - 1 of the chunk on their summaries.  Returns the number of
// bursts consumed; `nflag` = bursts that got a flag word (one more when the last one sent the decoder back to its SyncPhase
// without being consumed, :163-170, :201-204); `to_sync` says so.  `room` = bytes left in the output row.
// The machine crosses the burst loop in THREE scalar registers -- g = slot + 1 | (stability + 128) << 2 | sync count << 10 | active
// slot + 1 << 13; x0 / x1 = sync type + 1 | slot sync count << 2 | superframe << 5 | embedded offset << 8 -- and is unpacked into
// locals per burst.  (Measured, profiles/r05_a_ab_logs.txt: with its thirteen members as thirteen loop-carried registers every join
// of the branches copies a dozen of them: 336 -> 323 scalar instructions per 1 000-sample run of the chain, decoder alone 0.75 ->
// 0.67 ms.  Two other variations lost: the members of the slot picked once and updated by selects instead of branches -- MORE scalar
// instructions, 0.84 ms -- and a short cut for data-sync bursts that find their slot in the "data sync held" state, a fixed
// point of the machine: it costs the other bursts more than it saves, 0.88 ms.)
DH_HD uint32_t dh_dmr_pass_b(DhDmrMachine& M, DhState& s, DhDecShared& S, DH_LANE_STRUCT_REF(DhDmrLane, L), uint32_t n, uint32_t& room,
                                    uint32_t& nflag, bool& to_sync, bool& overflow) {
    to_sync = false;
    uint32_t g = (uint32_t) (M.slot + 1) | (uint32_t) (M.stab + 128) << 2 | (uint32_t) M.sync_count << 10 | (uint32_t) (M.active + 1) << 13;
    uint32_t x0 = (uint32_t) (M.st0 + 1) | (uint32_t) M.ss0 << 2 | (uint32_t) M.sf0 << 5 | (uint32_t) M.eo0 << 8;
    uint32_t x1 = (uint32_t) (M.st1 + 1) | (uint32_t) M.ss1 << 2 | (uint32_t) M.sf1 << 5 | (uint32_t) M.eo1 << 8;
    const int filter = M.filter;
    uint32_t k = 0;
    bool stop = false;
    for (; k < n && !stop; k++) {
        const uint32_t sm = DH_LS_READ(L, summary, k);
        uint32_t fl = 0;
        int slot = (int) (g & 3u) - 1, stab = (int) ((g >> 2) & 255u) - 128, sync_count = (int) ((g >> 10) & 7u), active = (int) ((g >> 13) & 3u) - 1;
        const int tact_slot = (int) ((sm >> DH_DS_TACT_SLOT_SHIFT) & 1u);
        const int next = (slot ^ 1) & 0xFF;                      // unsigned char next = slot ^ 1  (:69)
        if (DH_LIKELY(sm & DH_DS_HAS_TACT)) {
            if (DH_UNLIKELY(tact_slot != next)) {
                if (stab < 5) {
                    stab = 0; slot = tact_slot;
                    const int other = slot ^ 1;
                    if (other) x1 &= ~3u; else x0 &= ~3u;         // syncTypes[other] = -1
                    fl |= DH_DF_RESET_OTHER;                      // the OTHER slot, after a TACT slot switch (:80)
                    if (active == other) active = -1;
                } else {
                    stab--;
                    if (slot != -1) slot = next;
                }
            } else {
                if (++stab > 100) stab = 100;
                slot = next;
            }
        } else if (slot != -1) {
            if (stab-- < -100) stab = -100;
            slot = next;
        }
        if (DH_LIKELY(slot != -1)) {
            const uint32_t x = slot ? x1 : x0;
            int st = (int) (x & 3u) - 1, ss = (int) ((x >> 2) & 7u), sf = (int) ((x >> 5) & 7u); uint32_t eo = (x >> 8) & 7u;
            fl |= slot ? DH_DF_SLOT : 0u;
            const int sync_type = (int) ((sm >> DH_DS_SYNC_SHIFT) & 3u);
            bool lost = false;
            if (sync_type > 0) {
                if (++sync_count > 5) sync_count = 5;
                if (++ss > 5) ss = 5;
                if (st == DH_SYNCTYPE_VOICE && sync_type != DH_SYNCTYPE_VOICE) fl |= DH_DF_SOFT;
                st = sync_type;
                fl |= DH_DF_SYNC;
                sf = 0; eo = 0;
            } else if (st == DH_SYNCTYPE_VOICE && sf < 5) {
                sf++;
                if (sm & DH_DS_EMB_OK) {
                    if (++sync_count > 5) sync_count = 5;
                    if (++ss > 5) ss = 5;
                    fl |= DH_DF_EMB;
                    const uint32_t lcss = (sm >> DH_DS_LCSS_SHIFT) & 3u;
                    const uint32_t dbase = slot ? DS_EMB_DATA1 : DS_EMB_DATA0;
                    if (lcss == 1) eo = 0;                                       // LCSS_START: reset, then collect
                    if (lcss != 0) {                                             // START / CONTINUATION / STOP collect
                        if (eo <= 3) { s[dbase + eo] = DH_LS_READ(L, frag, k); eo++; }
                    }
                    if (lcss == 2) {                                             // LCSS_STOP: pass C decodes what has been collected
                        if (eo >= 3) {
                            const uint32_t d0 = s[dbase], d1 = s[dbase + 1u], d2 = s[dbase + 2u], d3 = s[dbase + 3u];
                            DH_FOR_LANES(lane) {
                                if (DH_IS_LANE0(lane)) { S.dmr.emb_words[k][0] = d0; S.dmr.emb_words[k][1] = d1; S.dmr.emb_words[k][2] = d2; S.dmr.emb_words[k][3] = d3; }
                            }
                            fl |= DH_DF_EMB_LC;
                        }
                        eo = 0;
                    }
                } else lost = true;
            } else {
                sf = 0; eo = 0;
                lost = true;
            }
            if (lost) {
                if (--ss < 0) {                                                  // dmr_phase.cpp:175-182 == :194-200
                    ss = 0; st = -1;
                    fl |= DH_DF_SLOT_RESET;
                    if (active == slot) active = -1;
                }
                if (DH_UNLIKELY(--sync_count < 0)) { fl |= DH_DF_META_RESET; to_sync = true; stop = true; sync_count = 0; }
            }
            if (DH_LIKELY(!stop)) {
                if (st == DH_SYNCTYPE_VOICE) {
                    if (((slot + 1) & filter) && (active == -1 || active == slot)) {
                        active = slot;
                        if (DH_UNLIKELY(room < 27u)) overflow = true; else { fl |= DH_DF_VOICE; room -= 27u; }
                    }
                } else {
                    if (active == slot) active = -1;
                    if (st == DH_SYNCTYPE_DATA) fl |= (sm >> (DH_DS_DFLAGS_SHIFT - 6)) & (DH_DF_SLOTTYPE | DH_DF_BPTC | DH_DF_BPTC_TAIL);      // (:246-300, from pass A)
                    else fl |= DH_DF_SLOT_RESET2;
                }
            }
            const uint32_t xn = (uint32_t) (st + 1) | (uint32_t) ss << 2 | (uint32_t) sf << 5 | eo << 8;
            if (slot) x1 = xn; else x0 = xn;
        }
        g = (uint32_t) (slot + 1) | (uint32_t) (stab + 128) << 2 | (uint32_t) sync_count << 10 | (uint32_t) (active + 1) << 13;
        DH_LS_WRITE(L, flags, k, fl);
        if (DH_UNLIKELY(overflow)) stop = true;
    }
    // (k counts the bursts that got a flag word; the one that sent the decoder back to its SyncPhase is not consumed)
    M.slot = (int) (g & 3u) - 1; M.stab = (int) ((g >> 2) & 255u) - 128; M.sync_count = (int) ((g >> 10) & 7u); M.active = (int) ((g >> 13) & 3u) - 1;
    M.st0 = (int) (x0 & 3u) - 1; M.ss0 = (int) ((x0 >> 2) & 7u); M.sf0 = (int) ((x0 >> 5) & 7u); M.eo0 = (int) ((x0 >> 8) & 7u);
    M.st1 = (int) (x1 & 3u) - 1; M.ss1 = (int) ((x1 >> 2) & 7u); M.sf1 = (int) ((x1 >> 5) & 7u); M.eo1 = (int) ((x1 >> 8) & 7u);
    nflag = k;
    return to_sync ? k - 1u : k;
}

// exclusive prefix sum over the lanes of a small per-lane count (< 16), by votes
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
#define DH_LANES_BELOW(mask, lane) ((uint32_t) __builtin_amdgcn_mbcnt_hi((uint32_t) ((mask) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) (mask), 0u)))
#else
#define DH_LANES_BELOW(mask, lane) ((uint32_t) dh_popc64((mask) & (((uint64_t) 1 << (lane)) - 1u)))
#endif

DH_HD void dh_dmr_channel(const DhDecParams& P, uint32_t ch, DhDecShared& S, uint32_t sym_base = 0, bool append = false) {
    DhDecCtx c;
    c.P = &P; c.T = &dh_lds_tables(S);
    uint32_t* const st_global = P.state + (size_t) ch * P.state_stride;
    DhState s; s.load(st_global);
    c.st = &s;
    c.out = P.out + (size_t) ch * P.out_stride;
    c.ev = P.events ? P.events + (size_t) ch * P.ev_stride : nullptr;
    c.nout = dh_uniform(append ? P.out_count[ch] : 0u); c.nev = dh_uniform(append && P.ev_count ? P.ev_count[ch] : 0u); c.overflow = false;      // (append: the second part of a split push, k_chain)
    c.consumed = s[DS_CONSUMED];
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    c.writer = threadIdx.x == 0;
#else
    c.writer = true;
#endif
    uint8_t* const carry_buf = P.carry + (size_t) ch * P.carry_stride;
    DhSymView syms; syms.carry = S.carry; syms.nc = s[DS_CARRY]; syms.fresh = P.syms + (size_t) ch * P.sym_stride + sym_base;
    syms.nfresh = P.sym_count[ch] - sym_base; syms.win = S.symwin; syms.wbase = 0; syms.wlen = 0;
    const uint32_t total = syms.nc + syms.nfresh;
    dh_stage_decoder_lds(P, S, carry_buf, syms.nc);
    const DhFecTables& T = dh_lds_tables(S);
    uint32_t pos = 0;
    uint32_t phase = s[DS_PHASE];
    DhDmrMachine M; M.load(s);
    DH_LANE_STRUCT(DhDmrLane, L);

    for (;;) {
        const uint32_t avail = total - pos;
        if (DH_UNLIKELY(phase == 0)) {                                 // SyncPhase (dmr_phase.cpp:35-47)
            if (!(avail > 90)) break;
            DhPlanes& pl = S.planes;
            dh_view_ensure(syms, pos, 192);
            dh_load_planes(syms, pos, total, pl, 3);
            uint64_t hits = 0;
            DH_FOR_LANES(lane) {
                const bool valid = avail - (uint32_t) lane > 90 && avail > (uint32_t) lane;
                const bool hit = valid && dh_dmr_sync_type(pl, 66 + lane) > 0;
                DH_BALLOT_ACC(hits, hit, lane);
            }
            if (hits) {
                const uint32_t l = (uint32_t) dh_ffs64(hits);
                pos += l; c.consumed += l;
                phase = 1; dh_dmr_enter_frame_phase(s); M.load(s);
            } else {
                const uint32_t adv = dh_min<uint32_t>(64u, avail - 90u);
                pos += adv; c.consumed += adv;
            }
            continue;
        }
        // FramePhase (dmr_phase.cpp:61-302): a chunk of bursts
        if (!(avail > 144)) break;
        const uint32_t a16 = (uint32_t) ((uintptr_t) syms.fresh & 15u);
        const uint32_t off0 = (pos - syms.nc + a16) & 15u;             // dibits between the planes' origin and the first burst
        uint32_t n = dh_min<uint32_t>((avail - 1u) / 144u, (DH_DMR_PLANE_GROUPS * 16u - off0) / 144u);
        const uint32_t ngroups = (off0 + 144u * n + 15u) / 16u;
        DH_DMARK("planes");
        dh_build_chunk_planes(syms, (int32_t) pos - (int32_t) off0, ngroups, total, S.dmr.plane_h, S.dmr.plane_l);
        DH_DMARK("passA");
        dh_dmr_pass_a(P, T, S, off0, L);
        DH_DMARK("passA end");
        const DhState s_before = s; const DhDmrMachine m_before = M;
        uint32_t nflag = 0, ncons = 0; bool to_sync = false;
        uint64_t voice_mask = 0;
        DH_LANE_VALUE(uint32_t, ev_at);
        DH_LANE_VALUE(uint32_t, ev_set);
        uint32_t nev_chunk = 0;
        for (;;) {
            uint32_t room = dh_uniform(P.out_cap - c.nout);
            bool ovf = false;
            ncons = dh_dmr_pass_b(M, s, S, L, n, room, nflag, to_sync, ovf);
            DH_DMARK("passC");
            DH_BARRIER();                                              // the embedded-signalling words are in LDS
            // pass C, part 1: which events does every burst emit, and where do they go
            uint64_t cnt_bits[4] = { 0, 0, 0, 0 }, lcmask = 0;
            DH_FOR_LANES(lane) {
                DhDmrLane& me = DH_LS(L, lane);
                uint32_t fl = (uint32_t) lane < nflag ? me.flags : 0u;
                DH_BALLOT_ACC(lcmask, (fl & DH_DF_EMB_LC) != 0u, lane);
                me.flags = fl;
            }
            if (lcmask) {
                DH_FOR_LANES(lane) {
                    DhDmrLane& me = DH_LS(L, lane);
                    if (me.flags & DH_DF_EMB_LC) {
                        uint8_t lc[9];
                        const uint32_t data[4] = { S.dmr.emb_words[lane][0], S.dmr.emb_words[lane][1], S.dmr.emb_words[lane][2], S.dmr.emb_words[lane][3] };
                        if (dh_dmr_embedded_lc_lane(T, data, lc)) {
                            // the 9 LC bytes take the place of the BPTC words (a burst with an EMB is a voice burst: no BPTC event)
                            me.bptc[0] = (uint32_t) lc[0] | (uint32_t) lc[1] << 8 | (uint32_t) lc[2] << 16 | (uint32_t) lc[3] << 24;
                            me.bptc[1] = (uint32_t) lc[4] | (uint32_t) lc[5] << 8 | (uint32_t) lc[6] << 16 | (uint32_t) lc[7] << 24;
                            me.bptc[2] = (uint32_t) lc[8];
                        } else me.flags &= ~(uint32_t) DH_DF_EMB_LC;
                    }
                }
            }
            DH_FOR_LANES(lane) {
                const uint32_t fl = DH_LS(L, lane).flags;
                const uint32_t cnt = (uint32_t) dh_popc32(fl & DH_DF_EVENTS);
                DH_LV(ev_set, lane) = fl;
                DH_BALLOT_ACC(cnt_bits[0], (cnt & 1u) != 0u, lane); DH_BALLOT_ACC(cnt_bits[1], (cnt & 2u) != 0u, lane);
                DH_BALLOT_ACC(cnt_bits[2], (cnt & 4u) != 0u, lane); DH_BALLOT_ACC(cnt_bits[3], (cnt & 8u) != 0u, lane);
                DH_BALLOT_ACC(voice_mask, (fl & DH_DF_VOICE) != 0u, lane);
            }
            uint64_t over = 0;
            const uint32_t ev_room = c.ev ? P.ev_cap - c.nev : 0xFFFFFFFFu;
            DH_FOR_LANES(lane) {
                const uint32_t at = DH_LANES_BELOW(cnt_bits[0], lane) + 2u * DH_LANES_BELOW(cnt_bits[1], lane) +
                                    4u * DH_LANES_BELOW(cnt_bits[2], lane) + 8u * DH_LANES_BELOW(cnt_bits[3], lane);
                const uint32_t cnt = (uint32_t) dh_popc32(DH_LV(ev_set, lane) & DH_DF_EVENTS);
                DH_LV(ev_at, lane) = at;
                DH_BALLOT_ACC(over, c.ev != nullptr && at + cnt > ev_room, lane);
            }
            nev_chunk = (uint32_t) (dh_popc64(cnt_bits[0]) + 2 * dh_popc64(cnt_bits[1]) + 4 * dh_popc64(cnt_bits[2]) + 8 * dh_popc64(cnt_bits[3]));
            if (DH_UNLIKELY(over != 0)) {
                // The event row overflows in burst k_ov: the reference-shaped loop drops what does not fit, finishes that burst and
                // stops (dh_emit / `if (c.overflow) break`).  Walk the chunk again up to that burst only.
                const uint32_t k_ov = (uint32_t) dh_ffs64(over);
                c.overflow = true;
                if (k_ov + 1u < nflag) { s = s_before; M = m_before; n = k_ov + 1u; voice_mask = 0; continue; }
            }
            if (ovf) c.overflow = true;
            break;
        }
        // pass C, part 2: the event records
        if (c.ev != nullptr && nev_chunk) {
            dh_event* const evrow = c.ev + c.nev;
            const uint32_t ev_room = P.ev_cap - c.nev;
            const uint32_t sym0 = c.consumed;
            DH_FOR_LANES(lane) {
                const DhDmrLane& me = DH_LS(L, lane);
                const uint32_t fl = DH_LV(ev_set, lane), sm = me.summary;
                uint32_t at = DH_LV(ev_at, lane);
                const uint32_t sym_index = sym0 + 144u * (uint32_t) lane;
                const uint32_t slot = (fl & DH_DF_SLOT) ? 1u : 0u;
                const uint32_t dt = (sm >> DH_DS_DT_SHIFT) & 15u;
#pragma unroll
                for (int e = 0; e < 10; e++) {
                    if (!((fl >> e) & 1u)) continue;
                    uint32_t type = 0, a = slot, b = 0, len = 0, p0 = 0, p1 = 0, p2 = 0;
                    switch (e) {
                    case 0: type = DH_EV_DMR_SLOT_RESET; a = slot ^ 1u; b = 1; break;
                    case 1: type = DH_EV_DMR_SYNC; b = (sm >> DH_DS_SYNC_SHIFT) & 3u; len = 1; p0 = (fl & DH_DF_SOFT) ? 1u : 0u; break;
                    case 2: type = DH_EV_DMR_EMB; b = (sm >> DH_DS_LCSS_SHIFT) & 3u; len = 1; p0 = (sm >> DH_DS_EMB_CC_SHIFT) & 15u; break;
                    case 3: type = DH_EV_DMR_LC; b = 1; len = 9; p0 = me.bptc[0]; p1 = me.bptc[1]; p2 = me.bptc[2] & 0xFFu; break;
                    case 4: type = DH_EV_DMR_SLOT_RESET; break;
                    case 5: type = DH_EV_DMR_META_RESET; a = 0; break;
                    case 6: type = DH_EV_DMR_SLOTTYPE; b = dt; len = 1; p0 = (sm >> DH_DS_ST_CC_SHIFT) & 15u; break;
                    case 7: type = DH_EV_DMR_SLOT_RESET; break;
                    case 8: type = DH_EV_DMR_BPTC; b = dt; len = 12; p0 = me.bptc[0]; p1 = me.bptc[1]; p2 = me.bptc[2]; break;
                    default:
                        if (dt == 1u) { type = DH_EV_DMR_LC; b = 0; len = 9; p0 = me.bptc[0]; p1 = me.bptc[1]; p2 = me.bptc[2] & 0xFFu; }
                        else { type = DH_EV_DMR_SOFT_RESET; b = dt; }
                        break;
                    }
                    if (at < ev_room) {
                        uint32_t* const w = reinterpret_cast<uint32_t*>(evrow + at);
                        w[0] = sym_index; w[1] = type | a << 8 | b << 16 | len << 24;
                        w[2] = p0; w[3] = p1; w[4] = p2; w[5] = 0u; w[6] = 0u; w[7] = 0u;
                    }
                    at++;
                }
            }
            c.nev += dh_min<uint32_t>(nev_chunk, ev_room);
        }
        // pass C, part 3: voice payloads (dmr_phase.cpp:213-226): 108 dibits -> 27 bytes, first dibit in the top bits
        if (voice_mask) {
            DH_BARRIER();                                              // the embedded-signalling words have been read
            DH_FOR_LANES(lane) {
                const DhDmrLane& me = DH_LS(L, lane);
                if ((voice_mask >> lane) & 1u) {
                    const uint32_t rank = DH_LANES_BELOW(voice_mask, lane);
                    // dibits 12..65 and 90..143 as one string of 108
                    const uint32_t h0 = dh_fld<12, 32>(me.hw), h1 = dh_fld<44, 22>(me.hw) | dh_fld<90, 10>(me.hw) << 22, h2 = dh_fld<100, 32>(me.hw), h3 = dh_fld<132, 12>(me.hw);
                    const uint32_t l0 = dh_fld<12, 32>(me.lw), l1 = dh_fld<44, 22>(me.lw) | dh_fld<90, 10>(me.lw) << 22, l2 = dh_fld<100, 32>(me.lw), l3 = dh_fld<132, 12>(me.lw);
                    const uint32_t hh[4] = { h0, h1, h2, h3 }, ll[4] = { l0, l1, l2, l3 };
#pragma unroll
                    for (int j = 0; j < 7; j++) {
                        const uint32_t h16 = (hh[j >> 1] >> (16 * (j & 1))) & 0xFFFFu, l16 = (ll[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                        const uint32_t be = j < 6 ? dh_pack_msb(h16, l16, 16) : dh_pack_msb(h16, l16, 12) << 8;      // first dibit on top
                        S.dmr.voice[rank][j] = (be >> 24) | ((be >> 8) & 0xFF00u) | ((be << 8) & 0xFF0000u) | (be << 24);
                    }
                }
            }
            DH_BARRIER();
            const uint32_t nbytes = 27u * (uint32_t) dh_popc64(voice_mask);
            uint8_t* const o = c.out + c.nout;
            const uint8_t* const stage = reinterpret_cast<const uint8_t*>(S.dmr.voice);
            DH_FOR_LANES(lane) {
                for (uint32_t j = (uint32_t) lane; j < nbytes; j += DH_WAVE) {
                    const uint32_t r = j / 27u;
                    o[j] = stage[r * 28u + (j - r * 27u)];
                }
            }
            c.nout += nbytes;
            DH_BARRIER();
        }
        DH_DMARK("chunk end");
        pos += 144u * ncons; c.consumed += 144u * ncons;
        if (to_sync) phase = 0;
        if (c.overflow) break;
    }

    // carry the unread symbols to the front of the buffer
    const uint32_t rem = total - pos;
    dh_view_ensure(syms, pos, rem < DH_SYM_CARRY_MAX ? rem : DH_SYM_CARRY_MAX);
    DH_FOR_LANES(lane) {
        // sources are LDS (carried part / window), destination is the global carry row: no overlap to worry about
        for (uint32_t j = lane; j < rem && j < DH_SYM_CARRY_MAX; j += DH_WAVE) carry_buf[j] = (uint8_t) dh_view_at(syms, pos + j);
        if (DH_IS_LANE0(lane)) {
            P.out_count[ch] = c.nout;
            if (P.ev_count) P.ev_count[ch] = c.nev;
            if ((c.overflow || rem > DH_SYM_CARRY_MAX) && P.overflow) *P.overflow = 1u;
        }
    }
    if (phase == 1) M.store(s);
    s[DS_PHASE] = phase; s[DS_CONSUMED] = c.consumed;
    s[DS_CARRY] = rem < DH_SYM_CARRY_MAX ? rem : DH_SYM_CARRY_MAX;
    s.store(st_global);
    DH_BARRIER();
}


// =============================================================================================
// YSF
// =============================================================================================

// rate-1/2 K=5 Viterbi for up to four codewords at once: codeword g on lanes 16g..16g+15, one
// trellis state per lane.  Same decisions as the reference's register-exchange decoder
// (src/ysf_decoder/trellis.c:32-109): uint8 path metrics starting at 0 in every state, the k = 0
// predecessor wins ties, the best final state is the lowest index among the minimum metric;
// survivors are recovered by trace-back over the stored decision votes instead of copying
// sixteen bit strings per step.
// Final step shared by both forward passes: pick the best end state of every codeword (lowest index among the
// minimum metric, trellis.c:94-98) and trace the decisions back.  Output bits are assembled in a register and
// stored once per byte; the eight decision words of a byte are fetched together (their addresses do not depend
// on the trace-back state).
// four dibits packed MSB first (the reference's input format, trellis.c:47-49) -> one dibit per byte, first in byte 0:
// the layout of S.vit_in, whose words are v_perm selectors for the branch-metric tables as they stand
DH_HD uint32_t dh_vit_word(uint32_t packed) {
    return ((packed >> 6) & 3u) | ((packed >> 4) & 3u) << 8 | ((packed >> 2) & 3u) << 16 | (packed & 3u) << 24;
}

DH_HD void dh_viterbi_finish(DhDecShared& S, const int* sizes, int fin) {
    DH_FOR_LANES(lane) {
        if ((lane & 15) == 0) {
            const int g = lane >> 4;
            const int size = g == 0 ? sizes[0] : g == 1 ? sizes[1] : g == 2 ? sizes[2] : sizes[3];
            if (size > 0) {
                uint32_t best = 0, bm = S.vit_metric[fin][g * 16];
                for (uint32_t i = 1; i < 16; i++) {
                    const uint32_t m = S.vit_metric[fin][g * 16 + i];
                    if (m < bm) { bm = m; best = i; }
                }
                S.vit_best_metric[g] = (uint8_t) bm;
                const int nbytes = (size + 7) >> 3;
                for (int b = nbytes; b < 24; b++) S.vit_out[g][b] = 0;
                // trace back without masking the state register: H = (H << 1) | k keeps the state in its low four bits
                // and pushes the older output bits up, so after the eight steps of a byte they sit in bits 4..11,
                // first step highest: the byte is their bit reversal.  A step is: 16 decision bits of this codeword
                // (read as a halfword), pick bit `state`, shift it in.
                uint32_t H = best;
                const uint16_t* dec16 = reinterpret_cast<const uint16_t*>(S.vit_dec) + g;
                for (int b = nbytes - 1; b >= 0; b--) {
                    uint32_t d[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) d[j] = dec16[4 * ((b * 8 + j) < 192 ? (b * 8 + j) : 191)];
#pragma unroll
                    for (int j = 7; j >= 0; j--) {
                        if (b * 8 + j < size) H = (H << 1) | ((d[j] >> (H & 15u)) & 1u);
                    }
                    S.vit_out[g][b] = (uint8_t) (dh_brev32((H >> 4) & 0xFFu) >> 24);
                }
            }
        }
    }
    DH_BARRIER();
}

// Clean-codeword shortcut of the reference's decoder (src/ysf_decoder/trellis.c:32-109).  The code is G1 = 1 + D^3 + D^4 (bit 1 of a
// dibit), G2 = 1 + D + D^2 + D^4 (bit 0).  With h, l the two received bit streams:
//   * syndrome  s_t = h_t + h_(t-1) + h_(t-2) + h_(t-4) + l_t + l_(t-3) + l_(t-4)   (h G2 = l G1), t = 4 .. N - 1: all zero <=> the N
//     dibits are EXACTLY the encoder's output for some start state and some message (2 N bits, N + 4 free, N - 4 independent checks);
//   * the encoder map (start state, message) -> dibits is injective for N >= 4 (rank N + 4, tools/trellis_rank.py), so that path is the
//     ONLY one of metric 0.  The reference starts every state at metric 0 and keeps, per state, the smaller of two candidates (k = 0 on
//     ties): on the true path the true predecessor arrives with 0 and the other one with at least 2 (the two transitions into a state
//     differ in both bits), so it survives strictly at every step; no other end state can have metric 0 (it would be a second path of
//     metric 0), and no metric wraps (every state is within four steps = 8 errors of the true path).  The reference therefore returns
//     metric 0 and the message bits of that path, whatever its tie rules would do elsewhere;
//   * the message comes straight out of the dibits: D^2 = (1 + D + D^2) G1 + (1 + D^2) G2, so u_t = h_(t+2) + h_(t+1) + h_t + l_(t+2) + l_t
//     for t <= N - 3, and the last two bits from G1: u_t = h_t + u_(t-3) + u_(t-4).
// Used lane-locally, one codeword per lane, by the YSF decoder (dh_ysf_clean100 / dh_ysf_decode_ahead) and by the batch entry dh_trellis for
// 100-dibit codewords; tests/test_fec.py checks it there against the reference compiled in place for every start state, and that one
// flipped bit anywhere falls through to the full decoder.  (Round 4's wave-level form -- ~150 scalar instructions per codeword, all
// four codewords of a pass clean or nothing -- lost in the YSF pipe: 8 % of the passes qualified.  profiles/r04_b_ab_logs.txt.)


#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
// One trellis step of the forward pass below; C = step & 63 = the lane of the decision registers that takes its vote, an
// IMMEDIATE of v_writelane (the steps are spelled out, not looped over: as a loop variable the lane went through m0 --
// three scalar moves per step and 64 scalar registers of constants, which the decoders then spilled).  The vote comes
// straight out of a vector compare: v_writelane may not read a scalar register a vector instruction wrote less than four
// wait states ago -- without the s_nop the YSF decoder ran 2.2 times SLOWER (3.1 -> 7.0 ms), with it 3 % faster than the
// m0 form (profiles/r03_d_ab_logs.txt); inline asm gets no hazard handling from the compiler.
#define DH_VIT_IMM_NOP "s_nop 3\n\t"
template <bool NXDN, bool RAGGED, int C>
__device__ __forceinline__ void dh_vitb_step(uint32_t& m, uint32_t h0, uint32_t h1, int src0, int src1, uint32_t i, int blk, int steps, int mysize,
                                             uint32_t& dlo, uint32_t& dhi) {
    constexpr int q = C & 3;
    const int pos = blk * 64 + C;
    if (pos < steps) {
        const uint32_t a = (uint32_t) __shfl((int) m, src0), b = (uint32_t) __shfl((int) m, src1);
        const uint32_t m0 = (a + ((h0 >> (8 * q)) & 0xFFu)) & 0xFFu;
        const uint32_t m1 = (b + ((h1 >> (8 * q)) & 0xFFu)) & 0xFFu;
        bool take1 = m1 < m0;
        if (NXDN && C < 4) take1 = take1 && (blk > 0 || (i & ((0xFu << C) & 0xFu)) == 0u);
        if (RAGGED) {
            const bool active = pos < mysize;
            take1 = take1 && active;
            if (active) m = take1 ? m1 : m0;
        } else if (NXDN && C < 4) {
            m = take1 ? m1 : m0;
        } else {
            m = m1 < m0 ? m1 : m0;
        }
        const uint64_t dec = __builtin_amdgcn_ballot_w64(take1);
        asm(DH_VIT_IMM_NOP "v_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(dlo), "+v"(dhi) : "s"((uint32_t) dec), "s"((uint32_t) (dec >> 32)), "n"(C));
    }
}
// gfx950 forward pass: path metrics stay in registers, the two predecessor metrics come through ds_bpermute
// (__shfl), decisions are wave votes; no LDS traffic or barrier inside the step loop.  Per step and lane: two byte
// adds (SDWA: uint8 wrap for free, like the reference's uint8 metrics), a compare, a min, two v_writelane; the branch
// metrics of four steps come from two v_perm whose selector is the input word itself.
// NXDN = the reference's NXDN flavour (trellis.cpp:35-60); RAGGED = codewords of different non-zero lengths in one
// pass (a shorter one must stop updating at its own length); otherwise zero-length slots simply run along and are
// ignored.
template <bool NXDN = false, bool RAGGED = NXDN>
__device__ __forceinline__ void dh_viterbi_wave(DhDecShared& S, const int* sizes) {
    const int lane = (int) threadIdx.x, g = lane >> 4, i = lane & 15;
    int steps = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) steps = sizes[q] > steps ? sizes[q] : steps;
    const int mysize = g == 0 ? sizes[0] : g == 1 ? sizes[1] : g == 2 ? sizes[2] : sizes[3];   // static indices only
    const uint32_t outbit = (uint32_t) i >> 3;
    const uint32_t p0 = ((uint32_t) i << 1) & 0xEu, p1 = p0 | 1u;
    const uint32_t t0 = dh_trellis_out(p0, outbit), t1 = dh_trellis_out(p1, outbit);
    const int src0 = (g << 4) | (int) p0, src1 = src0 | 1;
    // branch metrics as byte tables indexed by the received dibit: byte v = popcount(v ^ t)
    uint32_t tab0 = 0, tab1 = 0;
#pragma unroll
    for (uint32_t v = 0; v < 4; v++) { tab0 |= (uint32_t) __popc(v ^ t0) << (8 * v); tab1 |= (uint32_t) __popc(v ^ t1) << (8 * v); }
    const uint32_t* inw = S.vit_in[g];                  // one dibit per byte, four steps per word
    uint32_t m = 0;
    for (int blk = 0; blk * 64 < steps; blk++) {
        // decisions of 64 steps collect in one register pair, step (64 blk + c) in lane c, written with v_writelane
        uint32_t dlo = 0, dhi = 0;
#define DH_VITB_WORD(WQ) \
        if (blk * 64 + (WQ) * 4 < steps) { \
            const uint32_t w = inw[blk * 16 + (WQ)]; \
            const uint32_t h0 = __builtin_amdgcn_perm(tab0, tab0, w), h1 = __builtin_amdgcn_perm(tab1, tab1, w); \
            dh_vitb_step<NXDN, RAGGED, (WQ) * 4 + 0>(m, h0, h1, src0, src1, (uint32_t) i, blk, steps, mysize, dlo, dhi); \
            dh_vitb_step<NXDN, RAGGED, (WQ) * 4 + 1>(m, h0, h1, src0, src1, (uint32_t) i, blk, steps, mysize, dlo, dhi); \
            dh_vitb_step<NXDN, RAGGED, (WQ) * 4 + 2>(m, h0, h1, src0, src1, (uint32_t) i, blk, steps, mysize, dlo, dhi); \
            dh_vitb_step<NXDN, RAGGED, (WQ) * 4 + 3>(m, h0, h1, src0, src1, (uint32_t) i, blk, steps, mysize, dlo, dhi); \
        }
        DH_VITB_WORD(0) DH_VITB_WORD(1) DH_VITB_WORD(2) DH_VITB_WORD(3) DH_VITB_WORD(4) DH_VITB_WORD(5) DH_VITB_WORD(6) DH_VITB_WORD(7)
        DH_VITB_WORD(8) DH_VITB_WORD(9) DH_VITB_WORD(10) DH_VITB_WORD(11) DH_VITB_WORD(12) DH_VITB_WORD(13) DH_VITB_WORD(14) DH_VITB_WORD(15)
#undef DH_VITB_WORD
        S.vit_dec[blk * 64 + lane] = (uint64_t) dhi << 32 | dlo;
    }
    S.vit_metric[0][lane] = m;
    __syncthreads();
    dh_viterbi_finish(S, sizes, 0);
}
#else
// plain statement of the same recursion for the CPU harness: metrics exchanged through the LDS arrays
template <bool NXDN = false, bool RAGGED = NXDN>
inline void dh_viterbi_wave(DhDecShared& S, const int* sizes /*[4]*/) {
    int steps = 0;
    for (int g = 0; g < 4; g++) steps = sizes[g] > steps ? sizes[g] : steps;
    DH_FOR_LANES(lane) { S.vit_metric[0][lane] = 0; }
    for (int pos = 0; pos < steps; pos++) {
        const int cur = pos & 1;
        uint64_t dec = 0;
        DH_FOR_LANES(lane) {
            const int g = lane >> 4, i = lane & 15;
            bool sel = false;
            uint32_t nm = S.vit_metric[cur][lane];
            if (pos < sizes[g]) {
                const uint32_t in = (S.vit_in[g][pos >> 2] >> (8 * (pos & 3))) & 3u;
                const uint32_t outbit = (uint32_t) i >> 3;
                const uint32_t p0 = ((uint32_t) i << 1) & 0xEu, p1 = p0 | 1u;
                const uint32_t m0 = (S.vit_metric[cur][g * 16 + p0] + (uint32_t) dh_popc32(in ^ dh_trellis_out(p0, outbit))) & 0xFFu;
                const uint32_t m1 = (S.vit_metric[cur][g * 16 + p1] + (uint32_t) dh_popc32(in ^ dh_trellis_out(p1, outbit))) & 0xFFu;
                const bool both = !NXDN || pos >= 4 || ((uint32_t) i & ((0xFu << pos) & 0xFu)) == 0u;     // trellis.cpp:35-60
                sel = both && (m1 < m0);
                nm = sel ? m1 : m0;
            }
            S.vit_metric[cur ^ 1][lane] = nm;
            DH_BALLOT_ACC(dec, sel, lane);
        }
        S.vit_dec[pos] = dec;
    }
    dh_viterbi_finish(S, sizes, steps & 1);
}
#endif

// note: when sizes differ, a shorter codeword's metrics simply stop updating, so the final metric
// buffer is the right one for every group.

// PN9 whitening sequence (whitening.c:6-22), packed: bits first..first+63, bit (first+j) in bit j
constexpr uint64_t dh_pn9_word(int first) {      // bits first..first+63, bit (first+j) in bit j
    uint32_t wsr = 0x1C9u;
    uint64_t r = 0;
    for (int k = 0; k < first + 64; k++) {
        const uint32_t wb = wsr & 1u;
        const uint32_t fb = ((wsr >> 4) & 1u) ^ wb;
        wsr = ((wsr & 0x1FEu) >> 1) | (fb << 8);
        if (k >= first) r |= (uint64_t) wb << (k - first);
    }
    return r;
}

// AMBE bit order of the V/D type 2 voice channel (ysf_phase.hpp:46-51): voice bit ib -> output bit.
// The table is three arithmetic runs (step 3 up to 39/40/38, then step 2), stated here in closed form.
constexpr int dh_v2_forward(int ib) {
    return ib < 18 ? (ib < 14 ? 3 * ib : 39 + 2 * (ib - 13))
         : ib < 36 ? ((ib - 18) < 14 ? 1 + 3 * (ib - 18) : 40 + 2 * (ib - 18 - 13))
                   : 2 + 3 * (ib - 36);
}
// its inverse: output bit -> voice bit
constexpr int dh_v2_inverse(int ob) {
    return ob <= 40 ? (ob % 3 == 0 ? ob / 3 : ob % 3 == 1 ? 18 + ob / 3 : 36 + ob / 3)
                    : ((ob & 1) ? 14 + (ob - 41) / 2 : 32 + (ob - 42) / 2);
}
constexpr bool dh_v2_check() {
    for (int ib = 0; ib < 49; ib++) if (dh_v2_inverse(dh_v2_forward(ib)) != ib) return false;
    return dh_v2_forward(13) == 39 && dh_v2_forward(17) == 47 && dh_v2_forward(18) == 1 && dh_v2_forward(35) == 48 &&
           dh_v2_forward(36) == 2 && dh_v2_forward(48) == 38;
}
static_assert(dh_v2_check(), "v2 voice mapping closed form");

// decodeV2VoicePayload (ysf_phase.cpp:180-256) of one 52-dibit voice block, lane-local, on the block's two bit masks (dibit j of the
// block in bit j of `hx` / `lx`).  The 26 x 4 de-interleave (:188-197) sends de[26 r + c] = bit (r & 1 ? 0 : 1) of dibit 2 c + (r >> 1):
// the four rows ARE the even / odd positions of the two masks, so nothing is moved -- the whitening sequence is spread the same
// way (a constant), a tribit (:221-239) whose three bits lie in one row is a majority of the mask with itself shifted by two and
// four, the two tribits that straddle rows (voice bits 8 and 17) are taken bit by bit, and the 49 voice bits go straight to their
// place in the AMBE order (ysf_phase.hpp:46-51).  Returns output bytes 0..3 and 4..6 (memory order).
struct DhV2Masks {
    uint64_t pnh, pnl;
    constexpr DhV2Masks(): pnh(0), pnl(0) {
        uint32_t wsr = 0x1C9u;                                    // whitening.c:8
        for (int k = 0; k < 104; k++) {
            const uint64_t wb = wsr & 1u;
            const uint32_t fb = ((wsr >> 4) & 1u) ^ (wsr & 1u);
            wsr = ((wsr & 0x1FEu) >> 1) | (fb << 8);
            const int r = k / 26, c = k % 26, at = 2 * c + (r >> 1);
            if (r & 1) pnl |= wb << at; else pnh |= wb << at;
        }
    }
};
// where de[k] sits: mask (0 = h, 1 = l) and bit
constexpr int dh_v2_de_mask(int k) { return (k / 26) & 1; }
constexpr int dh_v2_de_bit(int k) { return 2 * (k % 26) + ((k / 26) >> 1); }
DH_HD void dh_ysf_v2_block(uint64_t hx, uint64_t lx, uint32_t& out_lo, uint32_t& out_hi) {
    constexpr DhV2Masks W{};
    const uint64_t hw = hx ^ W.pnh, lw = lx ^ W.pnl;
    const uint64_t mh = (hw & (hw >> 2)) | (hw & (hw >> 4)) | ((hw >> 2) & (hw >> 4));
    const uint64_t ml = (lw & (lw >> 2)) | (lw & (lw >> 4)) | ((lw >> 2) & (lw >> 4));
    uint64_t out = 0;                                             // output bit ob (first on the air) in bit ob
#pragma unroll
    for (int ob = 0; ob < 49; ob++) {
        const int ib = dh_v2_inverse(ob);
        uint64_t bit;
        if (ib < 27) {
            const int k = 3 * ib;
            if (k / 26 == (k + 2) / 26) {                        // the tribit lies in one row
                bit = ((dh_v2_de_mask(k) ? ml : mh) >> dh_v2_de_bit(k)) & 1ull;
            } else {
                const uint64_t a = ((dh_v2_de_mask(k) ? lw : hw) >> dh_v2_de_bit(k)) & 1ull;
                const uint64_t b = ((dh_v2_de_mask(k + 1) ? lw : hw) >> dh_v2_de_bit(k + 1)) & 1ull;
                const uint64_t c = ((dh_v2_de_mask(k + 2) ? lw : hw) >> dh_v2_de_bit(k + 2)) & 1ull;
                bit = (a + b + c) >> 1;
            }
        } else {
            const int k = 81 + (ib - 27);
            bit = ((dh_v2_de_mask(k) ? lw : hw) >> dh_v2_de_bit(k)) & 1ull;
        }
        out |= bit << ((ob & ~7) | (7 - (ob & 7)));               // byte ob / 8, first bit on top
    }
    out_lo = (uint32_t) out; out_hi = (uint32_t) (out >> 32);
}

DH_HD bool dh_ysf_is_sync(const DhPlanes& p, int start) {         // ysf_phase.cpp:16-18
    constexpr uint32_t YH = DH_YSF_SYNC_H, YL = DH_YSF_SYNC_L;
    return dh_popc32(dh_plane_range(p.h, start, 20) ^ YH) + dh_popc32(dh_plane_range(p.l, start, 20) ^ YL) <= 3;
}

// ---------------------------------------------------------------------------------------------
// The rate-1/2 codewords of the next frames, decoded ahead (round 5).
//
// Every frame carries two 100-dibit codewords, FICH (fich.cpp:12-23) and -- in V/D mode 2 -- the DCH (ysf_phase.cpp:258-261), and the
// Viterbi forward pass over them was the largest single item of this decoder (1.5 ms of the 7 ms chain launch).  More than half of
// the codewords arrive without a single wrong dibit (the clean-codeword argument above: zero syndrome <=> exactly one path of
// metric 0, whose message falls out of the dibits), but a pass holds four codewords, of two frames, and is only saved when all
// four are clean.  So while the frame grid holds (sync kept: pos + 480 k) the codewords of up to DH_YSF_CHUNK frames are taken
// together: bit planes of the chunk in LDS (16 dibits per lane and load, as in the DMR decoder), then one CODEWORD per lane (a frame's
// FICH and DCH in lanes f and 16 + f) -- out of the planes (the 20 x 5 de-interleave as five nibble spreads per row), syndrome and message with 128-bit shifts,
// lane-locally -- and only the codewords with a non-zero syndrome go through the Viterbi decoder, packed four to a pass whatever
// frames they belong to.  The frame loop finds the decoded bytes in S.ysf.res.
struct DhU128 { uint64_t lo, hi; };
DH_HD DhU128 dh_u128_shl(const DhU128& x, int k) { DhU128 r; r.lo = x.lo << k; r.hi = (x.hi << k) | (x.lo >> (64 - k)); return r; }      // 0 < k < 64
DH_HD DhU128 dh_u128_shr(const DhU128& x, int k) { DhU128 r; r.lo = (x.lo >> k) | (x.hi << (64 - k)); r.hi = x.hi >> k; return r; }
DH_HD DhU128 dh_u128_xor(const DhU128& a, const DhU128& b) { DhU128 r; r.lo = a.lo ^ b.lo; r.hi = a.hi ^ b.hi; return r; }

// `cnt` (<= 32) bits from bit `at` of a plane
DH_HD uint32_t dh_plane_bits(const uint32_t* plane, uint32_t at, uint32_t cnt) {
    const uint32_t w = at >> 5, sh = at & 31u;
    const uint64_t v = (((uint64_t) plane[w + 1u] << 32) | plane[w]) >> sh;
    return (uint32_t) v & (cnt >= 32u ? 0xFFFFFFFFu : ((1u << cnt) - 1u));
}
// 100 bits, bit 20 b + a (b < 5, a < 20) -> bit 5 a + b: row b spread to every fifth position, four bits at a time (a nibble times
// 0x1111 puts bit j at j + 4 m for m = 0..3, of which j + 4 j = 5 j is kept)
DH_HD void dh_transpose_5x20(const uint32_t* x, uint32_t* y) {
    y[0] = y[1] = y[2] = y[3] = 0u;
#pragma unroll
    for (int b = 0; b < 5; b++) {
#pragma unroll
        for (int g = 0; g < 5; g++) {
            const int src = 20 * b + 4 * g, sw = src >> 5, ss = src & 31;
            uint32_t nib = x[sw] >> ss;
            if (ss > 28) nib |= x[sw + 1] << (32 - ss);
            const uint32_t sp = ((nib & 15u) * 0x1111u) & 0x8421u;
            const int dst = 20 * g + b, dw = dst >> 5, ds = dst & 31;
            y[dw] |= sp << ds;
            if (ds > 16) y[dw + 1] |= sp >> (32 - ds);
        }
    }
}
// one 100-dibit codeword given as its two bit streams: true = the full decoder must look at it; otherwise the 100 message bits as the
// 13 bytes trellis.c:55-56,84 writes (first bit in the MSB of byte 0), in four words, and the metric the reference returns.
// Two cases are settled here:
//   * zero syndrome: a codeword (the argument above), metric 0;
//   * ONE wrong dibit at position p, 16 <= p <= 88.  A wrong bit 1 at p shows in the syndrome at p, p + 1, p + 2, p + 4 (h G2), a wrong
//     bit 0 at p, p + 3, p + 4 (l G1), both together at p + 1, p + 2, p + 3 -- a syndrome that is exactly one of these three patterns
//     puts the received word within w = 1 or 2 bits of the codeword c with that dibit flipped back.  The reference returns c: the code
//     is linear with free distance 7, every competitor c' that has merged with c around p costs w + d(c, c') > w, and one that has not
//     -- a detour inside the block (>= 7), a different start state not yet merged at step p (f_start(p + 1) >= 5 for p >= 14), a
//     divergence still open at the block's end (f_end(100 - p) >= 5 for p <= 91) -- costs at least d(c, c') - w >= 5 - 2 > w, so c is
//     the unique best path into every state it passes and the unique best end state, whatever the tie rules (tools/trellis_margin.py
//     computes f_start / f_end; no metric wraps in 100 steps).  Its metric is w.
// tests/test_fec.py runs every position and pattern of single dibits, from every start state, against the reference compiled in place.
#define DH_YSF_REPAIR_LO 16
#define DH_YSF_REPAIR_HI 88
DH_HD bool dh_ysf_clean100(const uint32_t* h, const uint32_t* l, uint32_t* out4, uint32_t* metric = nullptr) {
    DhU128 H, L;
    H.lo = (uint64_t) h[1] << 32 | h[0]; H.hi = (uint64_t) h[3] << 32 | h[2];
    L.lo = (uint64_t) l[1] << 32 | l[0]; L.hi = (uint64_t) l[3] << 32 | l[2];
    // syndrome, checks t = 4 .. 99
    DhU128 syn = dh_u128_xor(dh_u128_xor(dh_u128_xor(H, dh_u128_shl(H, 1)), dh_u128_xor(dh_u128_shl(H, 2), dh_u128_shl(H, 4))),
                             dh_u128_xor(dh_u128_xor(L, dh_u128_shl(L, 3)), dh_u128_shl(L, 4)));
    syn.lo &= ~0xFull; syn.hi &= (1ull << 36) - 1ull;
    bool dirty = (syn.lo | syn.hi) != 0ull;
    uint32_t m = 0;
    if (dirty) {
        const uint32_t t0 = syn.lo ? (uint32_t) dh_ffs64(syn.lo) : 64u + (uint32_t) dh_ffs64(syn.hi);      // the first failing check
        uint64_t five;                                             // checks t0 .. t0 + 4
        if (t0 < 64u) five = (syn.lo >> t0) | (t0 > 59u ? syn.hi << (64u - t0) : 0ull); else five = syn.hi >> (t0 - 64u);
        const uint32_t pat = (uint32_t) five & 31u;
        // nothing beyond those five?
        uint64_t rest_lo = syn.lo, rest_hi = syn.hi;
        if (t0 < 64u) { rest_lo &= ~(31ull << t0); if (t0 > 59u) rest_hi &= ~(31ull >> (64u - t0)); } else rest_hi &= ~(31ull << (t0 - 64u));
        const bool eh = pat == 0x17u || pat == 0x07u, el = pat == 0x19u || pat == 0x07u;
        const uint32_t p = pat == 0x07u ? t0 - 1u : t0;
        if ((rest_lo | rest_hi) == 0ull && (eh || el) && p >= DH_YSF_REPAIR_LO && p <= DH_YSF_REPAIR_HI) {
            const uint64_t bit_lo = p < 64u ? 1ull << p : 0ull, bit_hi = p < 64u ? 0ull : 1ull << (p - 64u);
            if (eh) { H.lo ^= bit_lo; H.hi ^= bit_hi; m++; }
            if (el) { L.lo ^= bit_lo; L.hi ^= bit_hi; m++; }
            dirty = false;
        }
    }
    // message bits 0 .. 97 by the inverse, 98 and 99 from G1 on bits that are then known
    DhU128 U = dh_u128_xor(dh_u128_xor(dh_u128_xor(dh_u128_shr(H, 2), dh_u128_shr(H, 1)), H), dh_u128_xor(dh_u128_shr(L, 2), L));
    U.hi &= (1ull << 34) - 1ull;
    const DhU128 t = dh_u128_xor(H, dh_u128_xor(dh_u128_shl(U, 3), dh_u128_shl(U, 4)));
    U.hi |= t.hi & (3ull << 34);
    const uint32_t u[4] = { (uint32_t) U.lo, (uint32_t) (U.lo >> 32), (uint32_t) U.hi, (uint32_t) (U.hi >> 32) };
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t r = dh_brev32(u[j]);                    // bit i -> 31 - i: the bytes in reverse order, each with its first bit on top
        out4[j] = (r >> 24) | ((r >> 8) & 0xFF00u) | ((r << 8) & 0xFF0000u) | (r << 24);
    }
    if (metric) *metric = m;
    return dirty;
}

// CRC-16 (crc16.c:3-22) of N bytes held in registers, bit by bit (lane-local: a table would be a dependent load per byte)
template <int N> DH_HD uint32_t dh_crc16_regs(const uint32_t* bytes) {
    uint32_t crc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        crc ^= bytes[k] << 8;
#pragma unroll
        for (int i = 0; i < 8; i++) crc = (crc & 0x8000u) ? ((crc << 1) ^ 0x1021u) : (crc << 1);
    }
    return ~crc & 0xFFFFu;
}

// what every frame's codewords are worth, one frame per lane: FICH = 4 x Golay(24,12) + CRC-16 (fich.cpp:24-49), DCH = CRC-16 over
// its first ten bytes (ysf_phase.cpp:262-267)
DH_HD void dh_ysf_check_ahead(const DhDecParams& P, const DhFecTables& T, uint32_t n, DhDecShared& S) {
    DH_FOR_LANES(lane) {
        if ((uint32_t) lane < n) {
            uint32_t w[4], b[12];
#pragma unroll
            for (int j = 0; j < 3; j++) w[j] = S.ysf.res[lane][0][j];
#pragma unroll
            for (int k = 0; k < 12; k++) b[k] = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            uint32_t g[4]; bool fresh = true;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                g[i] = b[3 * i] << 16 | b[3 * i + 1] << 8 | b[3 * i + 2];
                fresh &= dh_block_decode_rows<12>(T.g2412, P.T->lut_g2412, g[i]);
            }
            const uint32_t fich = (g[0] & 0x00FFF000u) << 8 | (g[1] & 0x00FFF000u) >> 4 | (g[2] & 0x00FF0000u) >> 16;
            const uint32_t checksum = (g[2] & 0x0000F000u) | (g[3] & 0x00FFF000u) >> 12;
            const uint32_t be[4] = { fich >> 24, (fich >> 16) & 0xFFu, (fich >> 8) & 0xFFu, fich & 0xFFu };
            fresh = fresh && dh_crc16_regs<4>(be) == checksum;
#pragma unroll
            for (int j = 0; j < 3; j++) w[j] = S.ysf.res[lane][1][j];
#pragma unroll
            for (int k = 0; k < 12; k++) b[k] = (w[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            const bool dch_ok = dh_crc16_regs<10>(b) == (b[10] << 8 | b[11]);
            S.ysf.res[lane][0][0] = fich; S.ysf.res[lane][0][1] = (fresh ? 1u : 0u) | (dch_ok ? 2u : 0u);
        }
    }
    DH_BARRIER();
}

// frames at pos + 480 k, k < n: their FICH and DCH codewords decoded.  S.ysf.res[k][0] = { FICH word (fich.cpp:37-43), bit 0: FICH
// fresh (four Golay(24,12) words corrected and the CRC right, :24-49), bit 1: the DCH's CRC is right (ysf_phase.cpp:262-267) },
// S.ysf.res[k][1] = the 13 DCH bytes
DH_HD void dh_ysf_decode_ahead(const DhDecParams& P, const DhFecTables& T, const DhSymView& syms, uint32_t pos, uint32_t n, uint32_t total, DhDecShared& S) {
    const uint32_t a16 = (uint32_t) ((uintptr_t) syms.fresh & 15u);
    const uint32_t off0 = (pos - syms.nc + a16) & 15u;
    dh_build_chunk_planes(syms, (int32_t) pos - (int32_t) off0, (off0 + 480u * n + 15u) / 16u, total, S.ysf.plane_h, S.ysf.plane_l);
    // lanes 0..15: the FICH of frame `lane`, lanes 16..31: the DCH of frame `lane - 16` -- both are five rows of twenty dibits:
    //   FICH: dibits 20 .. 119 of the frame, codeword dibit i at 20 + 20 (i % 5) + i / 5 (fich.cpp:16-19)
    //   DCH:  the first 20 dibits of each of the five 72-dibit blocks behind dibit 120, dibit i at 120 + 72 (i % 5) + i / 5 (ysf_phase.cpp:103-106)
    uint64_t dirty = 0;
    DH_LANE_VALUE(uint32_t, keep0); DH_LANE_VALUE(uint32_t, keep1); DH_LANE_VALUE(uint32_t, keep2); DH_LANE_VALUE(uint32_t, keep3);      // a dirty codeword's streams: h
    DH_LANE_VALUE(uint32_t, keep4); DH_LANE_VALUE(uint32_t, keep5); DH_LANE_VALUE(uint32_t, keep6); DH_LANE_VALUE(uint32_t, keep7);      // l
    DH_FOR_LANES(lane) {
        const uint32_t kind = ((uint32_t) lane >> 4) & 1u, fl = (uint32_t) lane & 15u;
        const bool mine = (uint32_t) lane < 32u && fl < n;
        const uint32_t f = mine ? fl : 0u;                                    // (lanes beyond the chunk repeat frame 0 and store nothing)
        const uint32_t o = off0 + 480u * f, first = o + (kind ? 120u : 20u), step = kind ? 72u : 20u;
        uint32_t xh[4], xl[4], ch[4], cl[4], out[4];
        {
            uint32_t rh[5], rl[5];
#pragma unroll
            for (int b = 0; b < 5; b++) { rh[b] = dh_plane_bits(S.ysf.plane_h, first + step * (uint32_t) b, 20); rl[b] = dh_plane_bits(S.ysf.plane_l, first + step * (uint32_t) b, 20); }
            xh[0] = rh[0] | rh[1] << 20; xh[1] = rh[1] >> 12 | rh[2] << 8 | rh[3] << 28; xh[2] = rh[3] >> 4 | rh[4] << 16; xh[3] = rh[4] >> 16;
            xl[0] = rl[0] | rl[1] << 20; xl[1] = rl[1] >> 12 | rl[2] << 8 | rl[3] << 28; xl[2] = rl[3] >> 4 | rl[4] << 16; xl[3] = rl[4] >> 16;
        }
        dh_transpose_5x20(xh, ch); dh_transpose_5x20(xl, cl);
        const bool bad = dh_ysf_clean100(ch, cl, out);
        // the frame's sync word (ysf_phase.cpp:16-18): one bit, kept in the top of the DCH entry's fourth word (byte 12 of the DCH is the
        // word's first byte; 13..15 are unused)
        constexpr uint32_t YH = DH_YSF_SYNC_H, YL = DH_YSF_SYNC_L;
        const bool is_sync = dh_popc32(dh_plane_bits(S.ysf.plane_h, o, 20) ^ YH) + dh_popc32(dh_plane_bits(S.ysf.plane_l, o, 20) ^ YL) <= 3;
        if (mine) {
            uint32_t* r = S.ysf.res[f][kind];
            r[0] = out[0]; r[1] = out[1]; r[2] = out[2]; r[3] = kind ? (out[3] & 0xFFu) | (is_sync ? 0x80000000u : 0u) : out[3];
        }
        DH_BALLOT_ACC(dirty, mine && bad, lane);
        DH_LV(keep0, lane) = ch[0]; DH_LV(keep1, lane) = ch[1]; DH_LV(keep2, lane) = ch[2]; DH_LV(keep3, lane) = ch[3];
        DH_LV(keep4, lane) = cl[0]; DH_LV(keep5, lane) = cl[1]; DH_LV(keep6, lane) = cl[2]; DH_LV(keep7, lane) = cl[3];
    }
    // the V/D2 voice blocks (ysf_phase.cpp:180-256) of every frame, one (frame, block) per lane, while the planes are there
    DH_FOR_LANES(lane) {
        for (uint32_t e = (uint32_t) lane; e < 5u * n; e += DH_WAVE) {
            const uint32_t f = e / 5u, b = e - 5u * f, at = off0 + 480u * f + 140u + 72u * b;
            const uint64_t hx = (uint64_t) dh_plane_bits(S.ysf.plane_h, at, 32) | (uint64_t) dh_plane_bits(S.ysf.plane_h, at + 32u, 20) << 32;
            const uint64_t lx = (uint64_t) dh_plane_bits(S.ysf.plane_l, at, 32) | (uint64_t) dh_plane_bits(S.ysf.plane_l, at + 32u, 20) << 32;
            uint32_t v0, v1;
            dh_ysf_v2_block(hx, lx, v0, v1);
            S.ysf.voice[f][2u * b] = v0; S.ysf.voice[f][2u * b + 1u] = v1;
        }
    }
    const uint32_t nd = (uint32_t) dh_popc64(dirty);
    if (nd == 0u) { DH_BARRIER(); dh_ysf_check_ahead(P, T, n, S); return; }
    DH_BARRIER();                                                  // every lane has read the planes: their block takes the dirty streams
    uint8_t* const who = reinterpret_cast<uint8_t*>(S.colword);    // rank -> frame | codeword << 7
    DH_FOR_LANES(lane) {
        if ((dirty >> lane) & 1ull) {
            const uint32_t r = DH_LANES_BELOW(dirty, lane);
            uint32_t* d = S.ysf.dirty[r];
            d[0] = DH_LV(keep0, lane); d[1] = DH_LV(keep1, lane); d[2] = DH_LV(keep2, lane); d[3] = DH_LV(keep3, lane);
            d[4] = DH_LV(keep4, lane); d[5] = DH_LV(keep5, lane); d[6] = DH_LV(keep6, lane); d[7] = DH_LV(keep7, lane);
            who[r] = (uint8_t) (((uint32_t) lane & 15u) | (((uint32_t) lane >> 4) << 7));
        }
    }
    DH_BARRIER();
    for (uint32_t r0 = 0; r0 < nd; r0 += 4u) {
        // four codewords into the decoder's input format: one dibit per byte, first in byte 0 (dh_vit_word)
        DH_FOR_LANES(lane) {
            for (uint32_t e = (uint32_t) lane; e < 100u; e += DH_WAVE) {
                const uint32_t g = e / 25u, w = e - 25u * g;
                uint32_t v = 0;
                if (r0 + g < nd) {
                    const uint32_t* d = S.ysf.dirty[r0 + g];
                    const uint32_t hb = (d[w >> 3] >> (4u * (w & 7u))) & 15u, lb = (d[4u + (w >> 3)] >> (4u * (w & 7u))) & 15u;
                    v = ((hb * 0x00204081u) & 0x01010101u) << 1 | ((lb * 0x00204081u) & 0x01010101u);      // bit q -> byte q
                }
                S.vit_in[g][w] = v;
            }
        }
        DH_BARRIER();
        const int sizes[4] = { 100, r0 + 1u < nd ? 100 : 0, r0 + 2u < nd ? 100 : 0, r0 + 3u < nd ? 100 : 0 };
        dh_viterbi_wave<false, false>(S, sizes);
        DH_FOR_LANES(lane) {
            const uint32_t g = (uint32_t) lane >> 2, j = (uint32_t) lane & 3u;
            if (lane < 16 && r0 + g < nd) {
                const uint32_t id = who[r0 + g];
                const uint32_t w = reinterpret_cast<const uint32_t*>(S.vit_out[g])[j];
                if (j == 3u && (id >> 7)) S.ysf.res[id & 127u][1][3] = (w & 0xFFu) | (S.ysf.res[id & 127u][1][3] & 0x80000000u);      // (the DCH entry's fourth word also carries the sync bit)
                else S.ysf.res[id & 127u][id >> 7][j] = w;
            }
        }
        DH_BARRIER();
    }
    dh_ysf_check_ahead(P, T, n, S);
}

// One YSF channel, one push.
DH_HD void dh_ysf_channel(const DhDecParams& P, uint32_t ch, DhDecShared& S, uint32_t sym_base = 0, bool append = false) {
    const DhFecTables& T = dh_lds_tables(S);
    DhDecCtx c;
    c.P = &P; c.T = &T;
    uint32_t* const st_global = P.state + (size_t) ch * P.state_stride;
    c.out = P.out + (size_t) ch * P.out_stride;
    c.ev = P.events ? P.events + (size_t) ch * P.ev_stride : nullptr;
    c.nout = append ? P.out_count[ch] : 0u; c.nev = append && P.ev_count ? P.ev_count[ch] : 0u; c.overflow = false;      // (append: the second part of a split push, k_chain)
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    c.writer = threadIdx.x == 0;
#else
    c.writer = true;
#endif
    DhState s; s.load(st_global);
    c.st = &s;
    c.consumed = s[DS_CONSUMED];
    uint8_t* const carry_buf = P.carry + (size_t) ch * P.carry_stride;
    DhSymView syms; syms.carry = S.carry; syms.nc = s[DS_CARRY]; syms.fresh = P.syms + (size_t) ch * P.sym_stride + sym_base;
    syms.nfresh = P.sym_count[ch] - sym_base; syms.win = S.symwin; syms.wbase = 0; syms.wlen = 0;
    const uint32_t total = syms.nc + syms.nfresh;
    dh_stage_decoder_lds(P, S, carry_buf, syms.nc);
    uint32_t pos = 0, phase = s[DS_PHASE];
    // frames whose two codewords are already decoded (dh_ysf_decode_ahead): the one at ahead_pos is S.ysf.res[ahead_i], ahead_n in all
    uint32_t ahead_pos = 0xFFFFFFFFu, ahead_i = 0, ahead_n = 0;
    // ... and what the frame loop wants of them, one frame per lane: the FICH word; its flags (bit 0 fresh, bit 1 the DCH's CRC, bit 31 the
    // frame's sync word); the DCH's first three words.  A frame's values are read with v_readlane: fetched from LDS one by one, six round
    // trips per frame were a third of the loop's time.
    DH_LANE_VALUE(uint32_t, fr_fich); DH_LANE_VALUE(uint32_t, fr_flags); DH_LANE_VALUE(uint32_t, fr_d0); DH_LANE_VALUE(uint32_t, fr_d1); DH_LANE_VALUE(uint32_t, fr_d2);
    DH_FOR_LANES(lane) { DH_LV(fr_fich, lane) = 0; DH_LV(fr_flags, lane) = 0; DH_LV(fr_d0, lane) = 0; DH_LV(fr_d1, lane) = 0; DH_LV(fr_d2, lane) = 0; }
    // the frame phase's members (ysf_phase.hpp:36-44) in scalars across the loop; back into the state words behind it
    int sync_count = (int) s[DS_SYNC_COUNT];
    uint32_t running_fich = s[DS_FICH], has_fich = s[DS_HAS_FICH], expect_sub = s[DS_EXPECT_SUB];

    for (;;) {
        const uint32_t avail = total - pos;
        DhPlanes& pl = S.planes;
        if (DH_UNLIKELY(phase == 0)) {                             // SyncPhase (ysf_phase.cpp:20-34)
            if (!(avail > 20)) break;
            dh_view_ensure(syms, pos, 128);
            dh_load_planes(syms, pos, total, pl, 2);
            uint64_t hits = 0;
            DH_FOR_LANES(lane) {
                const bool valid = avail > (uint32_t) lane && avail - (uint32_t) lane > 20;
                DH_BALLOT_ACC(hits, valid && dh_ysf_is_sync(pl, lane), lane);
            }
            if (hits) {
                const uint32_t l = (uint32_t) dh_ffs64(hits);
                pos += l; c.consumed += l; phase = 1; sync_count = 0; has_fich = 0; running_fich = 0; expect_sub = 0;      // FramePhase::FramePhase()
            } else {
                const uint32_t adv = dh_min<uint32_t>(64u, avail - 20u);
                pos += adv; c.consumed += adv;
            }
            continue;
        }
        // FramePhase (ysf_phase.cpp:41-172)
        if (!(avail > 480)) break;
        if (!(ahead_pos == pos && ahead_i < ahead_n)) {
            // the codewords of every frame of the push that is complete (up to DH_YSF_CHUNK), on the grid this frame starts
            const uint32_t n = dh_min<uint32_t>((avail - 1u) / 480u, (uint32_t) DH_YSF_CHUNK);
            dh_ysf_decode_ahead(P, T, syms, pos, n, total, S);
            ahead_pos = pos; ahead_i = 0; ahead_n = n;
            DH_FOR_LANES(lane) {
                const uint32_t f = (uint32_t) lane < n ? (uint32_t) lane : 0u;
                DH_LV(fr_fich, lane) = S.ysf.res[f][0][0]; DH_LV(fr_flags, lane) = (S.ysf.res[f][0][1] & 3u) | (S.ysf.res[f][1][3] & 0x80000000u);
                DH_LV(fr_d0, lane) = S.ysf.res[f][1][0]; DH_LV(fr_d1, lane) = S.ysf.res[f][1][1]; DH_LV(fr_d2, lane) = S.ysf.res[f][1][2];
            }
        }
        // (the frame's bit planes are only built for the payloads that are not decoded ahead: V/D mode 1, voice full rate, header)
        bool have_planes = false;
        auto need_planes = [&]() { if (!have_planes) { dh_view_ensure(syms, pos, 512); dh_load_planes(syms, pos, total, pl, 8); have_planes = true; } };
        const uint32_t cw_flags = DH_LV_READ(fr_flags, ahead_i), cw_fich = DH_LV_READ(fr_fich, ahead_i);
        if (cw_flags >> 31) { if (++sync_count > 12) sync_count = 12; }
        else if (DH_UNLIKELY(--sync_count < 0)) {
            dh_emit(c, DH_EV_YSF_META_RESET, 0, 0, nullptr, 0);
            sync_count = 0; phase = 0; ahead_pos = 0xFFFFFFFFu; continue;
        }
        const uint32_t frame_i = ahead_i;
        ahead_i++; ahead_pos = pos + 480u;

        // FICH (fich.cpp:24-49): decoded and checked ahead
        const uint32_t fich = (cw_flags & 1u) ? cw_fich : 0u; const bool fresh = (cw_flags & 1u) != 0u;
        if (fresh) {
            running_fich = fich; has_fich = 1;
            dh_emit_w(c, DH_EV_YSF_FICH, 0, 0, 4, (fich >> 24) | ((fich >> 8) & 0xFF00u) | ((fich << 8) & 0xFF0000u) | (fich << 24));      // the four bytes, first on the air first
        }

        if (has_fich) {
            const uint32_t rf = running_fich;
            const uint32_t frame_type = (rf >> 30) & 3u, data_type = (rf >> 8) & 3u;
            if (frame_type == 1) {                                                  // communication channel
                dh_emit_w(c, DH_EV_YSF_MODE, 0, data_type, 0);
                if (data_type == 0) {                                               // V/D mode 1 (:73-84)
                    if (P.out_cap - c.nout < 50) c.overflow = true;
                    else {
                        need_planes();
                        uint8_t* o = c.out + c.nout;
                        DH_FOR_LANES(lane) {
                            if (lane < 50) {
                                const int blk = lane / 10, j = lane % 10;
                                // `=` instead of `|=` at ysf_phase.cpp:176: only dibit 4j+3 survives, unshifted
                                o[lane] = j == 0 ? (uint8_t) data_type : (uint8_t) dh_sym_at(pl, 120 + 36 + blk * 72 + 4 * (j - 1) + 3);
                            }
                        }
                        c.nout += 50;
                    }
                } else if (data_type == 2) {                                        // V/D mode 2 (:85-110)
                    if (P.out_cap - c.nout < 40) c.overflow = true;
                    else {
                        uint8_t* o = c.out + c.nout;
                        const uint32_t* vw = S.ysf.voice[frame_i];
                        DH_FOR_LANES(lane) {
                            if (lane < 40) {                            // five blocks of mode byte + seven voice bytes, decoded ahead
                                const uint32_t blk = (uint32_t) lane >> 3, j = (uint32_t) lane & 7u;
                                const uint32_t w = vw[2u * blk + (j > 4u ? 1u : 0u)];
                                o[lane] = j == 0u ? (uint8_t) data_type : (uint8_t) (w >> (8u * ((j - 1u) & 3u)));
                            }
                        }
                        c.nout += 40;
                    }
                    if (fresh) {                                                    // decodeV2DataChannel (:258-269)
                        if (cw_flags & 2u) {                                        // de-whitened (whitening.c:6-22): the first ten bytes, as words
                            constexpr uint32_t PN0 = dh_pn9_le_word(0), PN1 = dh_pn9_le_word(1), PN2 = dh_pn9_le_word(2);
                            dh_emit_w(c, DH_EV_YSF_DCH, (fich >> 19) & 7u, 0, 10, DH_LV_READ(fr_d0, frame_i) ^ PN0, DH_LV_READ(fr_d1, frame_i) ^ PN1, (DH_LV_READ(fr_d2, frame_i) ^ PN2) & 0xFFFFu);
                        }
                    }
                } else if (data_type == 3) {                                        // voice full rate (:111-130)
                    const int start_frame = expect_sub ? 3 : 0;
                    expect_sub = 0;
                    const uint32_t nbytes = (uint32_t) (5 - start_frame) * 19u;
                    if (P.out_cap - c.nout < nbytes) c.overflow = true;
                    else {
                        need_planes();
                        uint8_t* o = c.out + c.nout;
                        DH_FOR_LANES(lane) {
                            for (uint32_t e = lane; e < nbytes; e += DH_WAVE) {
                                const int blk = start_frame + (int) (e / 19u), j = (int) (e % 19u);
                                uint32_t v = data_type;
                                if (j > 0) {
                                    v = 0;
                                    for (int q = 0; q < 4; q++) v = (v << 2) | dh_sym_at(pl, 120 + blk * 72 + (j - 1) * 4 + q);
                                }
                                o[e] = (uint8_t) v;
                            }
                        }
                        c.nout += nbytes;
                    }
                }
            } else if (DH_UNLIKELY(frame_type == 0)) {                              // header (:139-161)
                dh_emit(c, DH_EV_YSF_META_RESET, 0, 1, nullptr, 0);
                // stage 2 (header frames only): CSD1 and CSD2, 180 dibits each (ysf_phase.cpp:323-333)
                need_planes();
                DH_FOR_LANES(lane) {
                    if (lane < 45) {
                        uint32_t a = 0, b = 0;
                        for (int q = 0; q < 4; q++) {
                            const int i = lane * 4 + q;
                            const int streampos = (i % 9) * 20 + i / 9;
                            const int inpos = 120 + (streampos / 36) * 72 + streampos % 36;
                            a = (a << 2) | dh_sym_at(pl, inpos);
                            b = (b << 2) | dh_sym_at(pl, inpos + 36);
                        }
                        S.vit_in[2][lane] = dh_vit_word(a); S.vit_in[3][lane] = dh_vit_word(b);
                    }
                }
                DH_BARRIER();
                {
                    const int sizes2[4] = { 0, 0, 180, 180 };
                    dh_viterbi_wave(S, sizes2);
                }
                for (int half = 0; half < 2; half++) {
                    const uint8_t* w = S.vit_out[2 + half];
                    const uint32_t checksum = (uint32_t) w[20] << 8 | w[21];
                    if (dh_crc16_bytewise(w, 20) == checksum) {
                        uint8_t dch[20];
                        dh_whiten_packed(w, dch, 160);
                        dh_emit(c, DH_EV_YSF_HEADER_DCH, (uint8_t) half, 0, dch, 20);
                    }
                }
                expect_sub = 1;
            } else if (frame_type == 2) {                                           // terminator (:162-164)
                dh_emit(c, DH_EV_YSF_META_RESET, 0, 2, nullptr, 0);
            }
        }
        DH_BARRIER();
        pos += 480; c.consumed += 480;
        if (c.overflow) break;
    }

    const uint32_t rem = total - pos;
    dh_view_ensure(syms, pos, rem < DH_SYM_CARRY_MAX ? rem : DH_SYM_CARRY_MAX);
    DH_FOR_LANES(lane) {
        // sources are LDS (carried part / window), destination is the global carry row: no overlap to worry about
        for (uint32_t j = lane; j < rem && j < DH_SYM_CARRY_MAX; j += DH_WAVE) carry_buf[j] = (uint8_t) dh_view_at(syms, pos + j);
        if (DH_IS_LANE0(lane)) {
            P.out_count[ch] = c.nout;
            if (P.ev_count) P.ev_count[ch] = c.nev;
            if ((c.overflow || rem > DH_SYM_CARRY_MAX) && P.overflow) *P.overflow = 1u;
        }
    }
    s[DS_SYNC_COUNT] = (uint32_t) sync_count; s[DS_FICH] = running_fich; s[DS_HAS_FICH] = has_fich; s[DS_EXPECT_SUB] = expect_sub;
    s[DS_PHASE] = phase; s[DS_CONSUMED] = c.consumed;
    s[DS_CARRY] = rem < DH_SYM_CARRY_MAX ? rem : DH_SYM_CARRY_MAX;
    s.store(st_global);
    DH_BARRIER();
}

// =============================================================================================
// NXDN48 (reference: src/nxdn_decoder/nxdn_phase.cpp, lich.cpp, sacch.cpp,
// facch1.cpp, scrambler.cpp, trellis.cpp)
// =============================================================================================

// the scrambler sequence of a frame is a constant: it restarts after every sync word (nxdn_phase.cpp:56) and runs
// over LICH (8), SACCH (30) and the two 72-dibit blocks; bit j belongs to dibit 10 + j of the frame
struct DhNxdnPn {
    uint64_t w[3];
    constexpr DhNxdnPn(): w() {
        uint32_t sr = 0x0E4u;                                     // scrambler.cpp:9-11
        for (int j = 0; j < 182; j++) {
            const uint32_t wb = sr & 1u;
            w[j >> 6] |= (uint64_t) wb << (j & 63);
            const uint32_t fb = ((sr >> 4) & 1u) ^ wb;            // :21-23
            sr = ((sr & 0x1FEu) >> 1) | (fb << 8);
        }
    }
};
DH_HD uint32_t dh_nxdn_pn(uint32_t j) {
    constexpr DhNxdnPn pn{};
    const uint64_t w = j < 64u ? pn.w[0] : j < 128u ? pn.w[1] : pn.w[2];
    return (uint32_t) (w >> (j & 63u)) & 1u;
}
// descrambled dibit `k` of the frame body (k = 0 is the first LICH dibit, frame offset 10)
DH_HD uint32_t dh_nxdn_dibit(const DhSymView& syms, uint32_t pos, uint32_t k) {
    return (dh_view_at(syms, pos + 10u + k) & 3u) ^ (dh_nxdn_pn(k) << 1);
}
// bit `b` (0 = first) of a run of descrambled dibits starting at body dibit k0: high bit first
DH_HD uint32_t dh_nxdn_bit(const DhSymView& syms, uint32_t pos, uint32_t k0, uint32_t b) {
    return (dh_nxdn_dibit(syms, pos, k0 + (b >> 1)) >> (1u - (b & 1u))) & 1u;
}

constexpr uint32_t DH_NXDN_SYNC_H = dh_sync_plane(0xCDF59ull, 10, 1), DH_NXDN_SYNC_L = dh_sync_plane(0xCDF59ull, 10, 0);   // 3 0 3 1 3 3 1 1 2 1

DH_HD bool dh_nxdn_is_sync_planes(const DhPlanes& p, int start) {   // nxdn_phase.cpp:21: <= 2 bit errors over 10 dibits
    return dh_popc32(dh_plane_range(p.h, start, 10) ^ DH_NXDN_SYNC_H) + dh_popc32(dh_plane_range(p.l, start, 10) ^ DH_NXDN_SYNC_L) <= 2;
}

// sacch.cpp:70-84 / facch1.cpp:63-75 on wave-uniform bytes
DH_HD bool dh_nxdn_crc_ok(const uint8_t* in, int nbits, int width, uint32_t init, uint32_t poly, uint32_t expect) {
    uint32_t crc = init;
    const uint32_t top = 1u << (width - 1), mask = ((1u << width) - 1u) & ~1u;
    for (int i = 0; i < nbits; i++) {
        const uint32_t cb = ((crc & top) ? 1u : 0u) ^ (((uint32_t) in[i >> 3] >> (7 - (i & 7))) & 1u);
        if (cb) crc ^= poly;
        crc = ((crc << 1) & mask) | cb;
    }
    return crc == expect;
}

// One NXDN channel, one push.
DH_HD void dh_nxdn_channel(const DhDecParams& P, uint32_t ch, DhDecShared& S, uint32_t sym_base = 0, bool append = false) {
    DhDecCtx c;
    c.P = &P; c.T = &dh_lds_tables(S);
    uint32_t* const st_global = P.state + (size_t) ch * P.state_stride;
    DhState s; s.load(st_global);
    c.st = &s;
    c.out = P.out + (size_t) ch * P.out_stride;
    c.ev = P.events ? P.events + (size_t) ch * P.ev_stride : nullptr;
    c.nout = append ? P.out_count[ch] : 0u; c.nev = append && P.ev_count ? P.ev_count[ch] : 0u; c.overflow = false;      // (append: the second part of a split push, k_chain)
    c.consumed = s[DS_CONSUMED];
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    c.writer = threadIdx.x == 0;
#else
    c.writer = true;
#endif
    uint8_t* const carry_buf = P.carry + (size_t) ch * P.carry_stride;
    DhSymView syms; syms.carry = S.carry; syms.nc = s[DS_CARRY]; syms.fresh = P.syms + (size_t) ch * P.sym_stride + sym_base;
    syms.nfresh = P.sym_count[ch] - sym_base; syms.win = S.symwin; syms.wbase = 0; syms.wlen = 0;
    const uint32_t total = syms.nc + syms.nfresh;
    dh_stage_decoder_lds(P, S, carry_buf, syms.nc);
    uint32_t pos = 0, phase = s[DS_PHASE];
    uint32_t ahead_pos = 0xFFFFFFFFu, ahead_n = 0;                 // SACCH codewords decoded ahead: frames at ahead_pos + 192 j, j < ahead_n (S.colword)

    for (;;) {
        const uint32_t avail = total - pos;
        if (phase == 0) {                                          // SyncPhase (nxdn_phase.cpp:18-30): slide one dibit at a time
            if (!(avail > 10)) break;
            DhPlanes& pl = S.planes;
            dh_view_ensure(syms, pos, 128);
            dh_load_planes(syms, pos, total, pl, 2);
            uint64_t hits = 0;
            DH_FOR_LANES(lane) {
                const bool valid = avail > (uint32_t) lane && avail - (uint32_t) lane > 10;
                DH_BALLOT_ACC(hits, valid && dh_nxdn_is_sync_planes(pl, lane), lane);
            }
            if (hits) {
                const uint32_t l = (uint32_t) dh_ffs64(hits);
                pos += l; c.consumed += l; phase = 1;
                s[DS_SYNC_COUNT] = 0; s[DS_NX_LICH] = 0; s[DS_NX_HAVE] = 0;      // FramedPhase::FramedPhase()
                for (int i = 0; i < 4; i++) s[DS_NX_SACCH0 + i] = 0;
            } else {
                const uint32_t adv = dh_min<uint32_t>(64u, avail - 10u);
                pos += adv; c.consumed += adv;
            }
            continue;
        }
        // FramedPhase::process (nxdn_phase.cpp:43-170)
        if (!(avail > 192)) break;
        dh_view_ensure(syms, pos, 256);
        // one vote: lanes 0..9 bit 1 of the sync dibits, 10..19 their bit 0, 20..27 the descrambled LICH bits
        uint64_t headvote = 0;
        DH_FOR_LANES(lane) {
            const uint32_t l = (uint32_t) lane;
            bool b = false;
            if (l < 10u) b = ((dh_view_at(syms, pos + l) >> 1) & 1u) != 0;
            else if (l < 20u) b = (dh_view_at(syms, pos + l - 10u) & 1u) != 0;
            else if (l < 28u) b = ((dh_nxdn_dibit(syms, pos, l - 20u) >> 1) & 1u) != 0;       // lich.cpp:10-12
            DH_BALLOT_ACC(headvote, b, lane);
        }
        int sync_count = (int) s[DS_SYNC_COUNT];
        const uint32_t sh = (uint32_t) headvote & 0x3FFu, sl = (uint32_t) (headvote >> 10) & 0x3FFu;
        if (dh_popc32(sh ^ DH_NXDN_SYNC_H) + dh_popc32(sl ^ DH_NXDN_SYNC_L) <= 2) { if (++sync_count > 6) sync_count = 6; }
        else if (DH_UNLIKELY(--sync_count < 0)) {
            dh_emit(c, DH_EV_NXDN_META_RESET, 0, 0, nullptr, 0);
            phase = 0; continue;
        }
        s[DS_SYNC_COUNT] = (uint32_t) sync_count;
        uint32_t used = 18;                                        // sync + LICH

        // LICH: bit i of the vote = lich_bits[i]; parity over the first four (lich.cpp:14-24)
        const uint32_t lb = (uint32_t) (headvote >> 20) & 0xFFu;
        if ((uint32_t) (dh_popc32(lb & 0xFu) & 1) == ((lb >> 7) & 1u)) {
            const uint32_t lich = dh_brev32(lb & 0x7Fu) >> 25;     // lich_bits[0] is the MSB
            s[DS_NX_LICH] = lich + 1u;
            dh_emit_w(c, DH_EV_NXDN_LICH, 0, 0, 1, lich & 0xFFu);
        }
        const uint32_t have_lich = s[DS_NX_LICH];
        const uint32_t lich = have_lich - 1u;
        if (have_lich != 0u && ((lich >> 5) & 3u) != 0u /* RCCH */ && ((lich >> 3) & 3u) != 1u /* UDCH */) {
            const uint32_t option = (lich >> 1) & 3u;
            const bool want_sacch = ((lich >> 3) & 3u) == 2u;      // SACCH superframe
            const bool f0 = ((option >> 1) & 1u) == 0u, f1 = (option & 1u) == 0u;     // block i is a FACCH1
            // Viterbi inputs: codeword 0 = SACCH (60 -> 72 bits), codewords 1 / 2 = the two FACCH1 blocks (144 -> 192 bits);
            // every lane assembles one byte of an inflated codeword straight from the symbols
            // SACCH codewords decoded AHEAD: a voice frame needs one Viterbi pass of 36 steps for its SACCH alone, on the sixteen lanes of
            // one codeword -- the pass costs the same with four codewords in it.  When this frame's SACCH is not at hand, those of the next
            // three frames on the grid (pos + 192 j; the descrambler restarts per frame, the decoder starts from state 0: nothing of it
            // depends on the frames in between) ride along, and their five bytes wait in S.colword until their frame comes -- if it
            // comes there: a sync loss, a TX_RELEASE or a frame without SACCH simply leaves them unused.
            const bool sacch_cached = want_sacch && pos >= ahead_pos && pos - ahead_pos < 192u * ahead_n && (pos - ahead_pos) % 192u == 0u;
            uint32_t sacch_slot = sacch_cached ? (pos - ahead_pos) / 192u : 0u;
            if (want_sacch && !sacch_cached) {
                const uint32_t nahead = (f0 || f1) ? 1u : dh_min<uint32_t>(4u, (avail - 1u) / 192u);     // (with a FACCH1 in the pass the other codeword groups are taken)
                if (nahead > 1u) dh_view_ensure(syms, pos, 192u * nahead);
                DH_FOR_LANES(lane) {
                    const int j = lane / 9, l = lane - 9 * j;
                    if ((uint32_t) j < nahead) {                    // sacch.cpp:45-68
                        uint32_t v = 0;
                        for (int t = 0; t < 8; t++) {
                            const int i = l * 8 + t;
                            uint32_t x = 0;
                            if ((i + 1) % 6 != 0) {
                                const int o = i - (i + 1) / 6;          // position among the 60 transmitted bits
                                const int inpos = (o % 12) * 5 + o / 12;
                                x = dh_nxdn_bit(syms, pos + 192u * (uint32_t) j, 8u, (uint32_t) inpos);
                            }
                            v = (v << 1) | x;
                        }
                        S.vit_in[j][l] = dh_vit_word(v);
                    }
                }
                if (nahead > 1u) {
                    DH_BARRIER();
                    const int sizes[4] = { 36, 36, nahead > 2u ? 36 : 0, nahead > 3u ? 36 : 0 };
                    dh_viterbi_wave<true, false>(S, sizes);
                    DH_FOR_LANES(lane) {
                        // lanes 0 .. 3: the five bytes of one frame's SACCH and its CRC-6 (sacch.cpp:70-84), checked here, one frame per lane
                        // (bit 31 of the second word says "CRC right"; bytes 5 .. 7 of an entry are not used)
                        if (lane < 4) {
                            const uint32_t* w = reinterpret_cast<const uint32_t*>(S.vit_out[lane]);
                            const uint32_t w0 = (uint32_t) lane < nahead ? w[0] : 0u, w1 = (uint32_t) lane < nahead ? w[1] & 0xFFu : 0u;
                            const uint8_t by[5] = { (uint8_t) w0, (uint8_t) (w0 >> 8), (uint8_t) (w0 >> 16), (uint8_t) (w0 >> 24), (uint8_t) w1 };
                            const bool ok = dh_nxdn_crc_ok(by, 26, 6, 0x3Fu, 0x13u, by[3] & 0x3Fu);
                            S.colword[2 * lane] = w0; S.colword[2 * lane + 1] = w1 | (ok ? 0x80000000u : 0u);
                        }
                    }
                    DH_BARRIER();
                    ahead_pos = pos; ahead_n = nahead; sacch_slot = 0;
                }
            }
            const bool sacch_from_cache = want_sacch && (sacch_cached || !(f0 || f1) && (avail - 1u) / 192u > 1u);
            if (f0 || f1) DH_FOR_LANES(lane) {
                const int l = lane;
                if (l >= 16 && l < 64) {                            // facch1.cpp:39-61
                    const int blk = (l - 16) / 24, by = (l - 16) % 24;
                    uint32_t v = 0;
                    for (int t = 0; t < 8; t++) {
                        const int i = by * 8 + t;
                        uint32_t x = 0;
                        if ((i - 1) % 4 != 0) {
                            const int o = i - (i + 2) / 4;          // bits 1, 5, 9, ... are punctured
                            const int inpos = (o % 16) * 9 + o / 16;
                            x = dh_nxdn_bit(syms, pos, 38u + 72u * (uint32_t) blk, (uint32_t) inpos);
                        }
                        v = (v << 1) | x;
                    }
                    S.vit_in[1 + blk][by] = dh_vit_word(v);
                }
            }
            DH_BARRIER();
            {
                const bool sacch_here = want_sacch && !sacch_from_cache;
                const int sizes[4] = { sacch_here ? 36 : 0, f0 ? 96 : 0, f1 ? 96 : 0, 0 };
                if (f0 || f1) dh_viterbi_wave<true, true>(S, sizes);           // 36 / 96 / 96 steps side by side
                else if (sacch_here) dh_viterbi_wave<true, false>(S, sizes);   // (the last frame of a push: its SACCH alone)
            }
            if (want_sacch) {
                const uint8_t* w = sacch_from_cache ? reinterpret_cast<const uint8_t*>(S.colword + 2u * sacch_slot) : S.vit_out[0];
                uint8_t sacch[5];
                for (int i = 0; i < 5; i++) sacch[i] = (uint8_t) dh_uniform(w[i]);
                const bool sacch_ok = sacch_from_cache ? (dh_uniform(S.colword[2u * sacch_slot + 1u]) >> 31) != 0u
                                                       : dh_nxdn_crc_ok(sacch, 26, 6, 0x3Fu, 0x13u, sacch[3] & 0x3Fu);
                if (sacch_ok) {
                    const uint32_t index = ((uint32_t) sacch[0] >> 6) ^ 3u;
                    dh_emit_w(c, DH_EV_NXDN_SACCH, index, 0, 5, (uint32_t) sacch[0] | (uint32_t) sacch[1] << 8 | (uint32_t) sacch[2] << 16 | (uint32_t) sacch[3] << 24, sacch[4]);
                    uint32_t have = s[DS_NX_HAVE];
                    if (!(index > 0u && !((have >> (index - 1u)) & 1u))) {        // SacchSuperframeCollector::push (sacch.cpp:90-98)
                        have |= 1u << index;
                        s[DS_NX_SACCH0 + index] = (uint32_t) sacch[1] << 24 | (uint32_t) sacch[2] << 16 | (uint32_t) sacch[3] << 8 | sacch[4];
                    }
                    if (have == 0xFu) {                                           // getSuperframe (:116-131): 4 x 18 bits
                        uint8_t sf[9];
                        uint64_t acc = 0; int nacc = 0, ob = 0;
                        for (int i = 0; i < 4; i++) {
                            acc = (acc << 18) | ((uint32_t) s[DS_NX_SACCH0 + i] >> 14); nacc += 18;
                            while (nacc >= 8) { sf[ob++] = (uint8_t) (acc >> (nacc - 8)); nacc -= 8; }
                        }
                        dh_emit(c, DH_EV_NXDN_SACCH_SF, 0, 0, sf, 9);
                        have = 0;
                    }
                    s[DS_NX_HAVE] = have;
                }
            }
            used += 30;
            bool released = false;
            if (option == 3u && sync_count >= 1 && P.out_cap - c.nout >= 36u) {
                // both blocks are voice (the usual frame of a call): their 2 x 18 bytes in ONE pass of 36 lanes (nxdn_phase.cpp:136-150, twice)
                dh_emit_w(c, DH_EV_NXDN_SYNC_VOICE, 0, 0, 0);
                dh_emit_w(c, DH_EV_NXDN_SYNC_VOICE, 0, 0, 0);
                uint8_t* o = c.out + c.nout;
                DH_FOR_LANES(lane) {
                    if (lane < 36) {
                        const uint32_t blk = lane >= 18 ? 1u : 0u, j = (uint32_t) lane - 18u * blk;
                        uint32_t v = 0;
                        for (int q = 0; q < 4; q++) v = (v << 2) | dh_nxdn_dibit(syms, pos, 38u + 72u * blk + 4u * j + (uint32_t) q);
                        o[lane] = (uint8_t) v;
                    }
                }
                c.nout += 36;
                used += 144;
            } else
            for (int i = 0; i < 2 && !released; i++) {
                if ((option >> (1 - i)) & 1u) {                                   // voice (nxdn_phase.cpp:136-150)
                    if (sync_count >= 1) {
                        dh_emit_w(c, DH_EV_NXDN_SYNC_VOICE, 0, 0, 0);
                        if (P.out_cap - c.nout < 18) { c.overflow = true; break; }
                        uint8_t* o = c.out + c.nout;
                        DH_FOR_LANES(lane) {
                            if (lane < 18) {
                                uint32_t v = 0;
                                for (int q = 0; q < 4; q++) v = (v << 2) | dh_nxdn_dibit(syms, pos, 38u + 72u * (uint32_t) i + 4u * (uint32_t) lane + (uint32_t) q);
                                o[lane] = (uint8_t) v;
                            }
                        }
                        c.nout += 18;
                    }
                } else {                                                          // FACCH1 (:151-165)
                    const uint8_t* w = S.vit_out[1 + i];
                    uint8_t f[12];
                    for (int k = 0; k < 12; k++) f[k] = (uint8_t) dh_uniform(w[k]);
                    if (dh_nxdn_crc_ok(f, 80, 12, 0xFFFu, 0x407u, (uint32_t) f[10] << 4 | (uint32_t) f[11] >> 4)) {
                        dh_emit(c, DH_EV_NXDN_FACCH1, (uint8_t) i, 0, f, 12);
                        if ((f[0] & 0x3Fu) == 0x08u) {                            // TX_RELEASE: back to SyncPhase, block not consumed
                            dh_emit(c, DH_EV_NXDN_META_RESET, 0, 1, nullptr, 0);
                            released = true;
                            break;
                        }
                    }
                }
                used += 72;
            }
            DH_BARRIER();
            if (released) phase = 0;
        } else {
            used += 174;
        }
        pos += used; c.consumed += used;
        if (c.overflow) break;
    }

    const uint32_t rem = total - pos;
    dh_view_ensure(syms, pos, rem < DH_SYM_CARRY_MAX ? rem : DH_SYM_CARRY_MAX);
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < rem && j < DH_SYM_CARRY_MAX; j += DH_WAVE) carry_buf[j] = (uint8_t) dh_view_at(syms, pos + j);
        if (DH_IS_LANE0(lane)) {
            P.out_count[ch] = c.nout;
            if (P.ev_count) P.ev_count[ch] = c.nev;
            if ((c.overflow || rem > DH_SYM_CARRY_MAX) && P.overflow) *P.overflow = 1u;
        }
    }
    s[DS_PHASE] = phase; s[DS_CONSUMED] = c.consumed;
    s[DS_CARRY] = rem < DH_SYM_CARRY_MAX ? rem : DH_SYM_CARRY_MAX;
    s.store(st_global);
    DH_BARRIER();
}

// =============================================================================================
// POCSAG (reference: src/pocsag_decoder/pocsag_phase.cpp, codeword.cpp, message.cpp, bch_31_21.c;
// examples/pocsag-decoder.sh: fsk_demodulator -i -s 40 | pocsag_decoder).  Input symbols are bits.
// =============================================================================================
#define DH_POCSAG_SYNC 0x7CD215D8u          // pocsag_phase.hpp:15, first bit in the MSB
#define DH_POCSAG_IDLE 0x7A89C197u          // codeword.hpp:23

// the 32 symbols at `pos`, one per lane: `word` has symbol 0 in its MSB with any non-zero symbol as 1 (codeword.cpp:12:
// `input[i] && 1`); `dist` is hamming_distance() against the sync word (bit 0 differs, or bit 1 is set)
DH_HD void dh_pocsag_take32(const DhSymView& syms, uint32_t pos, uint32_t& word, int& dist) {
    uint64_t nz = 0, b0 = 0, b1 = 0;
    DH_FOR_LANES(lane) {
        uint32_t v = 0;
        if (lane < 32) v = dh_view_at(syms, pos + (uint32_t) lane);
        DH_BALLOT_ACC(nz, v != 0u, lane);
        DH_BALLOT_ACC(b0, (v & 1u) != 0u, lane);
        DH_BALLOT_ACC(b1, (v & 2u) != 0u, lane);
    }
    word = dh_brev32((uint32_t) nz);
    dist = dh_popc32(dh_brev32((uint32_t) b0) ^ DH_POCSAG_SYNC) + dh_popc32((uint32_t) b1);
}

// Message::serialize with the StringSerializer (message.cpp:17-25, meta.cpp:8-17): `address:<n>;message:<text>\n`
DH_HD void dh_pocsag_serialize(DhDecCtx& c, DhState& s, DhDecShared& S) {
    if (!s[DS_PC_HAS] || (uint32_t) s[DS_PC_POS] == 0u) return;
    uint8_t* line = reinterpret_cast<uint8_t*>(&S.vit_in[0][0]);     // 768 bytes of scratch, unused by this protocol
    uint32_t n = 0;
    const char* a = "address:";
    for (int i = 0; i < 8; i++) line[n++] = (uint8_t) a[i];
    char num[12]; int nn = 0; uint32_t v = s[DS_PC_ADDR];
    do { num[nn++] = (char) ('0' + v % 10u); v /= 10u; } while (v);
    while (nn) line[n++] = (uint8_t) num[--nn];
    const char* m = ";message:";
    for (int i = 0; i < 9; i++) line[n++] = (uint8_t) m[i];
    for (uint32_t i = 0; i < 80u; i++) {
        const uint32_t ch = ((uint32_t) s[DS_PC_CONTENT + (i >> 2)] >> (8u * (i & 3u))) & 0xFFu;
        if (ch == 0u) break;                              // std::string(content) ends at the first NUL
        line[n++] = (uint8_t) ch;
    }
    line[n++] = (uint8_t) '\n';
    if (c.P->out_cap - c.nout < n) { c.overflow = true; return; }
    DH_BARRIER();
    uint8_t* o = c.out + c.nout;
    DH_FOR_LANES(lane) { for (uint32_t j = (uint32_t) lane; j < n; j += DH_WAVE) o[j] = line[j]; }
    DH_BARRIER();
    c.nout += n;
}

// Message::append (message.cpp:27-72)
DH_HD void dh_pocsag_append(DhState& s, uint32_t data) {
    const uint32_t type = s[DS_PC_TYPE];
    uint32_t pos = s[DS_PC_POS];
    if (type == 3u) {
        if (pos + 20u < 80u * 7u) {
            for (int i = 0; i < 20; i++) {
                const uint32_t bit = (data >> (19 - i)) & 1u, ch = pos / 7u;
                s[DS_PC_CONTENT + (ch >> 2)] = (uint32_t) s[DS_PC_CONTENT + (ch >> 2)] | (bit << (pos % 7u)) << (8u * (ch & 3u));
                pos++;
            }
        }
    } else if (type == 0u) {
        if (pos + 5u < 80u) {
            for (int i = 0; i < 5; i++) {
                uint32_t ch = 0;
                const uint32_t base = (uint32_t) (4 - i) * 4u;
                for (int k = 0; k < 4; k++) ch |= ((data >> (base + (uint32_t) k)) & 1u) << (3 - k);
                const char tail[6] = { '*', 'U', ' ', '-', ')', '(' };
                ch = ch < 0xAu ? '0' + ch : (uint32_t) tail[ch - 0xAu];
                s[DS_PC_CONTENT + (pos >> 2)] = (uint32_t) s[DS_PC_CONTENT + (pos >> 2)] | ch << (8u * (pos & 3u));
                pos++;
            }
        }
    }
    s[DS_PC_POS] = pos;
}

DH_HD void dh_pocsag_channel(const DhDecParams& P, uint32_t ch, DhDecShared& S) {
    DhDecCtx c;
    c.P = &P; c.T = &dh_lds_tables(S);
    uint32_t* const st_global = P.state + (size_t) ch * P.state_stride;
    DhState s; s.load(st_global);
    c.st = &s;
    c.out = P.out + (size_t) ch * P.out_stride;
    c.ev = P.events ? P.events + (size_t) ch * P.ev_stride : nullptr;
    c.nout = 0; c.nev = 0; c.overflow = false;
    c.consumed = s[DS_CONSUMED];
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    c.writer = threadIdx.x == 0;
#else
    c.writer = true;
#endif
    uint8_t* const carry_buf = P.carry + (size_t) ch * P.carry_stride;
    DhSymView syms; syms.carry = S.carry; syms.nc = s[DS_CARRY]; syms.fresh = P.syms + (size_t) ch * P.sym_stride;
    syms.nfresh = P.sym_count[ch]; syms.win = S.symwin; syms.wbase = 0; syms.wlen = 0;
    const uint32_t total = syms.nc + syms.nfresh;
    dh_stage_decoder_lds(P, S, carry_buf, syms.nc);
    uint32_t pos = 0, phase = s[DS_PHASE];

    for (;;) {
        const uint32_t avail = total - pos;
        if (!(avail > 32)) break;                                  // both phases need 32 bits (pocsag_phase.cpp:14,34)
        if (phase == 0) {                                          // SyncPhase (:18-28): slide bit by bit
            DhPlanes& pl = S.planes;
            dh_view_ensure(syms, pos, 128);
            dh_load_planes(syms, pos, total, pl, 2);
            uint64_t hits = 0;
            DH_FOR_LANES(lane) {
                const bool valid = avail > (uint32_t) lane && avail - (uint32_t) lane > 32;
                const uint32_t l = dh_plane_range(pl.l, lane, 32), h = dh_plane_range(pl.h, lane, 32);
                DH_BALLOT_ACC(hits, valid && dh_popc32(dh_brev32(l) ^ DH_POCSAG_SYNC) + dh_popc32(h) <= 3, lane);
            }
            if (hits) {
                const uint32_t l = (uint32_t) dh_ffs64(hits);
                pos += l + 32u; c.consumed += l + 32u; phase = 1;
                s[DS_SYNC_COUNT] = 1; s[DS_PC_COUNTER] = 0; s[DS_PC_HAS] = 0;         // CodewordPhase members (pocsag_phase.hpp:31-34)
            } else {
                const uint32_t adv = dh_min<uint32_t>(64u, avail - 32u);
                pos += adv; c.consumed += adv;
            }
            continue;
        }
        // CodewordPhase::process (pocsag_phase.cpp:38-92)
        dh_view_ensure(syms, pos, 64);
        uint32_t word; int dist;
        dh_pocsag_take32(syms, pos, word, dist);
        uint32_t counter = s[DS_PC_COUNTER];
        if (counter >= 16u) {
            int sync_count = (int) s[DS_SYNC_COUNT];
            if (dist <= 3) { if (sync_count++ > 2) sync_count = 2; }
            else if (sync_count-- < 0) {
                dh_pocsag_serialize(c, s, S);
                s[DS_PC_HAS] = 0;
                phase = 0;
                continue;
            }
            s[DS_SYNC_COUNT] = (uint32_t) sync_count;
            s[DS_PC_COUNTER] = 0;
        } else {
            // Codeword::parse (codeword.cpp:9-32)
            uint32_t payload = word >> 1;
            bool ok = dh_block_decode_wave<10>(c.T->bch3121, P.T->lut_bch3121, payload);
            const uint32_t cw = (word & 1u) | (payload << 1);
            if (ok && (dh_popc32(cw) & 1)) ok = false;
            if (ok) {
                const uint8_t be[4] = { (uint8_t) (cw >> 24), (uint8_t) (cw >> 16), (uint8_t) (cw >> 8), (uint8_t) cw };
                dh_emit(c, DH_EV_POCSAG_CODEWORD, (uint8_t) counter, 0, be, 4);
                if (cw == DH_POCSAG_IDLE) {
                    dh_pocsag_serialize(c, s, S);
                    s[DS_PC_HAS] = 0;
                } else if ((cw >> 31) == 0u) {                     // address codeword
                    dh_pocsag_serialize(c, s, S);
                    s[DS_PC_HAS] = 0;
                    const uint32_t type = (cw >> 11) & 3u;
                    if (type == 1u || type == 3u) {
                        s[DS_PC_HAS] = 1;
                        s[DS_PC_ADDR] = (((cw >> 13) & 0x3FFFFu) << 3) | (counter / 2u);
                        s[DS_PC_TYPE] = type; s[DS_PC_POS] = 0;
                        for (int i = 0; i < 20; i++) s[DS_PC_CONTENT + i] = 0;
                    }
                } else if (s[DS_PC_HAS]) {
                    dh_pocsag_append(s, (cw >> 11) & 0xFFFFFu);
                }
            } else {
                s[DS_PC_HAS] = 0;
            }
            s[DS_PC_COUNTER] = counter + 1u;
        }
        pos += 32u; c.consumed += 32u;
        if (c.overflow) break;
    }

    const uint32_t rem = total - pos;
    dh_view_ensure(syms, pos, rem < DH_SYM_CARRY_MAX ? rem : DH_SYM_CARRY_MAX);
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < rem && j < DH_SYM_CARRY_MAX; j += DH_WAVE) carry_buf[j] = (uint8_t) dh_view_at(syms, pos + j);
        if (DH_IS_LANE0(lane)) {
            P.out_count[ch] = c.nout;
            if (P.ev_count) P.ev_count[ch] = c.nev;
            if ((c.overflow || rem > DH_SYM_CARRY_MAX) && P.overflow) *P.overflow = 1u;
        }
    }
    s[DS_PHASE] = phase; s[DS_CONSUMED] = c.consumed;
    s[DS_CARRY] = rem < DH_SYM_CARRY_MAX ? rem : DH_SYM_CARRY_MAX;
    s.store(st_global);
    DH_BARRIER();
}

// =============================================================================================
// D-Star (reference: src/dstar_decoder/dstar_phase.cpp, header.cpp, scrambler.cpp, crc.cpp;
// examples/dstar-decoder.sh: fsk_demodulator -s 10 | dstar_decoder).  Input symbols are bits; the output is the 9-byte
// AMBE voice frames.  A voice frame (96 bits) is two plane words and a handful of scalar integer operations; the radio
// header's 4-state Viterbi runs on the SCALAR unit (its inputs go through readfirstlane), leaving the vector ALUs to
// the other wavefronts of the SIMD.
// =============================================================================================
constexpr uint32_t dh_bits_lsb(const char* bits, int from, int cnt) {      // "0110.." -> character from+i in bit i
    uint32_t r = 0;
    for (int i = 0; i < cnt; i++) r |= (uint32_t) (bits[from + i] == '1') << i;
    return r;
}
// dstar_phase.hpp:18-41
#define DH_DSTAR_HEADER_SYNC dh_bits_lsb("010101010111011001010000", 0, 24)
#define DH_DSTAR_VOICE_SYNC  dh_bits_lsb("101010101011010001101000", 0, 24)
#define DH_DSTAR_TERM_LO     dh_bits_lsb("101010101010101010101010101010100001001101011110", 0, 24)
#define DH_DSTAR_TERM_HI     dh_bits_lsb("101010101010101010101010101010100001001101011110", 24, 24)

// the scrambler's whitening sequence from its reset state (scrambler.cpp:7-21, scrambler.hpp:13): bit n in w[n / 32]
struct DhDstarPn {
    uint32_t w[21];
    constexpr DhDstarPn(): w() {
        uint32_t sr = 0x7Fu;
        for (int n = 0; n < 660; n++) {
            const uint32_t wb = (sr & 1u) ^ ((sr >> 3) & 1u);
            w[n >> 5] |= wb << (n & 31);
            sr = ((sr & 0x7Eu) >> 1) | (wb << 6);
        }
    }
};
DH_HD uint32_t dh_dstar_pn(uint32_t n) {
    constexpr DhDstarPn pn{};
    return (pn.w[n >> 5] >> (n & 31u)) & 1u;
}

// Crc::isCrcValid's checksum (crc.cpp:6-23) over `len` bytes delivered by get(i)
template <typename Get> DH_HD uint32_t dh_dstar_crc(Get get, uint32_t len) {
    uint32_t checksum = 0xFFFFu;
    for (uint32_t k = 0; k < len; k++) {
        const uint32_t byte = get(k);
        for (int i = 0; i < 8; i++) {
            checksum ^= (byte >> i) & 1u;
            checksum = (checksum & 1u) ? (checksum >> 1) ^ 0x8408u : checksum >> 1;
        }
    }
    return checksum ^ 0xFFFFu;
}

// 128 symbols at `pos` as bit masks held in scalar registers (symbol i of the run in bit i): bit 0 of every symbol in
// l[], bit 1 in h[] (hamming_distance() counts a set bit 1 as a difference against the 0/1 patterns)
struct DhBits128 { uint64_t l[2], h[2]; };
DH_HD void dh_dstar_take128(const DhSymView& syms, uint32_t pos, uint32_t total, DhBits128& b) {
    for (int w = 0; w < 2; w++) {
        uint64_t ml = 0, mh = 0;
        DH_FOR_LANES(lane) {
            const uint32_t j = pos + (uint32_t) (w * 64 + lane);
            const uint32_t v = j < total ? dh_view_at(syms, j) : 0u;
            DH_BALLOT_ACC(ml, v & 1u, lane);
            DH_BALLOT_ACC(mh, (v >> 1) & 1u, lane);
        }
        b.l[w] = ml; b.h[w] = mh;
    }
}
// cnt (<= 32) bits starting at bit `start` (< 64 + cnt) of a 128-bit mask
DH_HD uint32_t dh_bits_range(const uint64_t* w, uint32_t start, uint32_t cnt) {
    uint64_t v = start < 64u ? (w[0] >> start) | (start ? w[1] << (64u - start) : 0ull) : w[1] >> (start - 64u);
    return (uint32_t) v & (cnt >= 32u ? 0xFFFFFFFFu : (1u << cnt) - 1u);
}

// LDS scratch of the header decoder, carved from S.carry (D-Star keeps its carried bits in S.vit_dec instead)
struct DhDstarScratch { uint64_t dbits[11]; uint32_t dec[44]; uint32_t out[12]; };
static_assert(sizeof(DhDstarScratch) <= DH_SYM_CARRY_MAX, "D-Star header scratch");

// Header::parseFromHeader (header.cpp:24-58) on the 660 bits at `pos`: true when the path metric is <= 10 and the CRC
// holds; the 41 header bytes are then in X.out (little-endian words)
DH_HD bool dh_dstar_header_parse(const DhSymView& syms, uint32_t pos, DhDstarScratch& X) {
    // descramble + de-interleave (header.cpp:26-30, :60-72): de-interleaved bit idx = 24 k + i comes from received
    // bit i * 28 + k (i < 12) or 12 + i * 27 + k; 64 bits per vote
    for (uint32_t c = 0; c < 11u; c++) {
        uint64_t m = 0;
        DH_FOR_LANES(lane) {
            const uint32_t idx = c * 64u + (uint32_t) lane;
            uint32_t bit = 0;
            if (idx < 660u) {
                const uint32_t k = idx / 24u, i = idx % 24u;
                const uint32_t src = i < 12u ? i * 28u + k : 12u + i * 27u + k;
                bit = (dh_view_at(syms, pos + src) & 1u) ^ dh_dstar_pn(src);
            }
            DH_BALLOT_ACC(m, bit != 0u, lane);
        }
        X.dbits[c] = m;
    }
    DH_BARRIER();
    // Header::viterbi_decode (:86-150) as add-compare-select + decision bits + traceback: the reference carries the
    // four survivor byte strings instead, which is the same path under the same tie rules (k = 0 wins ties, the
    // lowest end state wins).  State i = (newest bit << 1) | previous bit; predecessor ((i << 1) & 2) | k.
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    {
        // the four states on the four lanes of every quad: lane i reads the metrics of its two predecessors through
        // DPP quad permutes ([0,2,0,2] and [1,3,1,3]), its branch words are 00 / 10 / 11 / 01 for k = 0 and the
        // complement for k = 1; sixteen 2-bit branch distances are computed at once per 32 received bits
        const uint32_t st = threadIdx.x & 3u;
        const uint32_t ca = (st == 1u || st == 2u) ? 0x55555555u : 0u;      // first bit of the k = 0 branch word
        const uint32_t cb = st >= 2u ? 0x55555555u : 0u;                     // second bit
        uint32_t m = 0;
#define DH_DSTAR_ACS(q) do { \
            const uint32_t a_ = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) m, 0x88, 0xF, 0xF, true) + ((ha >> (2u * (q))) & 3u); \
            const uint32_t b_ = (uint32_t) __builtin_amdgcn_update_dpp(0, (int) m, 0xDD, 0xF, 0xF, true) + ((hb >> (2u * (q))) & 3u); \
            dw |= ((uint32_t) __ballot(b_ < a_) & 0xFu) << (4u * (q));      /* lanes 0..3 = new states 0..3 */ \
            m = b_ < a_ ? b_ : a_; } while (0)
        for (uint32_t g = 0; g < 42u; g++) {                                 // 8 trellis steps = 16 received bits per round
            const uint32_t seg = (uint32_t) (X.dbits[g >> 2] >> (16u * (g & 3u))) & 0xFFFFu;
            const uint32_t ha = ((seg & 0x5555u) ^ (ca & 0x5555u)) + (((seg >> 1) & 0x5555u) ^ (cb & 0x5555u));
            const uint32_t hb = 0xAAAAu - ha;
            uint32_t dw = 0;
            if (g < 41u) {
                DH_DSTAR_ACS(0); DH_DSTAR_ACS(1); DH_DSTAR_ACS(2); DH_DSTAR_ACS(3);
                DH_DSTAR_ACS(4); DH_DSTAR_ACS(5); DH_DSTAR_ACS(6); DH_DSTAR_ACS(7);
            } else { DH_DSTAR_ACS(0); DH_DSTAR_ACS(1); }                     // steps 328, 329
            X.dec[g] = dw;
        }
#undef DH_DSTAR_ACS
        m0 = (uint32_t) __builtin_amdgcn_readlane((int) m, 0); m1 = (uint32_t) __builtin_amdgcn_readlane((int) m, 1);
        m2 = (uint32_t) __builtin_amdgcn_readlane((int) m, 2); m3 = (uint32_t) __builtin_amdgcn_readlane((int) m, 3);
    }
#else
    for (uint32_t blk = 0; blk < 11u; blk++) {                     // 32 trellis steps per 64-bit word
        const uint64_t word = X.dbits[blk];
        uint32_t lo = dh_uniform((uint32_t) word), hi = dh_uniform((uint32_t) (word >> 32));
        const uint32_t steps = blk < 10u ? 32u : 10u;
        for (uint32_t g = 0; g < 4u; g++) {
          uint32_t dw = 0;
          const uint32_t half = g < 2u ? lo : hi;
          for (uint32_t qq = 0; qq < 8u; qq++) {
            const uint32_t q = g * 8u + qq;
            if (q >= steps) break;
            const uint32_t pair = (half >> (2u * (q & 15u))) & 3u;
            const uint32_t t = ((pair & 1u) << 1) | (pair >> 1);  // in_transition: first bit is the high one (:95)
            // hamming distance of t against the branch words 00 / 11 / 10 / 01 (trellis_transitions, :79-84)
            const uint32_t h00 = (t & 1u) + (t >> 1), h11 = 2u - h00, h10 = (t & 1u) + (1u - (t >> 1)), h01 = 2u - h10;
            // new state 0: from 0 (out 00) or 1 (out 11); 1: from 2 (10) or 3 (01); 2: from 0 (11) or 1 (00); 3: from 2 (01) or 3 (10)
            const uint32_t a0 = m0 + h00, b0 = m1 + h11, a1 = m2 + h10, b1 = m3 + h01;
            const uint32_t a2 = m0 + h11, b2 = m1 + h00, a3 = m2 + h01, b3 = m3 + h10;
            const uint32_t s0 = b0 < a0, s1 = b1 < a1, s2 = b2 < a2, s3 = b3 < a3;
            m0 = s0 ? b0 : a0; m1 = s1 ? b1 : a1; m2 = s2 ? b2 : a2; m3 = s3 ? b3 : a3;
            dw |= (s0 | (s1 << 1) | (s2 << 2) | (s3 << 3)) << (4u * qq);
          }
          X.dec[blk * 4u + g] = dw;
        }
    }
#endif
    DH_BARRIER();
    uint32_t state = 0, best = m0;
    if (m1 < best) { best = m1; state = 1; }
    if (m2 < best) { best = m2; state = 2; }
    if (m3 < best) { best = m3; state = 3; }
    for (int blk = 10; blk >= 0; blk--) {
        const int steps = blk < 10 ? 32 : 10;
        uint32_t ow = 0;
        for (int g = 3; g >= 0; g--) {
            const uint32_t dw = dh_uniform(X.dec[(uint32_t) (blk * 4 + g)]);
            for (int qq = 7; qq >= 0; qq--) {
                const int q = g * 8 + qq;
                if (q >= steps) continue;
                ow |= (state >> 1) << q;                           // output bit pos in bit pos % 8 of byte pos / 8 (:98-103)
                state = ((state << 1) & 2u) | ((dw >> (4u * (uint32_t) qq + state)) & 1u);
            }
        }
        X.out[blk] = ow;
    }
    DH_BARRIER();
    if (best > 10u) return false;
    const uint32_t* out = X.out;
    const uint32_t crc = dh_dstar_crc([out](uint32_t k) { return (dh_uniform(out[k >> 2]) >> (8u * (k & 3u))) & 0xFFu; }, 39u);
    const uint32_t got = ((X.out[9] >> 24) & 0xFFu) | ((X.out[10] & 0xFFu) << 8);     // bytes 39, 40 (:54)
    return crc == dh_uniform(got);
}

// MetaCollector::setFromHeader as two events; get(k) delivers header byte k
template <typename Get> DH_HD void dh_dstar_emit_header(DhDecCtx& c, Get get, uint8_t source) {
    uint8_t part[24];
    for (uint32_t i = 0; i < 24u; i++) part[i] = (uint8_t) get(i);
    dh_emit(c, DH_EV_DSTAR_HEADER, 0, source, part, 24);
    for (uint32_t i = 0; i < 17u; i++) part[i] = (uint8_t) get(24u + i);
    dh_emit(c, DH_EV_DSTAR_HEADER, 1, source, part, 17);
}

// the two VoicePhase constructors (dstar_phase.cpp:60-70, dstar_phase.hpp:71-76)
DH_HD void dh_dstar_enter_voice(DhDecCtx& c, DhState& s, bool after_header) {
    s[DS_DT_FRAME] = after_header ? 21u : 0u;
    s[DS_SYNC_COUNT] = after_header ? 1u : 0u;
    s[DS_DT_COLLECT0] = 0; s[DS_DT_COLLECT1] = 0; s[DS_DT_BLOCKS] = 0; s[DS_DT_HCOUNT] = 0;
    for (int i = 0; i < 5; i++) s[DS_DT_MESSAGE + i] = 0;
    for (int i = 0; i < 11; i++) s[DS_DT_HEADER + i] = 0;
    dh_emit(c, DH_EV_DSTAR_VOICE_START, 0, after_header ? 1 : 0, nullptr, 0);
}

// n (<= 5) bytes of `data` (byte 0 in the low bits) into the byte string kept in state words base.., at byte offset off
DH_HD void dh_dstar_put_bytes(DhState& s, uint32_t base, uint32_t off, uint64_t data, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t at = off + i, w = base + (at >> 2), sh = 8u * (at & 3u);
        s[w] = ((uint32_t) s[w] & ~(0xFFu << sh)) | ((uint32_t) ((data >> (8u * i)) & 0xFFu) << sh);
    }
}

// VoicePhase::collectDataFrame (dstar_phase.cpp:153-203) on the descrambled 24 bits `x` of data frame `frame_count`
DH_HD void dh_dstar_collect(DhDecCtx& c, DhState& s, uint32_t frame_count, uint32_t x) {
    if ((frame_count & 1u) == 0u) { s[DS_DT_COLLECT0] = x; return; }
    s[DS_DT_COLLECT1] = x;
    const uint32_t c0 = s[DS_DT_COLLECT0];
    const uint32_t mini = c0 & 0xFFu, n = mini & 0x0Fu;
    const uint64_t data = (uint64_t) (c0 >> 8) | ((uint64_t) x << 16);       // collected_data[1..5]
    if ((mini >> 4) == 0x4u) {
        if (n <= 3u) { dh_dstar_put_bytes(s, DS_DT_MESSAGE, n * 5u, data, 5u); s[DS_DT_BLOCKS] = (uint32_t) s[DS_DT_BLOCKS] | (1u << n); }
    } else if ((mini >> 4) == 0x5u) {
        const uint32_t hc = s[DS_DT_HCOUNT];
        if (n <= 5u && hc + n <= 41u) { dh_dstar_put_bytes(s, DS_DT_HEADER, hc, data, n); s[DS_DT_HCOUNT] = hc + n; }
    } else if ((mini >> 4) == 0x3u) {
        if (n <= 5u) {
            uint8_t b[5];
            for (uint32_t i = 0; i < 5u; i++) b[i] = (uint8_t) (data >> (8u * i));
            dh_emit(c, DH_EV_DSTAR_SIMPLE, 0, 0, b, (int) n);
        }
    }
}

// end pattern in a data frame (dstar_phase.cpp:92-96): all 48 bits, or its second half alone, with at most one error
DH_HD bool dh_dstar_is_terminator(uint32_t d0, uint32_t d1) {
    return dh_popc32(d0 ^ DH_DSTAR_TERM_LO) + dh_popc32(d1 ^ DH_DSTAR_TERM_HI) <= 1 || dh_popc32(d0 ^ DH_DSTAR_TERM_HI) <= 1;
}

#define DH_DSTAR_BATCH 5          // data frames handled per round on the fast path (5 x 96 + 24 bits <= 512)

DH_HD void dh_dstar_channel(const DhDecParams& P, uint32_t ch, DhDecShared& S, uint32_t sym_base = 0, bool append = false) {
    DhDecCtx c;
    c.P = &P; c.T = &dh_lds_tables(S);
    uint32_t* const st_global = P.state + (size_t) ch * P.state_stride;
    DhState s; s.load(st_global);
    c.st = &s;
    c.out = P.out + (size_t) ch * P.out_stride;
    c.ev = P.events ? P.events + (size_t) ch * P.ev_stride : nullptr;
    c.nout = append ? P.out_count[ch] : 0u; c.nev = append && P.ev_count ? P.ev_count[ch] : 0u; c.overflow = false;      // (append: the second part of a split push, k_chain)
    c.consumed = s[DS_CONSUMED];
#if DH_DEVICE_BUILD && defined(__HIP_DEVICE_COMPILE__)
    c.writer = threadIdx.x == 0;
#else
    c.writer = true;
#endif
    uint8_t* const carry_buf = P.carry + (size_t) ch * P.carry_stride;
    uint8_t* const lds_carry = reinterpret_cast<uint8_t*>(S.vit_dec);          // 1536 bytes, no K=5 Viterbi in this protocol
    static_assert(sizeof(S.vit_dec) >= DH_DSTAR_CARRY_MAX, "D-Star carry");
    DhDstarScratch& X = *reinterpret_cast<DhDstarScratch*>(S.carry);
    DhSymView syms; syms.carry = lds_carry; syms.nc = s[DS_CARRY]; syms.fresh = P.syms + (size_t) ch * P.sym_stride + sym_base;
    syms.nfresh = P.sym_count[ch] - sym_base; syms.win = S.symwin; syms.wbase = 0; syms.wlen = 0;
    const uint32_t total = syms.nc + syms.nfresh;
    DH_FOR_LANES(lane) { for (uint32_t j = (uint32_t) lane; j < syms.nc; j += DH_WAVE) lds_carry[j] = carry_buf[j]; }
    DH_BARRIER();
    uint32_t pos = 0, phase = s[DS_PHASE];
    DhBits128 bits;

    for (;;) {
        const uint32_t avail = total - pos;
        if (phase == 0) {                                          // SyncPhase (dstar_phase.cpp:17-34): slide bit by bit
            if (!(avail > 24)) break;
            dh_view_ensure(syms, pos, 128);
            dh_dstar_take128(syms, pos, total, bits);
            uint64_t hh = 0, hv = 0;
            DH_FOR_LANES(lane) {
                const bool valid = avail > (uint32_t) lane && avail - (uint32_t) lane > 24;
                const uint32_t l = dh_bits_range(bits.l, (uint32_t) lane, 24), h = dh_bits_range(bits.h, (uint32_t) lane, 24);
                const bool header = valid && dh_popc32(l ^ DH_DSTAR_HEADER_SYNC) + dh_popc32(h) <= 2;
                const bool voice = valid && !header && dh_popc32(l ^ DH_DSTAR_VOICE_SYNC) + dh_popc32(h) <= 1;
                DH_BALLOT_ACC(hh, header, lane);
                DH_BALLOT_ACC(hv, voice, lane);
            }
            if (hh | hv) {
                const uint32_t l = (uint32_t) dh_ffs64(hh | hv);
                pos += l + 24u; c.consumed += l + 24u;
                if ((hh >> l) & 1ull) phase = 2;                   // HeaderPhase
                else { phase = 1; dh_dstar_enter_voice(c, s, false); }
            } else {
                const uint32_t adv = dh_min<uint32_t>(64u, avail - 24u);
                pos += adv; c.consumed += adv;
            }
            continue;
        }
        if (phase == 2) {                                          // HeaderPhase (dstar_phase.cpp:36-57)
            if (!(avail > 660)) break;
            dh_view_ensure(syms, pos, 660);
            const bool parsed = dh_dstar_header_parse(syms, pos, X);
            if (!parsed) { pos += 1u; c.consumed += 1u; phase = 0; continue; }
            pos += 660u; c.consumed += 660u;
            if (!((X.out[0] >> 7) & 1u)) {                         // isVoice (header.cpp:150-152)
                const uint32_t* out = X.out;
                dh_dstar_emit_header(c, [out](uint32_t k) { return (out[k >> 2] >> (8u * (k & 3u))) & 0xFFu; }, 0);
                phase = 1; dh_dstar_enter_voice(c, s, true);
            } else phase = 0;
            if (c.overflow) break;
            continue;
        }
        // VoicePhase::process (dstar_phase.cpp:76-139)
        if (!(avail > 120)) break;
        int sync_count = (int) s[DS_SYNC_COUNT];
        {
            // Fast path: up to DH_DSTAR_BATCH consecutive data frames that are not sync frames.  Packed LSB first, a
            // frame is 12 bytes: 9 voice bytes (the output, :81-86) and the 3 data bytes (:122-126).  Every lane loads
            // eight symbols of this push straight from HBM / L2 (three aligned words, funnel-shifted) and packs them
            // into ONE byte with a multiply: lane 12 f + i then holds byte i of frame f.  The voice bytes of all frames
            // leave in one store from those registers; end-pattern tests and slow-data bytes read the lanes they need
            // (v_readlane).  Symbols other than 0 / 1, a frame holding the end pattern, the carried symbols at the
            // start of a push and the last 516 symbols of a push take the one-frame path below.
            const uint32_t fc0 = s[DS_DT_FRAME];
            uint32_t nb = fc0 < 20u ? dh_min<uint32_t>(DH_DSTAR_BATCH, 20u - fc0) : 0u;
            const uint32_t fpos = pos - syms.nc;                    // offset in this push's symbols (valid when pos >= nc)
            const uint32_t mis = (uint32_t) ((uintptr_t) (syms.fresh + fpos) & 3u);
            if (nb >= 2u && pos >= syms.nc && fpos >= mis && fpos - mis + 516u <= syms.nfresh &&
                (sync_count < 1 || c.P->out_cap - c.nout >= 9u * nb)) {
                DH_LANE_VALUE(uint32_t, pkb);
                uint64_t odd = 0;
                const uint32_t* words = reinterpret_cast<const uint32_t*>(syms.fresh + fpos - mis);
                DH_FOR_LANES(lane) {
                    const uint32_t w0 = words[2 * lane], w1 = words[2 * lane + 1], w2 = words[2 * lane + 2];
                    const uint32_t lo = mis ? (w0 >> (8u * mis)) | (w1 << (32u - 8u * mis)) : w0;
                    const uint32_t hi = mis ? (w1 >> (8u * mis)) | (w2 << (32u - 8u * mis)) : w1;
                    DH_BALLOT_ACC(odd, ((lo | hi) & 0xFEFEFEFEu) != 0u, lane);
                    // bit 0 of byte k -> bit k: 0x01020408 = 2^24 + 2^17 + 2^10 + 2^3 moves them to bits 24..27
                    DH_LV(pkb, lane) = (((lo & 0x01010101u) * 0x01020408u) >> 24) | ((((hi & 0x01010101u) * 0x01020408u) >> 24) << 4);
                }
                if (odd == 0) {
                    uint32_t n = nb, xs[DH_DSTAR_BATCH];
#pragma unroll
                    for (uint32_t f = 0; f < DH_DSTAR_BATCH; f++) {
                        if (f >= n) break;
                        const uint32_t d0 = DH_LV_READ(pkb, 12u * f + 9u) | DH_LV_READ(pkb, 12u * f + 10u) << 8 | DH_LV_READ(pkb, 12u * f + 11u) << 16;
                        const uint32_t d1 = DH_LV_READ(pkb, 12u * f + 12u) | DH_LV_READ(pkb, 12u * f + 13u) << 8 | DH_LV_READ(pkb, 12u * f + 14u) << 16;
                        xs[f] = d0;
                        if (dh_dstar_is_terminator(d0, d1)) n = f;
                    }
                    if (n >= 1u) {
                        if (sync_count >= 1) {
                            uint8_t* o = c.out + c.nout;
                            DH_FOR_LANES(lane) {
                                const uint32_t f = (uint32_t) lane / 12u, i = (uint32_t) lane % 12u;
                                if (f < n && i < 9u) o[9u * f + i] = (uint8_t) DH_LV(pkb, lane);
                            }
                            c.nout += 9u * n;
                        }
                        constexpr DhDstarPn pn{};
#pragma unroll
                        for (uint32_t f = 0; f < DH_DSTAR_BATCH; f++) {
                            if (f >= n) break;
                            dh_dstar_collect(c, s, fc0 + f, xs[f] ^ (pn.w[0] & 0xFFFFFFu));
                            pos += 96u; c.consumed += 96u;
                        }
                        s[DS_DT_FRAME] = fc0 + n;
                        if (c.overflow) break;
                        continue;
                    }
                }
            }
        }
        dh_view_ensure(syms, pos, 128);
        dh_dstar_take128(syms, pos, total, bits);
        if (sync_count >= 1) {
            if (c.P->out_cap - c.nout < 9u) { c.overflow = true; break; }
            uint8_t* o = c.out + c.nout;
            DH_FOR_LANES(lane) { if (lane < 9) o[lane] = (uint8_t) ((lane < 8 ? bits.l[0] >> (8 * lane) : bits.l[1]) & 0xFFull); }
            c.nout += 9u;
        }
        const uint32_t d0 = dh_bits_range(bits.l, 72, 24), d0h = dh_bits_range(bits.h, 72, 24);
        const uint32_t d1 = dh_bits_range(bits.l, 96, 24), d1h = dh_bits_range(bits.h, 96, 24);
        if (dh_popc32(d0 ^ DH_DSTAR_TERM_LO) + dh_popc32(d0h) + dh_popc32(d1 ^ DH_DSTAR_TERM_HI) + dh_popc32(d1h) <= 1 ||
            dh_popc32(d0 ^ DH_DSTAR_TERM_HI) + dh_popc32(d0h) <= 1) {
            dh_emit(c, DH_EV_DSTAR_META_RESET, 0, 0, nullptr, 0);
            pos += 120u; c.consumed += 120u; phase = 0;
            if (c.overflow) break;
            continue;
        }
        uint32_t frame_count = s[DS_DT_FRAME];
        if (frame_count >= 20u) {                                  // isSyncDue
            bool lost = false;
            if (dh_popc32(d0 ^ DH_DSTAR_VOICE_SYNC) + dh_popc32(d0h) > 1) {
                if (--sync_count < 0) lost = true;
            } else {
                if (++sync_count > 3) sync_count = 3;
                if (sync_count > 1) dh_emit(c, DH_EV_DSTAR_SYNC_VOICE, 0, 0, nullptr, 0);
            }
            if (lost) {
                dh_emit(c, DH_EV_DSTAR_META_RESET, 0, 1, nullptr, 0);
                pos += 96u; c.consumed += 96u; phase = 0;
                if (c.overflow) break;
                continue;
            }
            s[DS_SYNC_COUNT] = (uint32_t) sync_count;
            // parseFrameData (:205-216) + resetFrames (:145-151)
            if ((uint32_t) s[DS_DT_BLOCKS] == 0xFu) {
                uint8_t msg[20];
                for (uint32_t i = 0; i < 20u; i++) msg[i] = (uint8_t) ((uint32_t) s[DS_DT_MESSAGE + (i >> 2)] >> (8u * (i & 3u)));
                dh_emit(c, DH_EV_DSTAR_MESSAGE, 0, 0, msg, 20);
            }
            if ((uint32_t) s[DS_DT_HCOUNT] == 41u) {
                DhState* sp = &s;
                auto get = [sp](uint32_t k) { return ((uint32_t) (*sp)[DS_DT_HEADER + (k >> 2)] >> (8u * (k & 3u))) & 0xFFu; };
                const uint32_t crc = dh_dstar_crc(get, 39u);
                if (crc == (get(39u) | (get(40u) << 8))) dh_dstar_emit_header(c, get, 1);
            }
            dh_emit(c, DH_EV_DSTAR_FRAME_SYNC, 0, 0, nullptr, 0);
            s[DS_DT_FRAME] = 0; s[DS_DT_BLOCKS] = 0; s[DS_DT_HCOUNT] = 0;
            for (int i = 0; i < 5; i++) s[DS_DT_MESSAGE + i] = 0;
            for (int i = 0; i < 11; i++) s[DS_DT_HEADER + i] = 0;
        } else {
            constexpr DhDstarPn pn{};
            dh_dstar_collect(c, s, frame_count, d0 ^ (pn.w[0] & 0xFFFFFFu));     // scrambler->reset() + 24 bits, packed LSB first (:120-129)
            s[DS_DT_FRAME] = frame_count + 1u;
        }
        pos += 96u; c.consumed += 96u;
        if (c.overflow) break;
    }

    const uint32_t rem = total - pos;
    dh_view_ensure(syms, pos, rem < DH_DSTAR_CARRY_MAX ? rem : DH_DSTAR_CARRY_MAX);
    DH_FOR_LANES(lane) {
        for (uint32_t j = lane; j < rem && j < DH_DSTAR_CARRY_MAX; j += DH_WAVE) carry_buf[j] = (uint8_t) dh_view_at(syms, pos + j);
        if (DH_IS_LANE0(lane)) {
            P.out_count[ch] = c.nout;
            if (P.ev_count) P.ev_count[ch] = c.nev;
            if ((c.overflow || rem > DH_DSTAR_CARRY_MAX) && P.overflow) *P.overflow = 1u;
        }
    }
    s[DS_PHASE] = phase; s[DS_CONSUMED] = c.consumed;
    s[DS_CARRY] = rem < DH_DSTAR_CARRY_MAX ? rem : DH_DSTAR_CARRY_MAX;
    s.store(st_global);
    DH_BARRIER();
}
