/*
 * decoders.c -- oracle restatement of Digiham::Decoder + the DMR and YSF phase
 * state machines (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED for the state machines (reference TUs need csdr headers, see
 * dh_oracle.h); the FEC they call is the pinned fec.c and the burst / frame elements
 * (CACH/TACT, EMB, slot type, embedded LC, FICH) are the pinned elements.c.  Metadata bookkeeping
 * (MetaCollector, talker alias, GPS, callsign strings) is out of scope: every
 * call the reference makes into its MetaCollector is recorded as an orc_event
 * carrying the raw FEC-corrected bytes instead.
 */
#include "dh_oracle.h"
#include <stdlib.h>
#include <string.h>

enum { PROTO_DMR = 1, PROTO_YSF = 2, PROTO_NXDN = 3, PROTO_POCSAG = 4, PROTO_DSTAR = 5 };
enum { PH_HEADER = 2 };                 /* D-Star's third phase (dstar_phase.hpp:50) */
enum { PH_SYNC = 0, PH_FRAME = 1 };

/* ---- DMR constants: src/dmr_decoder/dmr_phase.hpp:6-12,25-33 */
#define DMR_SYNC_SIZE 24
#define DMR_CACH_SIZE 12
#define DMR_FRAME_SIZE 144
#define DMR_SYNC_OFFSET (54 + DMR_CACH_SIZE)
#define SYNCTYPE_DATA 1
#define SYNCTYPE_VOICE 2
static const uint8_t dmr_bs_data_sync[24]  = { 3,1,3,3,3,3,1,1,1,3,3,1,1,3,1,1,3,1,3,3,1,1,3,1 };
static const uint8_t dmr_bs_voice_sync[24] = { 1,3,1,1,1,1,3,3,3,1,1,3,3,1,3,3,1,3,1,1,3,3,1,3 };
static const uint8_t dmr_ms_data_sync[24]  = { 3,1,1,1,3,1,1,3,3,3,1,3,1,3,3,3,3,1,1,3,1,1,1,3 };
static const uint8_t dmr_ms_voice_sync[24] = { 1,3,3,3,1,3,3,1,1,1,3,1,3,1,1,1,1,3,3,1,3,3,3,1 };
/* emb.hpp:5-8, slottype.hpp:6-17 */
enum { LCSS_SINGLE = 0, LCSS_START = 1, LCSS_STOP = 2, LCSS_CONTINUATION = 3 };
enum { DT_VOICE_LC = 1, DT_TERMINATOR_LC = 2, DT_RATE_3_4_DATA = 8, DT_IDLE = 9 };

/* ---- YSF constants: src/ysf_decoder/ysf_phase.hpp:7-12,21; fich.hpp:3-11 */
#define YSF_SYNC_SIZE 20
#define YSF_FICH_SIZE 100
#define YSF_PAYLOAD_SIZE 360
#define YSF_FRAME_SIZE 480
static const uint8_t ysf_sync[20] = { 3,1,1,0,1,3,0,1,3,0,2,1,1,2,0,3,1,0,3,1 };
enum { FT_HEADER = 0, FT_COMMUNICATION = 1, FT_TERMINATOR = 2 };
enum { YDT_VD1 = 0, YDT_DATA_FR = 1, YDT_VD2 = 2, YDT_VOICE_FR = 3 };
static const uint8_t tribit_majority_table[8] = { 0, 0, 0, 1, 0, 1, 1, 1 };
/* AMBE bit order of the V/D type 2 voice channel (ysf_phase.hpp:46-51) */
static const uint8_t v2_voice_mapping[49] = {
    0, 3, 6,  9, 12, 15, 18, 21, 24, 27, 30, 33, 36, 39, 41, 43, 45, 47,
    1, 4, 7, 10, 13, 16, 19, 22, 25, 28, 31, 34, 37, 40, 42, 44, 46, 48,
    2, 5, 8, 11, 14, 17, 20, 23, 26, 29, 32, 35, 38,
};

typedef struct { uint8_t data[16]; uint8_t offset; } emb_collector;   /* embedded.hpp:7-19 */

struct orc_decoder {
    int proto, phase;
    uint64_t consumed;       /* symbols consumed so far (for event stamps) */
    /* sinks for the current process() call */
    uint8_t* out; size_t out_cap, out_n;
    orc_event* ev; size_t ev_cap, ev_n;
    int overflow;
    /* Dmr::Decoder (dmr_decoder.hpp:16) */
    uint8_t slot_filter_decoder;
    /* Dmr::FramePhase (dmr_phase.hpp:51-60) */
    int sync_count, slot, slot_stability, sync_types[2], slot_sync_count[2], active_slot;
    emb_collector emb[2];
    uint8_t slot_filter, superframe_counter[2];
    /* Ysf::FramePhase (ysf_phase.hpp:53-56) */
    int has_running_fich; uint32_t running_fich; int expect_sub_frame;
    /* Nxdn::FramedPhase (nxdn_phase.hpp:31-40): syncCount shares sync_count; lich < 0 = nullptr;
     * SacchSuperframeCollector (sacch.hpp:37-46): collected[i] != nullptr <=> bit i of sacch_have */
    int lich; uint8_t sacch_have; uint8_t sacch_data[4][4];
    /* Pocsag::CodewordPhase (pocsag_phase.hpp:27-36; syncCount shares sync_count) + its Message (message.hpp:11-23) */
    int codeword_counter, has_message; uint32_t msg_address; uint8_t msg_type; int msg_pos; char msg_content[80];
    /* DStar::VoicePhase (dstar_phase.hpp:67-77; syncCount shares sync_count); simpleData lives with the event consumer */
    int frame_count; uint8_t collected_data[6], ds_message[20], ds_message_blocks, ds_header[41], ds_header_count;
};

static void emit(orc_decoder* d, uint8_t type, uint8_t a, uint8_t b, const uint8_t* payload, uint8_t len) {
    if (d->ev == NULL) return;
    if (d->ev_n >= d->ev_cap) { d->overflow = 1; return; }
    orc_event* e = &d->ev[d->ev_n++];
    memset(e, 0, sizeof(*e));
    e->sym_index = (uint32_t) d->consumed;
    e->type = type; e->a = a; e->b = b; e->len = len;
    if (len) memcpy(e->payload, payload, len);
}

static void enter_dmr_frame_phase(orc_decoder* d) {
    /* FramePhase::FramePhase() (dmr_phase.cpp:49-52) + member initialisers (dmr_phase.hpp:51-60);
     * Decoder::setPhase pushes the decoder's slot filter into the new phase (dmr_decoder.cpp:17-23) */
    d->sync_count = 0; d->slot = -1; d->slot_stability = 0;
    d->sync_types[0] = d->sync_types[1] = -1;
    d->slot_sync_count[0] = d->slot_sync_count[1] = 0;
    memset(d->emb, 0, sizeof(d->emb));
    d->active_slot = -1;
    d->superframe_counter[0] = d->superframe_counter[1] = 0;
    d->slot_filter = d->slot_filter_decoder;
    if (((d->active_slot + 1) & d->slot_filter) == 0) d->active_slot = -1;   /* dmr_phase.cpp:341-345 */
}

static void enter_ysf_frame_phase(orc_decoder* d) {
    d->sync_count = 0; d->has_running_fich = 0; d->running_fich = 0; d->expect_sub_frame = 0;
}

orc_decoder* orc_dmr_new(void) {
    orc_decoder* d = (orc_decoder*) calloc(1, sizeof(orc_decoder));
    d->proto = PROTO_DMR; d->phase = PH_SYNC; d->slot_filter_decoder = 3;   /* dmr_decoder.hpp:16 */
    return d;
}

orc_decoder* orc_ysf_new(void) {
    orc_decoder* d = (orc_decoder*) calloc(1, sizeof(orc_decoder));
    d->proto = PROTO_YSF; d->phase = PH_SYNC;
    return d;
}

orc_decoder* orc_nxdn_new(void) {
    orc_decoder* d = (orc_decoder*) calloc(1, sizeof(orc_decoder));
    d->proto = PROTO_NXDN; d->phase = PH_SYNC; d->lich = -1;
    return d;
}

orc_decoder* orc_pocsag_new(void) {
    orc_decoder* d = (orc_decoder*) calloc(1, sizeof(orc_decoder));
    d->proto = PROTO_POCSAG; d->phase = PH_SYNC;
    return d;
}

orc_decoder* orc_dstar_new(void) {
    orc_decoder* d = (orc_decoder*) calloc(1, sizeof(orc_decoder));
    d->proto = PROTO_DSTAR; d->phase = PH_SYNC;
    return d;
}

void orc_decoder_free(orc_decoder* d) { free(d); }

/* dmr_decoder.cpp:9-15 + dmr_phase.cpp:341-345 */
void orc_dmr_set_slot_filter(orc_decoder* d, uint8_t filter) {
    d->slot_filter_decoder = filter;
    if (d->phase == PH_FRAME) {
        d->slot_filter = filter;
        if (((d->active_slot + 1) & d->slot_filter) == 0) d->active_slot = -1;
    }
}

/* ================================================================== DMR */

/* dmr_phase.cpp:18-33 */
static int dmr_get_sync_type(const uint8_t* s) {
    if (orc_hamming_distance(s, dmr_bs_data_sync, DMR_SYNC_SIZE) <= 3) return SYNCTYPE_DATA;
    if (orc_hamming_distance(s, dmr_bs_voice_sync, DMR_SYNC_SIZE) <= 3) return SYNCTYPE_VOICE;
    if (orc_hamming_distance(s, dmr_ms_data_sync, DMR_SYNC_SIZE) <= 3) return SYNCTYPE_DATA;
    if (orc_hamming_distance(s, dmr_ms_voice_sync, DMR_SYNC_SIZE) <= 3) return SYNCTYPE_VOICE;
    return -1;
}

static void dmr_slot_sync_lost(orc_decoder* d) {
    /* dmr_phase.cpp:175-182 == :194-200 */
    if (--d->slot_sync_count[d->slot] < 0) {
        d->slot_sync_count[d->slot] = 0;
        d->sync_types[d->slot] = -1;
        emit(d, ORC_EV_DMR_SLOT_RESET, (uint8_t) d->slot, 0, NULL, 0);
        if (d->active_slot == d->slot) d->active_slot = -1;
    }
}

/* dmr_phase.cpp:65-302.  returns 1 when the phase falls back to SyncPhase (no advance) */
static int dmr_frame(orc_decoder* d, const uint8_t* p) {
    uint8_t tact = 0;                                    /* Cach::parse + Tact::getSlot (elements.c) */
    int tact_slot = orc_dmr_cach_parse(p, &tact, NULL) ? (tact >> 5) & 1 : -1;
    uint8_t next = (uint8_t) (d->slot ^ 1);              /* :69 (0xFE while slot == -1) */
    if (tact_slot >= 0) {
        if ((uint8_t) tact_slot != next) {
            if (d->slot_stability < 5) {
                d->slot_stability = 0;
                d->slot = tact_slot;
                uint8_t other = (uint8_t) (d->slot ^ 1);
                d->sync_types[other] = -1;
                emit(d, ORC_EV_DMR_SLOT_RESET, other, 1, NULL, 0);      /* b = 1: the other slot after a TACT slot switch (:80) */
                if (d->active_slot == other) d->active_slot = -1;
            } else {
                d->slot_stability--;
                if (d->slot != -1) d->slot = next;
            }
        } else {
            if (++d->slot_stability > 100) d->slot_stability = 100;
            d->slot = next;
        }
    } else if (d->slot != -1) {
        if (d->slot_stability-- < -100) d->slot_stability = -100;
        d->slot = next;
    }

    if (d->slot != -1) {
        int slot = d->slot;
        int sync_type = dmr_get_sync_type(p + DMR_SYNC_OFFSET);
        if (sync_type > 0) {
            if (++d->sync_count > 5) d->sync_count = 5;
            if (++d->slot_sync_count[slot] > 5) d->slot_sync_count[slot] = 5;
            uint8_t soft_reset = d->sync_types[slot] == SYNCTYPE_VOICE && sync_type != d->sync_types[slot];
            d->sync_types[slot] = sync_type;
            emit(d, ORC_EV_DMR_SYNC, (uint8_t) slot, (uint8_t) sync_type, &soft_reset, 1);
            d->superframe_counter[slot] = 0;
            d->emb[slot].offset = 0;
        } else if (d->sync_types[slot] == SYNCTYPE_VOICE && d->superframe_counter[slot] < 5) {
            d->superframe_counter[slot]++;
            uint16_t emb_data = 0;
            for (int i = 0; i < 2; i++) {
                const uint8_t* raw = p + DMR_SYNC_OFFSET + i * 20;
                for (int k = 0; k < 4; k++) emb_data = (uint16_t) ((emb_data << 2) | raw[k]);
            }
            if (orc_dmr_emb_parse(&emb_data)) {
                if (++d->sync_count > 5) d->sync_count = 5;
                if (++d->slot_sync_count[slot] > 5) d->slot_sync_count[slot] = 5;
                uint8_t embedded_data[4] = { 0 };
                const uint8_t* emb_raw = p + DMR_SYNC_OFFSET + 4;
                for (int i = 0; i < 16; i++) embedded_data[i / 4] |= (uint8_t) (emb_raw[i] << (6 - (i % 4) * 2));
                emb_collector* c = &d->emb[slot];
                uint8_t lcss = orc_dmr_emb_lcss(emb_data), cc = orc_dmr_emb_color_code(emb_data);
                emit(d, ORC_EV_DMR_EMB, (uint8_t) slot, lcss, &cc, 1);
                switch (lcss) {
                    case LCSS_SINGLE: break;
                    case LCSS_START:
                        c->offset = 0;
                        /* fall through */
                    case LCSS_CONTINUATION:
                        if (c->offset <= 3) { memcpy(c->data + c->offset * 4, embedded_data, 4); c->offset++; }
                        break;
                    case LCSS_STOP: {
                        if (c->offset <= 3) { memcpy(c->data + c->offset * 4, embedded_data, 4); c->offset++; }
                        uint8_t lc[9];
                        if (orc_dmr_embedded_get_lc(c->data, c->offset, lc)) emit(d, ORC_EV_DMR_LC, (uint8_t) slot, 1, lc, 9);
                        c->offset = 0;
                        break;
                    }
                }
            } else {
                dmr_slot_sync_lost(d);
                if (--d->sync_count < 0) {
                    emit(d, ORC_EV_DMR_META_RESET, 0, 0, NULL, 0);
                    return 1;
                }
            }
        } else {
            d->superframe_counter[slot] = 0;
            d->emb[slot].offset = 0;
            dmr_slot_sync_lost(d);
            if (--d->sync_count < 0) {
                emit(d, ORC_EV_DMR_META_RESET, 0, 0, NULL, 0);
                return 1;
            }
        }

        if (d->sync_types[slot] == SYNCTYPE_VOICE) {
            if (((slot + 1) & d->slot_filter) && (d->active_slot == -1 || d->active_slot == slot)) {
                d->active_slot = slot;
                if (d->out_cap - d->out_n < 27) { d->overflow = 1; }
                else {
                    uint8_t* payload = d->out + d->out_n;
                    memset(payload, 0, 27);
                    const uint8_t* raw = p + DMR_CACH_SIZE;
                    for (int i = 0; i < 54; i++) payload[i / 4] |= (uint8_t) ((raw[i] & 3) << (6 - 2 * (i % 4)));
                    raw += 54 + DMR_SYNC_SIZE;
                    for (int i = 0; i < 54; i++) payload[(i + 54) / 4] |= (uint8_t) ((raw[i] & 3) << (6 - 2 * ((i + 54) % 4)));
                    d->out_n += 27;
                }
            }
        } else {
            if (d->active_slot == slot) d->active_slot = -1;
            if (d->sync_types[slot] == SYNCTYPE_DATA) {
                uint32_t slot_type = 0;
                const uint8_t* raw = p + DMR_SYNC_OFFSET - 5;
                for (int i = 0; i < 5; i++) slot_type = (slot_type << 2) | (raw[i] & 3);
                raw = p + DMR_SYNC_OFFSET + DMR_SYNC_SIZE;
                for (int i = 0; i < 5; i++) slot_type = (slot_type << 2) | (raw[i] & 3);
                if (orc_dmr_slottype_parse(&slot_type)) {
                    uint8_t data_type = orc_dmr_slottype_data_type(slot_type), cc = orc_dmr_slottype_color_code(slot_type);
                    emit(d, ORC_EV_DMR_SLOTTYPE, (uint8_t) slot, data_type, &cc, 1);
                    if (data_type != DT_RATE_3_4_DATA) {
                        uint8_t payload[25] = { 0 };
                        const uint8_t* pr = p + DMR_CACH_SIZE;
                        for (int k = 0; k < 49; k++) payload[k / 4] |= (uint8_t) ((pr[k] & 3) << (6 - 2 * (k % 4)));
                        pr += 54 + DMR_SYNC_SIZE + 5;
                        for (int k = 0; k < 49; k++) payload[(k + 49) / 4] |= (uint8_t) ((pr[k] & 3) << (6 - 2 * ((k + 49) % 4)));
                        uint8_t lc_data[12] = { 0 };
                        if (orc_bptc_196_96(payload, lc_data)) {
                            emit(d, ORC_EV_DMR_BPTC, (uint8_t) slot, data_type, lc_data, 12);
                            if (data_type == DT_VOICE_LC) {
                                emit(d, ORC_EV_DMR_LC, (uint8_t) slot, 0, lc_data, 9);
                            } else if (data_type == DT_TERMINATOR_LC || data_type == DT_IDLE) {
                                emit(d, ORC_EV_DMR_SOFT_RESET, (uint8_t) slot, data_type, NULL, 0);
                            }
                        }
                    }
                }
            } else {
                emit(d, ORC_EV_DMR_SLOT_RESET, (uint8_t) slot, 0, NULL, 0);
            }
        }
    }
    return 0;
}

/* ================================================================== YSF */

/* ysf_phase.cpp:221-239 */
static void ysf_decode_tribits(const uint8_t* input, uint8_t* output, uint8_t num) {
    memset(output, 0, (size_t) (num + 7) / 8);
    for (int i = 0; i < num; i++) {
        int offset = i * 3;
        uint8_t tribit = 0;
        for (int k = 0; k < 3; k++) {
            int pos = (offset + k) / 8, shift = 7 - ((offset + k) % 8);
            tribit = (uint8_t) ((tribit << 1) | ((input[pos] >> shift) & 1));
        }
        output[i / 8] |= (uint8_t) (tribit_majority_table[tribit] << (7 - (i % 8)));
    }
}

/* ysf_phase.cpp:180-219 + :241-256 */
static void ysf_decode_v2_voice(const uint8_t* in, uint8_t* out) {
    uint8_t inter[13] = { 0 };
    for (int k = 0; k < 52; k++) inter[k / 4] |= (uint8_t) ((in[k] & 3) << (6 - 2 * (k % 4)));
    uint8_t whitened[13] = { 0 };
    for (int k = 0; k < 104; k++) {
        int offset = (k * 4) % 104 + k * 4 / 104;
        whitened[k / 8] |= (uint8_t) (((inter[offset / 8] >> (7 - offset % 8)) & 1) << (7 - k % 8));
    }
    uint8_t tribit[13] = { 0 };
    orc_decode_whitening(whitened, tribit, 104);
    uint8_t voice[7] = { 0 };
    ysf_decode_tribits(tribit, voice, 27);
    for (int k = 0; k < 22; k++) {
        int ib = k + 81, ob = k + 27;
        voice[ob / 8] |= (uint8_t) (((tribit[ib / 8] >> (7 - (ib % 8))) & 1) << (7 - (ob % 8)));
    }
    for (int i = 0; i < 7; i++) out[i] = 0;
    for (int ib = 0; ib < 49; ib++) {
        int ob = v2_voice_mapping[ib];
        uint8_t x = (voice[ib / 8] >> (7 - (ib % 8))) & 1;
        out[ob / 8] |= (uint8_t) (x << (7 - (ob % 8)));
    }
}

/* ysf_phase.cpp:317-349; returns 1 and fills dch[20] on success */
static int ysf_decode_header_dch(const uint8_t* in, uint8_t* dch) {
    uint8_t raw[45] = { 0 };
    for (int i = 0; i < 180; i++) {
        int streampos = (i % 9) * 20 + i / 9;
        int inpos = (streampos / 36) * 72 + streampos % 36;
        raw[i / 4] |= (uint8_t) ((in[inpos] & 3) << (6 - 2 * (i % 4)));
    }
    uint8_t whitened[23] = { 0 };
    orc_decode_trellis(raw, 180, whitened);
    uint16_t checksum = (uint16_t) ((whitened[20] << 8) | whitened[21]);
    if (orc_crc16_checksum(whitened, 20) != checksum) return 0;
    orc_decode_whitening(whitened, dch, 160);
    return 1;
}

/* ysf_phase.cpp:45-172.  returns 1 on fall-back to SyncPhase (nothing consumed) */
static int ysf_frame(orc_decoder* d, const uint8_t* p) {
    if (orc_hamming_distance(p, ysf_sync, YSF_SYNC_SIZE) <= 3) {
        if (++d->sync_count > 12) d->sync_count = 12;
    } else {
        if (--d->sync_count < 0) {
            emit(d, ORC_EV_YSF_META_RESET, 0, 0, NULL, 0);
            return 1;
        }
    }
    uint32_t fich = 0;
    int fresh = orc_ysf_fich_parse(p + YSF_SYNC_SIZE, &fich);
    if (fresh) {
        d->running_fich = fich; d->has_running_fich = 1;
        uint8_t be[4] = { (uint8_t) (fich >> 24), (uint8_t) (fich >> 16), (uint8_t) (fich >> 8), (uint8_t) fich };
        emit(d, ORC_EV_YSF_FICH, 0, 0, be, 4);
    }
    const uint8_t* payload = p + YSF_SYNC_SIZE + YSF_FICH_SIZE;
    if (d->has_running_fich) {
        uint8_t frame_type = (d->running_fich >> 30) & 3;
        uint8_t data_type = (d->running_fich >> 8) & 3;
        switch (frame_type) {
            case FT_COMMUNICATION:
                emit(d, ORC_EV_YSF_MODE, 0, data_type, NULL, 0);
                switch (data_type) {
                    case YDT_VD1:
                        for (int i = 0; i < 5; i++) {
                            if (d->out_cap - d->out_n < 10) { d->overflow = 1; break; }
                            uint8_t* o = d->out + d->out_n;
                            o[0] = data_type;
                            /* ysf_phase.cpp:174-178: `=` (not `|=`): only dibit k%4==3 of each byte survives */
                            const uint8_t* in = payload + 36 + i * 72;
                            for (int k = 0; k < 36; k++) o[1 + k / 4] = (uint8_t) ((in[k] & 3) << (6 - 2 * (k % 4)));
                            d->out_n += 10;
                        }
                        break;
                    case YDT_VD2:
                        for (int i = 0; i < 5; i++) {
                            if (d->out_cap - d->out_n < 8) { d->overflow = 1; break; }
                            uint8_t* o = d->out + d->out_n;
                            o[0] = data_type;
                            ysf_decode_v2_voice(payload + 20 + i * 72, o + 1);
                            d->out_n += 8;
                        }
                        if (fresh) {
                            uint8_t dch_raw[25] = { 0 };
                            for (int i = 0; i < 100; i++) {
                                int inpos = ((i % 5) * 72 + (i * 2) / 10);
                                dch_raw[i / 4] |= (uint8_t) ((payload[inpos] & 3) << (6 - 2 * (i % 4)));
                            }
                            /* decodeV2DataChannel, ysf_phase.cpp:258-269 (metadata tail :271-305 out of scope) */
                            uint8_t whitened[13] = { 0 };
                            orc_decode_trellis(dch_raw, 100, whitened);
                            uint16_t checksum = (uint16_t) (whitened[10] << 8 | whitened[11]);
                            if (orc_crc16_checksum(whitened, 10) == checksum) {
                                uint8_t dch[13] = { 0 };
                                orc_decode_whitening(whitened, dch, 100);
                                emit(d, ORC_EV_YSF_DCH, (uint8_t) ((fich >> 19) & 7), 0, dch, 10);
                            }
                        }
                        break;
                    case YDT_VOICE_FR: {
                        int start_frame = d->expect_sub_frame ? 3 : 0;
                        d->expect_sub_frame = 0;
                        for (int i = start_frame; i < 5; i++) {
                            if (d->out_cap - d->out_n < 19) { d->overflow = 1; break; }
                            uint8_t* o = d->out + d->out_n;
                            o[0] = data_type;
                            memset(o + 1, 0, 18);
                            const uint8_t* in = payload + i * 72;
                            for (int k = 0; k < 72; k++) o[1 + k / 4] |= (uint8_t) ((in[k] & 3) << (6 - 2 * (k % 4)));
                            d->out_n += 19;
                        }
                        break;
                    }
                    case YDT_DATA_FR: break;
                }
                break;
            case FT_HEADER: {
                emit(d, ORC_EV_YSF_META_RESET, 0, 1, NULL, 0);
                uint8_t dch[20];
                if (ysf_decode_header_dch(payload, dch)) emit(d, ORC_EV_YSF_HEADER_DCH, 0, 0, dch, 20);
                if (ysf_decode_header_dch(payload + 36, dch)) emit(d, ORC_EV_YSF_HEADER_DCH, 1, 0, dch, 20);
                d->expect_sub_frame = 1;
                break;
            }
            case FT_TERMINATOR:
                emit(d, ORC_EV_YSF_META_RESET, 0, 2, NULL, 0);
                break;
        }
    }
    return 0;
}

/* ================================================================== NXDN */
/* nxdn_phase.hpp:12-15, nxdn_phase.cpp:15-16 */
#define NXDN_SYNC_SIZE 10
#define NXDN_FRAME_SIZE 192
static const uint8_t nxdn_sync[NXDN_SYNC_SIZE] = { 3, 0, 3, 1, 3, 3, 1, 1, 2, 1 };
/* lich.hpp:5-27, types.hpp:1-3 */
enum { NXDN_RF_RCCH = 0, NXDN_USC_UDCH = 1, NXDN_USC_SACCH_SF = 2, NXDN_MSG_TX_RELEASE = 0x08, NXDN_MSG_IDLE = 0x10 };

static void enter_nxdn_framed_phase(orc_decoder* d) {           /* FramedPhase::FramedPhase(), nxdn_phase.cpp:32-35 */
    d->sync_count = 0; d->lich = -1; d->sacch_have = 0;
    memset(d->sacch_data, 0, sizeof(d->sacch_data));
}

/* FramedPhase::process (nxdn_phase.cpp:43-170).  Returns the symbols consumed; *to_sync = 1 when the phase falls
 * back to SyncPhase (sync lost: nothing consumed; TX_RELEASE: the FACCH1 block itself is not consumed, :155-159). */
static size_t nxdn_frame(orc_decoder* d, const uint8_t* p, int* to_sync) {
    *to_sync = 0;
    if (orc_hamming_distance(p, nxdn_sync, NXDN_SYNC_SIZE) <= 2) {
        if (++d->sync_count > 6) d->sync_count = 6;
    } else if (--d->sync_count < 0) {
        emit(d, ORC_EV_NXDN_META_RESET, 0, 0, NULL, 0);
        *to_sync = 1;
        return 0;
    }
    size_t pos = NXDN_SYNC_SIZE;
    uint16_t sr = 0x0E4;                                        /* scrambler->reset(), scrambler.cpp:9-11 */
    uint8_t lich_descrambled[8];
    orc_nxdn_scramble(&sr, p + pos, lich_descrambled, 8);
    pos += 8;
    const int new_lich = orc_nxdn_lich_parse(lich_descrambled);
    if (new_lich >= 0) {
        d->lich = new_lich;
        const uint8_t b = (uint8_t) new_lich;
        emit(d, ORC_EV_NXDN_LICH, 0, 0, &b, 1);
    }
    if (d->lich >= 0 && ((d->lich >> 5) & 3) != NXDN_RF_RCCH && ((d->lich >> 3) & 3) != NXDN_USC_UDCH) {
        uint8_t sacch_descrambled[30];
        orc_nxdn_scramble(&sr, p + pos, sacch_descrambled, 30);
        if (((d->lich >> 3) & 3) == NXDN_USC_SACCH_SF) {
            uint8_t sacch[5];
            if (orc_nxdn_sacch_parse(sacch_descrambled, sacch)) {
                const int index = (sacch[0] >> 6) ^ 3;                       /* Sacch::getStructureIndex, sacch.cpp:16-18 */
                emit(d, ORC_EV_NXDN_SACCH, (uint8_t) index, 0, sacch, 5);
                /* SacchSuperframeCollector::push (sacch.cpp:90-98) */
                if (!(index > 0 && !(d->sacch_have & (1u << (index - 1))))) {
                    d->sacch_have |= (uint8_t) (1u << index);
                    memcpy(d->sacch_data[index], sacch + 1, 4);
                }
                if (d->sacch_have == 0xF) {                                  /* isComplete + getSuperframe (:109-131) */
                    uint8_t sf[9];
                    memset(sf, 0, 9);
                    for (int i = 0; i < 4; i++) for (int k = 0; k < 18; k++) {
                        const int outpos = i * 18 + k;
                        sf[outpos / 8] |= (uint8_t) (((d->sacch_data[i][k / 8] >> (7 - k % 8)) & 1u) << (7 - outpos % 8));
                    }
                    emit(d, ORC_EV_NXDN_SACCH_SF, 0, 0, sf, 9);
                    d->sacch_have = 0;                                       /* sacchCollector->reset() */
                }
            }
        }
        pos += 30;
        const unsigned option = (unsigned) (d->lich >> 1) & 3u;
        for (int i = 0; i < 2; i++) {
            uint8_t voice_descrambled[72];
            orc_nxdn_scramble(&sr, p + pos, voice_descrambled, 72);
            if ((option >> (1 - i)) & 1u) {
                if (d->sync_count >= 1) {
                    emit(d, ORC_EV_NXDN_SYNC_VOICE, 0, 0, NULL, 0);
                    if (d->out_cap - d->out_n < 18) { d->overflow = 1; return pos; }
                    uint8_t* o = d->out + d->out_n;
                    memset(o, 0, 18);
                    for (int k = 0; k < 72; k++) o[k / 4] |= (uint8_t) ((voice_descrambled[k] & 3u) << (6 - ((k % 4) * 2)));
                    d->out_n += 18;
                }
            } else {
                uint8_t facch1[12];
                if (orc_nxdn_facch1_parse(voice_descrambled, facch1)) {
                    emit(d, ORC_EV_NXDN_FACCH1, (uint8_t) i, 0, facch1, 12);
                    if ((facch1[0] & 0x3F) == NXDN_MSG_TX_RELEASE) {
                        emit(d, ORC_EV_NXDN_META_RESET, 0, 1, NULL, 0);
                        *to_sync = 1;
                        return pos;
                    }
                }
            }
            pos += 72;
        }
    } else {
        pos += 174;
    }
    return pos;
}

/* ================================================================ POCSAG */
/* pocsag_phase.hpp:8,15; codeword.hpp:5-6,23; message.hpp:8 */
#define POCSAG_SYNC_SIZE 32
#define POCSAG_CODEWORD_SIZE 32
#define POCSAG_CODEWORDS_PER_SYNC 16
#define POCSAG_MAX_MESSAGE_LENGTH 80
static const uint8_t pocsag_sync[32] = { 0,1,1,1,1,1,0,0,1,1,0,1,0,0,1,0,0,0,0,1,0,1,0,1,1,1,0,1,1,0,0,0 };
#define POCSAG_IDLE 0x7A89C197u

/* Message::serialize (message.cpp:17-25) with the StringSerializer (meta.cpp:8-17): `address:<n>;message:<text>\n`,
 * the text ending at its first NUL */
static void pocsag_serialize(orc_decoder* d) {
    if (!d->has_message || d->msg_pos == 0) return;
    char line[160];
    size_t n = 0;
    const char* a = "address:";
    memcpy(line + n, a, 8); n += 8;
    char num[16]; int nn = 0; uint32_t v = d->msg_address;
    do { num[nn++] = (char) ('0' + v % 10); v /= 10; } while (v);
    while (nn) line[n++] = num[--nn];
    memcpy(line + n, ";message:", 9); n += 9;
    for (int i = 0; i < POCSAG_MAX_MESSAGE_LENGTH && d->msg_content[i]; i++) line[n++] = d->msg_content[i];
    line[n++] = '\n';
    if (d->out_cap - d->out_n < n) { d->overflow = 1; return; }
    memcpy(d->out + d->out_n, line, n);
    d->out_n += n;
}

/* Message::append (message.cpp:27-72) */
static void pocsag_append(orc_decoder* d, uint32_t data) {
    switch (d->msg_type) {
        case 3:
            if (d->msg_pos + 20 < POCSAG_MAX_MESSAGE_LENGTH * 7) {
                for (int i = 0; i < 20; i++) {
                    const unsigned bit = (data >> (19 - i)) & 1u;
                    d->msg_content[d->msg_pos / 7] |= (char) (bit << (d->msg_pos % 7));
                    d->msg_pos++;
                }
            }
            break;
        case 0:
            if (d->msg_pos + 5 < POCSAG_MAX_MESSAGE_LENGTH) {
                static const char tail[6] = { '*', 'U', ' ', '-', ')', '(' };
                for (int i = 0; i < 5; i++) {
                    char c = 0;
                    const unsigned base = (unsigned) (4 - i) * 4;
                    for (int k = 0; k < 4; k++) c |= (char) (((data >> (base + k)) & 1u) << (3 - k));
                    c = c < 0xA ? (char) ('0' + c) : tail[c - 0xA];
                    d->msg_content[d->msg_pos++] = c;
                }
            }
            break;
    }
}

static void pocsag_drop_message(orc_decoder* d) { d->has_message = 0; }

/* CodewordPhase::process (pocsag_phase.cpp:38-92).  Returns the bits consumed; *to_sync = 1 on fall-back to SyncPhase. */
static size_t pocsag_step(orc_decoder* d, const uint8_t* p, int* to_sync) {
    *to_sync = 0;
    if (d->codeword_counter >= POCSAG_CODEWORDS_PER_SYNC) {
        if (orc_hamming_distance(p, pocsag_sync, POCSAG_SYNC_SIZE) <= 3) {
            if (d->sync_count++ > 2) d->sync_count = 2;
        } else if (d->sync_count-- < 0) {
            pocsag_serialize(d);
            *to_sync = 1;
            return 0;
        }
        d->codeword_counter = 0;
        return POCSAG_SYNC_SIZE;
    }
    uint32_t cw;
    if (orc_pocsag_codeword_parse(p, &cw)) {
        const uint8_t be[4] = { (uint8_t) (cw >> 24), (uint8_t) (cw >> 16), (uint8_t) (cw >> 8), (uint8_t) cw };
        emit(d, ORC_EV_POCSAG_CODEWORD, (uint8_t) d->codeword_counter, 0, be, 4);
        if (cw == POCSAG_IDLE) {                                   /* codeword.cpp:36-38 */
            pocsag_serialize(d);
            pocsag_drop_message(d);
        } else if ((cw >> 31) == 0) {                              /* address codeword (:45-47) */
            pocsag_serialize(d);
            pocsag_drop_message(d);
            const uint8_t type = (uint8_t) ((cw >> 11) & 3u);      /* getFunctionBits (:54-56) */
            if (type == 1 || type == 3) {
                d->has_message = 1;
                d->msg_address = (((cw >> 13) & 0x3FFFFu) << 3) | (uint32_t) (d->codeword_counter / 2);
                d->msg_type = type; d->msg_pos = 0;
                memset(d->msg_content, 0, sizeof(d->msg_content));
            }
        } else if (d->has_message) {
            pocsag_append(d, (cw >> 11) & 0xFFFFFu);               /* getPayload (:40-43) */
        }
    } else {
        pocsag_drop_message(d);
    }
    d->codeword_counter++;
    return POCSAG_CODEWORD_SIZE;
}

/* ================================================================ D-Star */
/* dstar_phase.hpp:12-13,18-41 */
#define DSTAR_SYNC_SIZE 24
#define DSTAR_TERMINATOR_SIZE 48
#define DSTAR_HEADER_BITS 660
#define DSTAR_VOICE_REQUIRED (72 + 24 + 24)
static const uint8_t dstar_header_sync[24] = { 0,1,0,1,0,1,0,1,0, 1,1,1,0,1,1,0,0,1,0,1,0,0,0,0 };
static const uint8_t dstar_voice_sync[24]  = { 1,0,1,0,1,0,1,0,1,0, 1,1,0,1,0,0,0, 1,1,0,1,0,0,0 };
static const uint8_t dstar_terminator[48]  = { 1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,1,0,
                                               0,0,0,1,0,0,1,1,0,1,0,1,1,1,1,0 };

/* the two VoicePhase constructors (dstar_phase.cpp:60-70) + member initialisers (dstar_phase.hpp:71-76) */
static void enter_dstar_voice_phase(orc_decoder* d, int after_header) {
    d->frame_count = after_header ? 21 : 0;
    d->sync_count = after_header ? 1 : 0;
    memset(d->collected_data, 0, 6); memset(d->ds_message, 0, 20); memset(d->ds_header, 0, 41);
    d->ds_message_blocks = 0; d->ds_header_count = 0;
    emit(d, ORC_EV_DSTAR_VOICE_START, 0, (uint8_t) after_header, NULL, 0);    /* a new, empty simpleData */
}

/* MetaCollector::setFromHeader(header) as two events (41 bytes do not fit one payload) */
static void dstar_emit_header(orc_decoder* d, const uint8_t* h41, uint8_t source) {
    emit(d, ORC_EV_DSTAR_HEADER, 0, source, h41, 24);
    emit(d, ORC_EV_DSTAR_HEADER, 1, source, h41 + 24, 17);
}

/* VoicePhase::collectDataFrame (dstar_phase.cpp:153-203) */
static void dstar_collect_data_frame(orc_decoder* d, const uint8_t* data3) {
    memcpy(d->collected_data + (d->frame_count % 2) * 3, data3, 3);
    if (d->frame_count % 2 == 0) return;
    const uint8_t* c = d->collected_data;
    const int n = c[0] & 0x0F;
    switch (c[0] >> 4) {
        case 0x04:
            if (n > 3) break;
            memcpy(d->ds_message + n * 5, c + 1, 5);
            d->ds_message_blocks |= (uint8_t) (1 << n);
            break;
        case 0x05:
            if (n > 5) break;
            if (d->ds_header_count + n > 41) break;
            memcpy(d->ds_header + d->ds_header_count, c + 1, (size_t) n);
            d->ds_header_count = (uint8_t) (d->ds_header_count + n);
            break;
        case 0x03:
            if (n > 5) break;
            emit(d, ORC_EV_DSTAR_SIMPLE, 0, 0, c + 1, (uint8_t) n);           /* simpleData += ... (:178) */
            break;
        default: break;                                                        /* reserved / unknown: only a log line */
    }
}

/* VoicePhase::parseFrameData (:205-216; the simpleData line parser runs at the consumer on DSTAR_FRAME_SYNC) */
static void dstar_parse_frame_data(orc_decoder* d) {
    if (d->ds_message_blocks == 0x0F) emit(d, ORC_EV_DSTAR_MESSAGE, 0, 0, d->ds_message, 20);
    if (d->ds_header_count == 41 && orc_dstar_header_crc_ok(d->ds_header)) dstar_emit_header(d, d->ds_header, 1);
    emit(d, ORC_EV_DSTAR_FRAME_SYNC, 0, 0, NULL, 0);
}

/* VoicePhase::process (dstar_phase.cpp:76-139).  Returns the bits consumed; *to_sync = 1 when it returns a SyncPhase. */
static size_t dstar_voice(orc_decoder* d, const uint8_t* p, int* to_sync) {
    *to_sync = 0;
    if (d->sync_count >= 1) {
        if (d->out_n + 9 > d->out_cap) { d->overflow = 1; return 0; }
        uint8_t* o = d->out + d->out_n;
        memset(o, 0, 9);
        for (int i = 0; i < 72; i++) o[i / 8] |= (uint8_t) ((p[i] & 1) << (i % 8));
        d->out_n += 9;
    }
    const uint8_t* data_frame = p + 72;
    if (orc_hamming_distance(data_frame, dstar_terminator, DSTAR_TERMINATOR_SIZE) <= 1 ||
        orc_hamming_distance(data_frame, dstar_terminator + 24, DSTAR_TERMINATOR_SIZE - 24) <= 1) {
        emit(d, ORC_EV_DSTAR_META_RESET, 0, 0, NULL, 0);
        *to_sync = 1;
        return 72 + 24 + 24;
    }
    if (d->frame_count >= 20) {                                               /* isSyncDue (:141-143) */
        if (orc_hamming_distance(data_frame, dstar_voice_sync, DSTAR_SYNC_SIZE) > 1) {
            if (--d->sync_count < 0) {
                emit(d, ORC_EV_DSTAR_META_RESET, 0, 1, NULL, 0);
                *to_sync = 1;
                return 72 + 24;
            }
        } else {
            if (++d->sync_count > 3) d->sync_count = 3;
            if (d->sync_count > 1) emit(d, ORC_EV_DSTAR_SYNC_VOICE, 0, 0, NULL, 0);
        }
        dstar_parse_frame_data(d);
        d->frame_count = 0;                                                   /* resetFrames (:145-151) */
        memset(d->ds_message, 0, 20); d->ds_message_blocks = 0;
        memset(d->ds_header, 0, 41); d->ds_header_count = 0;
    } else {
        uint8_t sr = 0x7F, descrambled[24], bytes[3] = { 0, 0, 0 };           /* scrambler->reset() (:120) */
        orc_dstar_scramble(&sr, data_frame, descrambled, 24);
        for (int i = 0; i < 24; i++) bytes[i / 8] |= (uint8_t) (descrambled[i] << (i % 8));
        dstar_collect_data_frame(d, bytes);
        d->frame_count++;
    }
    return 72 + 24;
}

/* ============================================ Digiham::Decoder main loop */
/* `while (canProcess()) process()` with canProcess = available > required
 * (src/lib/decoder.cpp:21-32; cli.cpp:29-33) */
size_t orc_decoder_process(orc_decoder* d, const uint8_t* in, size_t n,
                           uint8_t* out, size_t cap, size_t* n_out,
                           orc_event* ev, size_t ev_cap, size_t* n_ev) {
    d->out = out; d->out_cap = cap; d->out_n = 0;
    d->ev = ev; d->ev_cap = ev_cap; d->ev_n = 0; d->overflow = 0;
    size_t pos = 0;
    for (;;) {
        size_t avail = n - pos;
        const uint8_t* p = in + pos;
        if (d->proto == PROTO_DMR) {
            if (d->phase == PH_SYNC) {
                if (!(avail > DMR_SYNC_SIZE + DMR_SYNC_OFFSET)) break;      /* dmr_phase.cpp:35-37 */
                if (dmr_get_sync_type(p + DMR_SYNC_OFFSET) > 0) {           /* :39-47 */
                    d->phase = PH_FRAME; enter_dmr_frame_phase(d);
                } else { pos++; d->consumed++; }
            } else {
                if (!(avail > DMR_FRAME_SIZE)) break;                        /* :61-63 */
                if (dmr_frame(d, p)) d->phase = PH_SYNC;
                else { pos += DMR_FRAME_SIZE; d->consumed += DMR_FRAME_SIZE; }
            }
        } else if (d->proto == PROTO_POCSAG) {
            if (!(avail > POCSAG_SYNC_SIZE)) break;                          /* both phases need 32 (pocsag_phase.cpp:14,34) */
            if (d->phase == PH_SYNC) {
                if (orc_hamming_distance(p, pocsag_sync, POCSAG_SYNC_SIZE) <= 3) {   /* :18-28 */
                    pos += POCSAG_SYNC_SIZE; d->consumed += POCSAG_SYNC_SIZE;
                    d->phase = PH_FRAME;
                    d->sync_count = 1; d->codeword_counter = 0; d->has_message = 0;  /* pocsag_phase.hpp:31-34 */
                } else { pos++; d->consumed++; }
            } else {
                int to_sync;
                const size_t used = pocsag_step(d, p, &to_sync);
                pos += used; d->consumed += used;
                if (to_sync) { d->phase = PH_SYNC; d->has_message = 0; }
            }
        } else if (d->proto == PROTO_DSTAR) {
            if (d->phase == PH_SYNC) {
                if (!(avail > DSTAR_SYNC_SIZE)) break;                       /* dstar_phase.hpp:46 */
                if (orc_hamming_distance(p, dstar_header_sync, DSTAR_SYNC_SIZE) <= 2) {        /* dstar_phase.cpp:20-23 */
                    pos += DSTAR_SYNC_SIZE; d->consumed += DSTAR_SYNC_SIZE; d->phase = PH_HEADER;
                } else if (orc_hamming_distance(p, dstar_voice_sync, DSTAR_SYNC_SIZE) <= 1) {  /* :25-28 */
                    pos += DSTAR_SYNC_SIZE; d->consumed += DSTAR_SYNC_SIZE; d->phase = PH_FRAME;
                    enter_dstar_voice_phase(d, 0);
                } else { pos++; d->consumed++; }
            } else if (d->phase == PH_HEADER) {
                if (!(avail > DSTAR_HEADER_BITS)) break;                     /* dstar_phase.hpp:52 */
                uint8_t h[41];
                if (!orc_dstar_header_parse(p, h)) {                         /* dstar_phase.cpp:39-44 */
                    pos++; d->consumed++; d->phase = PH_SYNC;
                } else {
                    pos += DSTAR_HEADER_BITS; d->consumed += DSTAR_HEADER_BITS;
                    if (!((h[0] >> 7) & 1)) {                                /* isVoice (:48-54) */
                        dstar_emit_header(d, h, 0);
                        d->phase = PH_FRAME; enter_dstar_voice_phase(d, 1);
                    } else d->phase = PH_SYNC;
                }
            } else {
                if (!(avail > DSTAR_VOICE_REQUIRED)) break;                  /* dstar_phase.hpp:62 */
                int to_sync;
                const size_t used = dstar_voice(d, p, &to_sync);
                pos += used; d->consumed += used;
                if (to_sync) d->phase = PH_SYNC;
            }
        } else if (d->proto == PROTO_NXDN) {
            if (d->phase == PH_SYNC) {
                if (!(avail > NXDN_SYNC_SIZE)) break;                        /* nxdn_phase.hpp:23 */
                if (orc_hamming_distance(p, nxdn_sync, NXDN_SYNC_SIZE) <= 2) {  /* nxdn_phase.cpp:18-30 */
                    d->phase = PH_FRAME; enter_nxdn_framed_phase(d);
                } else { pos++; d->consumed++; }
            } else {
                if (!(avail > NXDN_FRAME_SIZE)) break;                       /* nxdn_phase.hpp:33 */
                int to_sync;
                const size_t used = nxdn_frame(d, p, &to_sync);
                pos += used; d->consumed += used;
                if (to_sync) d->phase = PH_SYNC;
            }
        } else {
            if (d->phase == PH_SYNC) {
                if (!(avail > YSF_SYNC_SIZE)) break;                         /* ysf_phase.cpp:20-22 */
                if (orc_hamming_distance(p, ysf_sync, YSF_SYNC_SIZE) <= 3) {   /* :25-34 */
                    d->phase = PH_FRAME; enter_ysf_frame_phase(d);
                } else { pos++; d->consumed++; }
            } else {
                if (!(avail > YSF_FRAME_SIZE)) break;                        /* :41-43 */
                if (ysf_frame(d, p)) d->phase = PH_SYNC;
                else { pos += YSF_FRAME_SIZE; d->consumed += YSF_FRAME_SIZE; }
            }
        }
        if (d->overflow) break;
    }
    if (n_out) *n_out = d->out_n;
    if (n_ev) *n_ev = d->ev_n;
    return pos;
}
