/*
 * ref_dmr.cpp -- batch C entry points over the REFERENCE's own DMR burst-element classes (TEST INFRASTRUCTURE ONLY).
 * Linked into oracle/_ref/libdigiham_ref_dmr.so together with the reference's unmodified
 *   src/dmr_decoder/{cach,tact,emb,embedded,slottype,lc,gps,talkeralias}.cpp, src/lib/{coordinate,charset}.cpp,
 *   src/dmr_decoder/{hamming_7_4,hamming_16_11,golay_20_8,quadratic_residue}.c
 * compiled where they lie (oracle/Makefile, target `ref`).  None of them includes csdr; charset.cpp uses ICU
 * (libicuuc), which this image ships.  The reference headers are included as they are; `private` is opened for
 * this translation unit only so that the corrected words an object holds can be read back (layouts unchanged).
 *
 * The entry points have the same signatures as the orc_el_dmr_* functions of oracle/elements.c, so that one
 * Python driver runs either implementation.
 */
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cstdlib>
#include <string>
#include <sstream>
#define private public
#include "cach.hpp"
#include "emb.hpp"
#include "embedded.hpp"
#include "slottype.hpp"
#include "lc.hpp"
#include "gps.hpp"
#include "talkeralias.hpp"
#undef private

using namespace Digiham::Dmr;

extern "C" {

/* Cach::parse + Tact getters.  raw [n][12] dibits -> out [n][8] = has_tact, tact (corrected), busy, slot, lcss, payload[3] */
void ref_el_dmr_cach(const uint8_t* raw, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; i++, raw += 12, out += 8) {
        Cach* c = Cach::parse(raw);
        std::memset(out, 0, 8);
        if (c->hasTact()) {
            Tact* t = c->getTact();
            out[0] = 1; out[1] = t->data; out[2] = t->isBusy(); out[3] = t->getSlot(); out[4] = t->getLcss();
        }
        std::memcpy(out + 5, c->payload, 3);
        delete c;
    }
}

/* Emb::parse.  out [n][4] = ok, colour code, lcss, 0; corrected[n] (input word when not ok) */
void ref_el_dmr_emb(const uint16_t* in, size_t n, uint8_t* out, uint16_t* corrected) {
    for (size_t i = 0; i < n; i++, out += 4) {
        Emb* e = Emb::parse(in[i]);
        std::memset(out, 0, 4);
        corrected[i] = in[i];
        if (e != nullptr) {
            out[0] = 1; out[1] = e->getColorCode(); out[2] = e->getLcss();
            corrected[i] = e->data;
            delete e;
        }
    }
}

/* SlotType::parse.  out [n][4] = ok, colour code, data type, 0; corrected[n] */
void ref_el_dmr_slottype(const uint32_t* in, size_t n, uint8_t* out, uint32_t* corrected) {
    for (size_t i = 0; i < n; i++, out += 4) {
        SlotType* s = SlotType::parse(in[i]);
        std::memset(out, 0, 4);
        corrected[i] = in[i];
        if (s != nullptr) {
            out[0] = 1; out[1] = s->getColorCode(); out[2] = s->getDataType();
            corrected[i] = s->data;
            delete s;
        }
    }
}

/* EmbeddedCollector: collect the four fragments of prev [n][16] (what an earlier superframe left in the buffer),
 * reset(), collect nfrags[i] fragments of frags [n][20] (a fifth one must be ignored), getLc().
 * out [n][10] = ok, lc[9] */
void ref_el_dmr_embedded_lc(const uint8_t* prev, const uint8_t* frags, const uint8_t* nfrags, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; i++, prev += 16, frags += 20, out += 10) {
        EmbeddedCollector c;
        for (int k = 0; k < 4; k++) c.collect(const_cast<uint8_t*>(prev + 4 * k));
        c.reset();
        for (int k = 0; k < nfrags[i]; k++) c.collect(const_cast<uint8_t*>(frags + 4 * k));
        Lc* lc = c.getLc();
        std::memset(out, 0, 10);
        if (lc != nullptr) {
            out[0] = 1;
            std::memcpy(out + 1, lc->data, 9);
            delete lc;
        }
    }
}

/* Lc getters over lc [n][9] (through parseFromVoiceHeader, which copies the first 9 bytes).
 * fields [n][4] = opcode, feature set id, source, target; data7 [n][7] = getData() */
void ref_el_dmr_lc(const uint8_t* lc, size_t n, uint32_t* fields, uint8_t* data7) {
    for (size_t i = 0; i < n; i++, lc += 9, fields += 4, data7 += 7) {
        Lc* l = Lc::parseFromVoiceHeader(const_cast<uint8_t*>(lc));
        fields[0] = l->getOpCode(); fields[1] = l->getFeatureSetId(); fields[2] = l->getSource(); fields[3] = l->getTarget();
        std::memcpy(data7, l->getData(), 7);
        delete l;
    }
}

/* Gps::parse over d [n][7] (= Lc::getData()).  latlon [n][2] */
void ref_el_dmr_gps(const uint8_t* d, size_t n, float* latlon) {
    for (size_t i = 0; i < n; i++, d += 7, latlon += 2) {
        Digiham::Coordinate* c = Gps::parse(d);
        latlon[0] = c->lat; latlon[1] = c->lon;
        delete c;
    }
}

/* TalkerAliasCollector: setBlock(order[k], blocks + 7 * order[k]) for k = 0.. until order[k] > 3; after every call
 * isComplete() is recorded (bit k of complete[i]); finally getContents().
 * blocks [n][28], order [n][4], complete [n], text [n][64] (zero padded), len [n] (bytes of the string) */
void ref_el_dmr_talkeralias(const uint8_t* blocks, const uint8_t* order, size_t n, uint8_t* complete, uint8_t* text, uint8_t* len) {
    for (size_t i = 0; i < n; i++, blocks += 28, order += 4, text += 64) {
        TalkerAliasCollector c;
        std::memset(c.data, 0, 28);                 /* the reference reads malloc'd memory here; a fresh heap page is zero */
        complete[i] = 0;
        for (int k = 0; k < 4 && order[k] < 4; k++) {
            c.setBlock(order[k], const_cast<uint8_t*>(blocks + 7 * order[k]));
            try {
                if (c.isComplete()) complete[i] |= (uint8_t) (1 << k);
            } catch (...) { complete[i] |= (uint8_t) (0x10 << k); }
        }
        std::memset(text, 0, 64);
        try {
            std::string s = c.getContents();
            len[i] = (uint8_t) (s.size() > 64 ? 64 : s.size());
            std::memcpy(text, s.data(), len[i]);
        } catch (...) {                             /* wstring_convert throws on a lone UTF-16 surrogate (the reference would abort) */
            len[i] = 255;
        }
    }
}

}
