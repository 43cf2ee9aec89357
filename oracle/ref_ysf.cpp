/*
 * ref_ysf.cpp -- batch C entry points over the REFERENCE's own YSF frame-element classes (TEST INFRASTRUCTURE ONLY).
 * Linked into oracle/_ref/libdigiham_ref_ysf.so together with the reference's unmodified
 *   src/ysf_decoder/{fich,data,gps}.cpp, src/lib/coordinate.cpp,
 *   src/ysf_decoder/{trellis,golay_24_12,crc16,radio_types}.c, src/lib/hamming_distance.c
 * compiled where they lie (oracle/Makefile, target `ref`); none of them includes csdr.  `private` is opened for
 * this translation unit only (layouts unchanged).  Same signatures as the orc_el_ysf_* functions of elements.c.
 */
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <cstdlib>
#include <string>
#define private public
#include "fich.hpp"
#include "data.hpp"
#include "gps.hpp"
#undef private

using namespace Digiham::Ysf;

extern "C" {

/* Fich::parse over dibits [n][100] (one dibit per byte, as the frame delivers them).
 * out [n][4] = ok, frame type, data type, frame number; data [n] = the 32-bit FICH (0 when not ok) */
void ref_el_ysf_fich(const uint8_t* dibits, size_t n, uint8_t* out, uint32_t* data) {
    for (size_t i = 0; i < n; i++, dibits += 100, out += 4) {
        Fich* f = Fich::parse(const_cast<uint8_t*>(dibits));
        std::memset(out, 0, 4);
        data[i] = 0;
        if (f != nullptr) {
            out[0] = 1; out[1] = f->getFrameType(); out[2] = f->getDataType(); out[3] = f->getFrameNumber();
            data[i] = f->data;
            delete f;
        }
    }
}

/* Gps::parse over d [n][9].  ok [n], latlon [n][2] (0 when not ok).
 * NOTE gps.cpp:30-50 leaves `lon` uninitialised unless (d[4] & 0xF0) is 0x50 or 0x30: callers only pass such rows. */
void ref_el_ysf_gps(const uint8_t* d, size_t n, uint8_t* ok, float* latlon) {
    for (size_t i = 0; i < n; i++, d += 9, latlon += 2) {
        Digiham::Coordinate* c = Gps::parse(d);
        ok[i] = c != nullptr;
        latlon[0] = latlon[1] = 0.0f;
        if (c != nullptr) { latlon[0] = c->lat; latlon[1] = c->lon; delete c; }
    }
}

/* DataCollector + DataFrame: for every item a script of up to 8 steps; step k: collect(chunks + 10 k, offsets[k])
 * (offsets[k] > 1 ends the script; the reference asserts offset < 2).  After each step hasCollected(2) goes to bit k
 * of has2[i].  At the end, if hasCollected(2): getDataFrame() -> frame [n][4] = non-null, command (24 bit),
 * gps-non-null, 0; radio [n][32] zero padded; latlon [n][2].
 * chunks [n][80], offsets [n][8] */
void ref_el_ysf_data(const uint8_t* chunks, const uint8_t* offsets, size_t n, uint8_t* has2, uint32_t* frame, uint8_t* radio, float* latlon) {
    for (size_t i = 0; i < n; i++, chunks += 80, offsets += 8, frame += 4, radio += 32, latlon += 2) {
        DataCollector c;
        std::memset(c.data, 0, 20);                 /* malloc'd in the reference; a fresh heap page is zero */
        has2[i] = 0;
        for (int k = 0; k < 8 && offsets[k] < 2; k++) {
            c.collect(const_cast<uint8_t*>(chunks + 10 * k), offsets[k]);
            if (c.hasCollected(2)) has2[i] |= (uint8_t) (1 << k);
        }
        frame[0] = frame[1] = frame[2] = frame[3] = 0;
        std::memset(radio, 0, 32);
        latlon[0] = latlon[1] = 0.0f;
        if (c.hasCollected(2)) {
            DataFrame* f = c.getDataFrame();
            if (f != nullptr) {
                frame[0] = 1; frame[1] = f->getCommand();
                std::string r = f->getRadio();
                std::memcpy(radio, r.data(), r.size() > 31 ? 31 : r.size());
                /* getGpsCoordinate() -> Gps::parse(data + 5): see the note above about (data[9] & 0xF0) */
                const uint8_t b = f->data[9] & 0xF0;
                if (f->getCommand() != COMMAND_SHORT_GPS || b == 0x50 || b == 0x30) {
                    Digiham::Coordinate* g = f->getGpsCoordinate();
                    if (g != nullptr) { frame[2] = 1; latlon[0] = g->lat; latlon[1] = g->lon; delete g; }
                } else frame[3] = 1;                /* skipped: undefined behaviour in the reference */
                delete f;
            }
        }
    }
}

}
