/*
 * ref_dstar.cpp -- C entry points over the REFERENCE's own D-Star scrambler and CRC (TEST INFRASTRUCTURE ONLY).
 * Linked into oracle/_ref/libdigiham_ref_dstar.so together with the reference's unmodified
 * src/dstar_decoder/{scrambler,crc,header}.cpp, src/lib/charset.cpp and src/lib/hamming_distance.c compiled where
 * they lie; none includes csdr, charset.cpp uses ICU (libicuuc), which this image ships.  `private` is opened for
 * this translation unit only so that the decoded header bytes can be read back (layouts unchanged).
 */
#include <cstring>
#include <cstdlib>
#include <string>
#include "scrambler.hpp"
#include "crc.hpp"
#define private public
#include "header.hpp"
#undef private

using namespace Digiham::DStar;

extern "C" {

void ref_dstar_scramble(const unsigned char* in, unsigned char* out, size_t len) {       /* from the reset state */
    Scrambler s;
    s.reset();
    s.scramble(const_cast<unsigned char*>(in), out, len);
}

/* 1 when isCrcValid(data, len, checksum) */
int ref_dstar_crc_valid(const unsigned char* data, size_t len, unsigned short checksum) {
    return Crc::isCrcValid(const_cast<unsigned char*>(data), len, checksum) ? 1 : 0;
}

/* Header::parseFromHeader over raw [n][660] received bits (one per byte).  ok [n]; data [n][41] decoded header bytes
 * (zero when not ok); text [n][160] = toString() zero padded (the four call sign fields through ICU iso-8859-1 -> utf-8) */
void ref_el_dstar_header(const unsigned char* raw, size_t n, unsigned char* ok, unsigned char* data, unsigned char* text) {
    for (size_t i = 0; i < n; i++, raw += 660, data += 41, text += 160) {
        Header* h = Header::parseFromHeader(const_cast<unsigned char*>(raw));
        ok[i] = h != nullptr;
        std::memset(data, 0, 41);
        std::memset(text, 0, 160);
        if (h != nullptr) {
            std::memcpy(data, h->data, 41);
            std::string s = h->toString();
            std::memcpy(text, s.data(), s.size() > 159 ? 159 : s.size());
            delete h;
        }
    }
}

}
