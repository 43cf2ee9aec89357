/*
 * ref_nxdn.cpp -- C entry points over the REFERENCE's own NXDN frame-element classes (TEST INFRASTRUCTURE
 * ONLY).  Linked into oracle/_ref/libdigiham_ref_nxdn.so together with the reference's unmodified
 * src/nxdn_decoder/{scrambler,lich,sacch,facch1,trellis}.cpp and src/lib/hamming_distance.c, compiled where
 * they lie (oracle/Makefile, target `ref`); those five files do not include csdr.  The reference headers are
 * included as they are; `private` is opened for this translation unit only so that the decoded bytes a
 * Sacch / Facch1 object holds can be read back (the class layouts are unchanged).
 */
#define private public
#include "scrambler.hpp"
#include "lich.hpp"
#include "sacch.hpp"
#include "facch1.hpp"
#include "trellis.hpp"
#undef private
#include <cstring>

using namespace Digiham::Nxdn;

extern "C" {

void ref_nxdn_scramble(const unsigned char* in, unsigned char* out, size_t len) {     /* from the reset state */
    Scrambler s;
    s.reset();
    s.scramble(const_cast<unsigned char*>(in), out, len);
}

int ref_nxdn_lich_parse(const unsigned char* raw8) {
    Lich* l = Lich::parse(const_cast<unsigned char*>(raw8));
    if (l == nullptr) return -1;
    const int v = l->data;
    delete l;
    return v;
}

unsigned ref_nxdn_trellis_decode(const unsigned char* input, unsigned char* output, size_t len_bits) {
    Trellis t;
    return t.decode(const_cast<unsigned char*>(input), output, len_bits);
}

int ref_nxdn_sacch_parse(const unsigned char* dibits30, unsigned char* out5) {
    Sacch* s = Sacch::parse(const_cast<unsigned char*>(dibits30));
    if (s == nullptr) return 0;
    std::memcpy(out5, s->data, 5);
    delete s;
    return 1;
}

int ref_nxdn_facch1_parse(const unsigned char* dibits72, unsigned char* out12) {
    Facch1* f = Facch1::parse(const_cast<unsigned char*>(dibits72));
    if (f == nullptr) return 0;
    std::memcpy(out12, f->data, 12);
    delete f;
    return 1;
}

}
