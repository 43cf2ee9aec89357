/*
 * ref_pocsag.cpp -- batch C entry point over the REFERENCE's own POCSAG Codeword class (TEST INFRASTRUCTURE ONLY).
 * Linked into oracle/_ref/libdigiham_ref_pocsag.so together with the reference's unmodified
 * src/pocsag_decoder/{codeword.cpp,bch_31_21.c} compiled where they lie (no csdr).  `private` is opened for this
 * translation unit only.  Same signature as orc_el_pocsag_codeword (elements.c).
 */
#include <cstdint>
#include <cstddef>
#include <cstring>
#define private public
#include "codeword.hpp"
#undef private

using namespace Digiham::Pocsag;

extern "C" {

/* Codeword::parse over bits [n][32] (one received bit per byte; any non-zero byte counts as 1, codeword.cpp:12).
 * out [n][4] = ok, isIdle, isAddressCodeword, function bits; words [n][3] = corrected word, payload (20 bit), address (18 bit) */
void ref_el_pocsag_codeword(const uint8_t* bits, size_t n, uint8_t* out, uint32_t* words) {
    for (size_t i = 0; i < n; i++, bits += 32, out += 4, words += 3) {
        Codeword* c = Codeword::parse(const_cast<uint8_t*>(bits));
        std::memset(out, 0, 4);
        words[0] = words[1] = words[2] = 0;
        if (c != nullptr) {
            out[0] = 1; out[1] = c->isIdle(); out[2] = c->isAddressCodeword(); out[3] = c->getFunctionBits();
            words[0] = c->data; words[1] = c->getPayload(); words[2] = c->getAddress();
            delete c;
        }
    }
}

}
