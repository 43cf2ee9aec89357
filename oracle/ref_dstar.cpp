/*
 * ref_dstar.cpp -- C entry points over the REFERENCE's own D-Star scrambler and CRC (TEST INFRASTRUCTURE ONLY).
 * Linked into oracle/_ref/libdigiham_ref_dstar.so together with the reference's unmodified
 * src/dstar_decoder/{scrambler,crc}.cpp compiled where they lie; those two files have no dependencies.
 * (header.cpp is not built: it includes charset.hpp, i.e. ICU.)
 */
#include "scrambler.hpp"
#include "crc.hpp"
#include <cstring>

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

}
