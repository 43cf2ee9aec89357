// elements_test.cpp -- runs the host-side element parsers of include/digiham/ over binary vectors (stdin -> stdout) so
// that tests/test_host_elements.py can hold them to the reference's own results (tests/golden/elements_ref.npz).
//   elements_test dmr_gps      n x 7 bytes              -> n x 2 float32
//   elements_test talkeralias  n x (28 blocks + 4 order) -> n x (1 complete bits + 1 length + 64 text bytes)
//   elements_test lc           n x 9 bytes              -> n x (4 uint32 + 7 bytes)
//   elements_test ysf_gps      n x 9 bytes              -> n x (1 ok + 2 float32)
//   elements_test latin1       n x 16 bytes             -> n x (1 length + 32 text bytes)
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "digiham/dmr_meta.hpp"
#include "digiham/ysf_meta.hpp"

template <size_t N> static bool get(unsigned char (&buf)[N]) { return fread(buf, 1, N, stdin) == N; }
static void put(const void* p, size_t n) { fwrite(p, 1, n, stdout); }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string what = argv[1];
    if (what == "dmr_gps") {
        unsigned char d[7];
        while (get(d)) {
            std::unique_ptr<Digiham::Coordinate> c(Digiham::Dmr::Gps::parse(d));
            const float out[2] = { c->lat, c->lon };
            put(out, sizeof(out));
        }
    } else if (what == "talkeralias") {
        unsigned char in[32];
        while (get(in)) {
            Digiham::Dmr::TalkerAliasCollector c;
            unsigned char out[66] = { 0 };
            for (int k = 0; k < 4 && in[28 + k] < 4; k++) {
                c.setBlock(in[28 + k], in + 7 * in[28 + k]);
                if (c.isComplete()) out[0] |= (unsigned char) (1 << k);
            }
            const std::string s = c.getContents();
            out[1] = (unsigned char) (s.size() > 64 ? 64 : s.size());
            std::memcpy(out + 2, s.data(), out[1]);
            put(out, sizeof(out));
        }
    } else if (what == "lc") {
        unsigned char d[9];
        while (get(d)) {
            Digiham::Dmr::Lc lc(d);
            const uint32_t f[4] = { lc.getOpCode(), lc.getFeatureSetId(), lc.getSource(), lc.getTarget() };
            put(f, sizeof(f));
            put(lc.getData(), 7);
        }
    } else if (what == "ysf_gps") {
        unsigned char d[9];
        while (get(d)) {
            std::unique_ptr<Digiham::Coordinate> c(Digiham::Ysf::Gps::parse(d));
            const unsigned char ok = c != nullptr;
            const float out[2] = { c ? c->lat : 0.0f, c ? c->lon : 0.0f };
            put(&ok, 1); put(out, sizeof(out));
        }
    } else if (what == "latin1") {
        unsigned char d[16];
        while (get(d)) {
            const std::string s = Digiham::Converter::convertToUtf8((const char*) d, 16);
            unsigned char out[33] = { 0 };
            out[0] = (unsigned char) s.size();
            std::memcpy(out + 1, s.data(), s.size());
            put(out, sizeof(out));
        }
    } else return 2;
    return 0;
}
