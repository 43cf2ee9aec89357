// meta_test.cpp -- replays decoder events (as the GPU engine or the CPU wave emulation produced them) through the
// host-side metadata collectors of include/digiham/{dmr,ysf}_meta.hpp and prints the metadata lines.
//   meta_test <dmr|ysf|nxdn|dstar> < batches      batches = repeated { uint32 n; dh_event[n] }  (one batch per decoder call)
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "digiham/dmr_meta.hpp"
#include "digiham/ysf_meta.hpp"
#include "digiham/nxdn_meta.hpp"
#include "digiham/dstar_meta.hpp"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string proto = argv[1];
    Digiham::MetaCollector* c = proto == "dmr" ? (Digiham::MetaCollector*) new Digiham::Dmr::MetaCollector()
                              : proto == "ysf" ? (Digiham::MetaCollector*) new Digiham::Ysf::MetaCollector()
                              : proto == "dstar" ? (Digiham::MetaCollector*) new Digiham::DStar::MetaCollector()
                              : (Digiham::MetaCollector*) new Digiham::Nxdn::MetaCollector();
    c->setWriter(new Digiham::FileMetaWriter(fdopen(1, "w")));
    uint32_t n;
    while (fread(&n, sizeof(n), 1, stdin) == 1) {
        std::vector<dh_event> ev(n);
        if (n && fread(ev.data(), sizeof(dh_event), n, stdin) != n) return 3;
        for (const auto& e : ev) c->consume(e);
        c->flush();
    }
    delete c;
    return 0;
}
