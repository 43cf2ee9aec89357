// pipe_test.cpp -- drives the Csdr::Module-shaped classes of include/digiham/ the way the reference's CLI
// driver does (`while (module->canProcess()) module->process();`, src/lib/cli.cpp:29-33) over memory-backed
// readers / writers, and dumps every stage's output for comparison with the oracle.
//   pipe_test <proto dmr|ysf> <in.f32> <out_prefix> [chunk]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "digiham/rrc_filter.hpp"
#include "digiham/gfsk_demodulator.hpp"
#include "digiham/dmr_decoder.hpp"
#include "digiham/ysf_decoder.hpp"
#include "digiham/digitalvoice_filter.hpp"

template <typename T> struct MemReader: Csdr::Reader<T> {
    std::vector<T> data; size_t pos = 0, limit = 0;
    size_t available() override { return limit - pos; }
    T* getReadPointer() override { return data.data() + pos; }
    void advance(size_t n) override { pos += n; }
};
template <typename T> struct MemWriter: Csdr::Writer<T> {
    std::vector<T> data; size_t pos = 0;
    explicit MemWriter(size_t cap): data(cap) {}
    size_t writeable() override { return data.size() - pos; }
    T* getWritePointer() override { return data.data() + pos; }
    void advance(size_t n) override { pos += n; }
};

template <typename T> static void dump(const std::string& path, const T* p, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    fwrite(p, sizeof(T), n, f);
    fclose(f);
}

template <typename M, typename T, typename U> static void run(M& m, MemReader<T>& r, MemWriter<U>& w, size_t chunk) {
    Csdr::Module<T, U>* b = &m;
    b->setReader(&r); b->setWriter(&w);
    // feed the reader in chunks, as a pipe would
    while (r.limit < r.data.size()) {
        r.limit = std::min(r.data.size(), r.limit + chunk);
        while (b->canProcess()) b->process();
    }
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string proto = argv[1], prefix = argv[3];
    const size_t chunk = argc > 4 ? (size_t) atol(argv[4]) : 4096;
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END); long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    MemReader<float> in; in.data.resize(bytes / sizeof(float));
    if (fread(in.data.data(), sizeof(float), in.data.size(), f) != in.data.size()) return 4;
    fclose(f);
    try {
        Digiham::RrcFilter::WideRrcFilter rrc;
        MemWriter<float> filtered(in.data.size());
        run(rrc, in, filtered, chunk);
        dump(prefix + ".filtered", filtered.data.data(), filtered.pos);

        MemReader<float> fin; fin.data.assign(filtered.data.begin(), filtered.data.begin() + filtered.pos);
        Digiham::Fsk::GfskDemodulator gfsk(10);
        MemWriter<unsigned char> syms(in.data.size() / 9 + 64);
        run(gfsk, fin, syms, chunk);
        dump(prefix + ".syms", syms.data.data(), syms.pos);

        MemReader<unsigned char> sin; sin.data.assign(syms.data.begin(), syms.data.begin() + syms.pos);
        MemWriter<unsigned char> out(syms.pos + (1 << 16));
        std::vector<dh_event> events;
        // metadata through a PipelineMetaWriter (include/meta.hpp:42-46): the decoder's lines arrive in a Csdr::Writer the
        // way pycsdr / OpenWebRX consume them; a second, tiny writer shows that a line that does not fit is dropped whole
        auto* pipeMeta = new Digiham::PipelineMetaWriter(new Digiham::StringSerializer());
        MemWriter<unsigned char> metaOut(1 << 16);
        pipeMeta->setWriter(&metaOut);
        if (proto == "dmr") {
            Digiham::Dmr::Decoder dec;
            dec.setEventCallback([&](const dh_event& e) { events.push_back(e); });
            dec.setMetaWriter(pipeMeta);
            run(dec, sin, out, chunk / 4 + 1);
        } else {
            Digiham::Ysf::Decoder dec;
            dec.setEventCallback([&](const dh_event& e) { events.push_back(e); });
            dec.setMetaWriter(pipeMeta);
            run(dec, sin, out, chunk / 4 + 1);
        }
        dump(prefix + ".out", out.data.data(), out.pos);
        dump(prefix + ".events", events.data(), events.size());
        dump(prefix + ".meta", metaOut.data.data(), metaOut.pos);
        {
            Digiham::PipelineMetaWriter small(new Digiham::StringSerializer());
            MemWriter<unsigned char> tiny(16);
            small.sendMetaData({ { "protocol", "DMR" } });                                   // no writer attached yet: dropped
            small.setWriter(&tiny);
            small.sendMetaData({ { "protocol", "DMR" }, { "slot", "0" }, { "sync", "voice" } });   // 31 bytes > 16: dropped whole
            small.sendMetaData({ { "a", "b" } });                                             // "a:b\n" fits
            dump(prefix + ".smallmeta", tiny.data.data(), tiny.pos);
        }

        // digital voice filter on a synthetic s16 ramp of the same length class
        MemReader<short> vin; vin.data.resize(8000);
        for (size_t i = 0; i < vin.data.size(); i++) vin.data[i] = (short) ((i * 7919) % 40000 - 20000);
        Digiham::DigitalVoice::DigitalVoiceFilter dvf;
        MemWriter<short> vout(vin.data.size());
        run(dvf, vin, vout, 160);
        dump(prefix + ".dvin", vin.data.data(), vin.data.size());
        dump(prefix + ".dvout", vout.data.data(), vout.pos);
    } catch (const std::exception& e) {
        fprintf(stderr, "pipe_test: %s\n", e.what());
        return 1;
    }
    return 0;
}
