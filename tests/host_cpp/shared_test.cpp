// shared_test.cpp -- N x (WideRrcFilter | GfskDemodulator | Dmr::Decoder) in ONE process on shared engines
// (include/digiham/shared_engine.hpp): the three stages of every channel are connected by ring buffers and driven round-robin by
// one thread, the way a receiver hosting many channels runs them.  Prints how many launches (ticks) the banks needed and dumps
// every channel's decoder bytes and metadata lines.
//   shared_test <n_channels> <in.f32 (rows of equal length)> <samples_per_row> <out_prefix> <feed_chunk> <shared 0|1> [meta_every]
// meta_every = k > 1: only every k-th decoder gets a meta writer (a bank that mixes instances with and without one).  A driver
// that still finds work after 200 000 rounds has met a livelock and says so (exit code 4).
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "digiham/rrc_filter.hpp"
#include "digiham/gfsk_demodulator.hpp"
#include "digiham/dmr_decoder.hpp"

template <typename T> struct MemWriter: Csdr::Writer<T> {
    std::vector<T> data; size_t pos = 0;
    explicit MemWriter(size_t cap): data(cap) {}
    size_t writeable() override { return data.size() - pos; }
    T* getWritePointer() override { return data.data() + pos; }
    void advance(size_t n) override { pos += n; }
};

struct Channel {
    Csdr::Ringbuffer<float> in { 1 << 16 };
    Csdr::RingbufferReader<float> inR { &in };
    Csdr::Ringbuffer<float> filtered { 1 << 16 };
    Csdr::RingbufferReader<float> filteredR { &filtered };
    Csdr::Ringbuffer<unsigned char> syms { 1 << 15 };
    Csdr::RingbufferReader<unsigned char> symsR { &syms };
    MemWriter<unsigned char> out { 1 << 20 };
    Digiham::RrcFilter::WideRrcFilter rrc;
    Digiham::Fsk::GfskDemodulator gfsk { 10 };
    Digiham::Dmr::Decoder dec;
    std::string metaPath;
    Channel(const std::string& metaPath, int slotFilter, bool withMeta): metaPath(metaPath) {
        rrc.setReader(&inR); rrc.setWriter(&filtered);
        gfsk.setReader(&filteredR); gfsk.setWriter(&syms);
        dec.setReader(&symsR); dec.setWriter(&out);
        if (withMeta) dec.setMetaWriter(new Digiham::FileMetaWriter(fopen(metaPath.c_str(), "w")));
        else fclose(fopen(metaPath.c_str(), "w"));
        if (slotFilter != 3) dec.setSlotFilter((unsigned char) slotFilter);
    }
};

int main(int argc, char** argv) {
    if (argc < 7) return 2;
    const int N = atoi(argv[1]);
    const size_t T = (size_t) atol(argv[3]), feed = (size_t) atol(argv[5]);
    const std::string prefix = argv[4];
    if (atoi(argv[6])) Digiham::Amd::SharedEngine::enable((unsigned int) N);
    const int metaEvery = argc > 7 ? atoi(argv[7]) : 1;
    std::vector<float> x((size_t) N * T);
    FILE* f = fopen(argv[2], "rb");
    if (!f || fread(x.data(), sizeof(float), x.size(), f) != x.size()) return 3;
    fclose(f);
    try {
        std::vector<std::unique_ptr<Channel>> ch;
        for (int i = 0; i < N; i++) ch.emplace_back(new Channel(prefix + "." + std::to_string(i) + ".meta", i % 5 == 4 ? 1 : 3, metaEvery <= 1 || i % metaEvery == 0));
        std::vector<size_t> fed(N, 0);
        unsigned long rounds = 0;
        for (;;) {
            bool busy = false;
            for (int i = 0; i < N; i++) {            // the source: ragged on purpose (channel i gets feed + 37 i samples per round)
                const size_t want = std::min(T - fed[i], feed + 37 * (size_t) i);
                const size_t n = std::min(want, ch[i]->in.writeable());
                if (n) { memcpy(ch[i]->in.getWritePointer(), x.data() + (size_t) i * T + fed[i], n * sizeof(float)); ch[i]->in.advance(n); fed[i] += n; busy = true; }
            }
            // one round: every module of every channel gets one call if it can work (stage by stage, as worker threads would interleave)
            for (int i = 0; i < N; i++) if (ch[i]->rrc.canProcess()) { ch[i]->rrc.process(); busy = true; }
            for (int i = 0; i < N; i++) if (ch[i]->gfsk.canProcess()) { ch[i]->gfsk.process(); busy = true; }
            for (int i = 0; i < N; i++) if (ch[i]->dec.canProcess()) { ch[i]->dec.process(); busy = true; }
            rounds++;
            if (!busy) break;
            if (rounds > 200000) { fprintf(stderr, "shared_test: still busy after %lu rounds (livelock)\n", rounds); return 4; }
        }
        printf("rounds %lu ticks %lu\n", rounds, Digiham::Amd::SharedEngine::ticksTotal());
        for (int i = 0; i < N; i++) {
            FILE* o = fopen((prefix + "." + std::to_string(i) + ".out").c_str(), "wb");
            fwrite(ch[i]->out.data.data(), 1, ch[i]->out.pos, o);
            fclose(o);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "shared_test: %s\n", e.what());
        return 1;
    }
    return 0;
}
