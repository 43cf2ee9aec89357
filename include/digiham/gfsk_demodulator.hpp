// gfsk_demodulator.hpp / fsk_demodulator.hpp share this implementation: Digiham::Fsk::GfskDemodulator and
// Digiham::Fsk::FskDemodulator with the reference's constructor signatures
// (include/gfsk_demodulator.hpp:12-34, include/fsk_demodulator.hpp:12-33).
//
// The reference slices ONE symbol per process() call; here a call hands everything the reader holds to the
// engine (which keeps the not-yet-consumable tail itself) and writes every symbol that became available.
// The symbol stream is identical; only the call granularity differs.
#pragma once

#include <memory>
#include <vector>

#include "csdr_compat.hpp"
#include "engine_handle.hpp"
#include "shared_engine.hpp"

namespace Digiham {

    namespace Fsk {

        class SlicerBase: public Csdr::Module<float, unsigned char> {
            public:
                ~SlicerBase() override { if (bank) bank->detach(slot); }
                bool canProcess() override {
                    std::lock_guard<std::mutex> lock(processMutex);
                    // +1 for variance calculation "jumps" (gfsk_demodulator.cpp:18-22)
                    const bool fresh = reader->available() > samplesPerSymbol + 1 && writer->writeable() >= minRoom;
                    if (!shared()) return fresh;
                    const size_t queued = bank->outputSize(slot);
                    return bank->hasPending(slot) || (queued > 0 && writer->writeable() > 0) || (queued == 0 && fresh);
                }
                void process() override {
                    std::lock_guard<std::mutex> lock(processMutex);
                    if (shared()) {                    // one engine for every instance of the process (shared_engine.hpp)
                        deliver();
                        if (bank->hasPending(slot)) { bank->settle(slot); deliver(); return; }
                        if (bank->outputSize(slot) > 0) return;
                        if (!(reader->available() > samplesPerSymbol + 1)) return;
                        const size_t n = std::min(reader->available(), Amd::SharedEngine::chunk);
                        bank->deposit(slot, reader->getReadPointer(), n);
                        reader->advance(n);
                        deliver();
                        return;
                    }
                    if (!engine) engine.reset(new Amd::Engine(DH_RRC_NONE, levels, samplesPerSymbol, DH_PROTO_NONE, invert ? DH_FLAG_FSK_INVERT : 0, chunk));
                    // never produce more symbols than the writer can take: a symbol needs at least sps - 1 samples, and the
                    // engine may still hold up to sps + 1 samples of the previous call: n samples give at most
                    // (n + sps + 1) / (sps - 1) + 1 <= n / (sps - 1) + 3 symbols
                    size_t n = reader->available();
                    size_t room = writer->writeable();
                    size_t cap = room > 3 ? (room - 3) * (samplesPerSymbol - 1) : 0;      // (a call without canProcess(): nothing fits)
                    if (n > cap) n = cap;
                    if (n > chunk) n = chunk;
                    if (n == 0) return;
                    Amd::check(dh_engine_push_host(engine->get(), reader->getReadPointer(), n, n), "dh_engine_push_host");
                    reader->advance(n);
                    size_t got = room;
                    Amd::check(dh_engine_read_symbols(engine->get(), 0, writer->getWritePointer(), &got), "dh_engine_read_symbols");
                    writer->advance(got);
                }
            protected:
                SlicerBase(int levels, unsigned int samplesPerSymbol, bool invert): levels(levels), samplesPerSymbol(samplesPerSymbol), invert(invert) {}
            private:
                bool shared() {
                    if (!decided) {
                        decided = true;
                        if (Amd::SharedEngine::enabled()) {
                            bank = Amd::SharedEngine::join(Amd::SharedEngine::SLICER, DH_RRC_NONE, levels, samplesPerSymbol, DH_PROTO_NONE, invert ? DH_FLAG_FSK_INVERT : 0, slot);
                        }
                    }
                    return (bool) bank;
                }
                void deliver() {
                    const size_t n = bank->take(slot, writer->getWritePointer(), writer->writeable());
                    if (n) writer->advance(n);
                }
                std::shared_ptr<Amd::SharedEngine> bank;
                int slot = -1;
                bool decided = false;
                static constexpr size_t chunk = 65536;
                static constexpr size_t minRoom = 4;
                int levels;
                unsigned int samplesPerSymbol;
                bool invert;
                std::unique_ptr<Amd::Engine> engine;
        };

        class GfskDemodulator: public SlicerBase {
            public:
                explicit GfskDemodulator(unsigned int samplesPerSymbol): SlicerBase(DH_DEMOD_GFSK4, samplesPerSymbol, false) {}
        };

        class FskDemodulator: public SlicerBase {
            public:
                explicit FskDemodulator(unsigned int samplesPerSymbol, bool invert = false): SlicerBase(DH_DEMOD_FSK2, samplesPerSymbol, invert) {}
        };

    }

}
