// rrc_filter.hpp -- Digiham::RrcFilter::{RrcFilter, NarrowRrcFilter, WideRrcFilter} on the MI355X engine.
// Same class names, base class and constructor signatures as the reference's include/rrc_filter.hpp:10-31.
#pragma once

#include <cmath>
#include <memory>
#include <vector>

#include "csdr_compat.hpp"
#include "engine_handle.hpp"
#include "shared_engine.hpp"

namespace Digiham {

    namespace RrcFilter {

        class RrcFilter: public Csdr::AnyLengthModule<float, float> {
            public:
                // Any coefficient table, as in the reference (include/rrc_filter.hpp:12): nZeros + 1 coefficients, applied
                // oldest sample first, sum / gain.  The reference itself only ever builds the two mkshape designs below
                // (rrc_filter.cpp:38-40, 86-88), which have their own tuned kernels; a foreign table runs on the engine's
                // generic FIR (exact arithmetic, up to 161 taps).
                RrcFilter(unsigned int nZeros, double gain, const float coeffs[]): kind(DH_RRC_CUSTOM), gain(gain), taps(coeffs, coeffs + nZeros + 1) {
                    if (nZeros < 1 || nZeros > 160) throw std::invalid_argument("Digiham::RrcFilter: 2 to 161 coefficients");
                }
                ~RrcFilter() override { if (bank) bank->detach(slot); }
                // With Digiham::Amd::SharedEngine::enable() every instance of a process deposits into one engine (one launch per
                // round for all of them, shared_engine.hpp); its output then arrives one call later.
                bool canProcess() override {
                    std::lock_guard<std::mutex> lock(this->processMutex);
                    if (!shared()) return std::min(this->reader->available(), this->writer->writeable()) > 0;
                    const bool pending = bank->hasPending(slot);
                    const size_t queued = bank->outputSize(slot);
                    return pending || (queued > 0 && this->writer->writeable() > 0) || (queued == 0 && this->reader->available() > 0);
                }
                void process() override {
                    std::lock_guard<std::mutex> lock(this->processMutex);
                    if (!shared()) {
                        size_t n = std::min(this->reader->available(), this->writer->writeable());
                        process(this->reader->getReadPointer(), this->writer->getWritePointer(), n);
                        this->reader->advance(n);
                        this->writer->advance(n);
                        return;
                    }
                    deliver();
                    if (bank->hasPending(slot)) { bank->settle(slot); deliver(); return; }
                    if (bank->outputSize(slot) > 0) return;                       // the writer is full: nothing new until it has taken what is there
                    const size_t n = std::min(this->reader->available(), Amd::SharedEngine::chunk);
                    if (n == 0) return;
                    bank->deposit(slot, this->reader->getReadPointer(), n);
                    this->reader->advance(n);
                    deliver();
                }
                void process(float* input, float* output, size_t length) override {
                    if (!engine) engine.reset(kind == DH_RRC_CUSTOM ? new Amd::Engine(taps.data(), (unsigned int) taps.size() - 1, gain, chunk)
                                                                   : new Amd::Engine(kind, DH_DEMOD_NONE, 0, DH_PROTO_NONE, DH_FLAG_KEEP_FILTERED, chunk));
                    while (length > 0) {
                        size_t n = length < chunk ? length : chunk;
                        Amd::check(dh_engine_push_host(engine->get(), input, n, n), "dh_engine_push_host");
                        size_t got = n;
                        Amd::check(dh_engine_read_filtered(engine->get(), 0, output, &got), "dh_engine_read_filtered");
                        input += n; output += n; length -= n;
                    }
                }
            protected:
                explicit RrcFilter(int kind): kind(kind) {}
            private:
                bool shared() {
                    if (!decided) {
                        decided = true;
                        if (kind != DH_RRC_CUSTOM && Amd::SharedEngine::enabled()) {
                            bank = Amd::SharedEngine::join(Amd::SharedEngine::RRC, kind, DH_DEMOD_NONE, 0, DH_PROTO_NONE, DH_FLAG_KEEP_FILTERED, slot);
                        }
                    }
                    return (bool) bank;
                }
                void deliver() {
                    const size_t n = bank->take(slot, this->writer->getWritePointer(), this->writer->writeable());
                    if (n) this->writer->advance(n);
                }
                std::shared_ptr<Amd::SharedEngine> bank;
                int slot = -1;
                bool decided = false;
                static constexpr size_t chunk = 65536;
                int kind;
                double gain = 0.0;
                std::vector<float> taps;
                std::unique_ptr<Amd::Engine> engine;
        };

        class NarrowRrcFilter: public RrcFilter {
            public:
                NarrowRrcFilter(): RrcFilter(DH_RRC_NARROW) {}
        };

        class WideRrcFilter: public RrcFilter {
            public:
                WideRrcFilter(): RrcFilter(DH_RRC_WIDE) {}
        };

    }

}
