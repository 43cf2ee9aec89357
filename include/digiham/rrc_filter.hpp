// rrc_filter.hpp -- Digiham::RrcFilter::{RrcFilter, NarrowRrcFilter, WideRrcFilter} on the MI355X engine.
// Same class names, base class and constructor signatures as the reference's include/rrc_filter.hpp:10-31.
#pragma once

#include <cmath>
#include <memory>
#include <vector>

#include "csdr_compat.hpp"
#include "engine_handle.hpp"

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
                ~RrcFilter() override = default;
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
