// rrc_filter.hpp -- Digiham::RrcFilter::{RrcFilter, NarrowRrcFilter, WideRrcFilter} on the MI355X engine.
// Same class names, base class and constructor signatures as the reference's include/rrc_filter.hpp:10-31.
#pragma once

#include <cmath>
#include <memory>

#include "csdr_compat.hpp"
#include "engine_handle.hpp"

namespace Digiham {

    namespace RrcFilter {

        class RrcFilter: public Csdr::AnyLengthModule<float, float> {
            public:
                // The reference takes an arbitrary coefficient table; it only ever builds the two mkshape designs
                // below (rrc_filter.cpp:38-40, 86-88).  The engine carries exactly those two, so any other table is
                // rejected instead of being filtered with the wrong taps.
                RrcFilter(unsigned int nZeros, double gain, const float coeffs[]) {
                    (void) coeffs;
                    if (nZeros == 80 && std::fabs(gain - 8.337797030e+00) < 1e-9) kind = DH_RRC_WIDE;
                    else if (nZeros == 160 && std::fabs(gain - 1.667711971e+01) < 1e-9) kind = DH_RRC_NARROW;
                    else throw std::invalid_argument("Digiham::RrcFilter: only the wide (81-tap) and narrow (161-tap) designs are available");
                }
                ~RrcFilter() override = default;
                void process(float* input, float* output, size_t length) override {
                    if (!engine) engine.reset(new Amd::Engine(kind, DH_DEMOD_NONE, 0, DH_PROTO_NONE, DH_FLAG_KEEP_FILTERED, chunk));
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
