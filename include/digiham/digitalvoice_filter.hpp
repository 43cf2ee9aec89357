// digitalvoice_filter.hpp -- Digiham::DigitalVoice::DigitalVoiceFilter (reference: include/digitalvoice_filter.hpp:12-19).
#pragma once

#include "csdr_compat.hpp"
#include "engine_handle.hpp"

namespace Digiham {

    namespace DigitalVoice {

        class DigitalVoiceFilter: public Csdr::AnyLengthModule<short, short> {
            public:
                DigitalVoiceFilter() {
                    Amd::check(dh_device_alloc(0, sizeof(float) * 22, &state), "dh_device_alloc");
                    float zero[22] = { 0 };
                    Amd::check(dh_copy_to_device(state, zero, sizeof(zero)), "dh_copy_to_device");
                }
                ~DigitalVoiceFilter() override { dh_device_free(state); dh_device_free(dIn); dh_device_free(dOut); }
                void process(short* input, short* output, size_t length) override {
                    if (length > capacity) {
                        dh_device_free(dIn); dh_device_free(dOut);
                        capacity = length < 4096 ? 4096 : length;
                        Amd::check(dh_device_alloc(0, sizeof(short) * capacity, &dIn), "dh_device_alloc");
                        Amd::check(dh_device_alloc(0, sizeof(short) * capacity, &dOut), "dh_device_alloc");
                    }
                    Amd::check(dh_copy_to_device(dIn, input, sizeof(short) * length), "dh_copy_to_device");
                    Amd::check(dh_dvfilter_s16((const int16_t*) dIn, (int16_t*) dOut, (float*) state, 1, capacity, length, nullptr), "dh_dvfilter_s16");
                    Amd::check(dh_copy_to_host(output, dOut, sizeof(short) * length), "dh_copy_to_host");
                }
            private:
                void* state = nullptr;
                void* dIn = nullptr;
                void* dOut = nullptr;
                size_t capacity = 0;
        };

    }

}
