// engine_handle.hpp -- RAII wrapper of a dh_engine for the single-channel operator classes.
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

#include "../digiham_amd.h"

namespace Digiham {
    namespace Amd {

        // every ABI failure is an exception -- DH_ECAPACITY too: a read that reports it has copied NOTHING and only
        // says how much room it wanted, so carrying on would hand stale bytes downstream.  The operator classes size
        // their pushes so that it cannot happen.
        inline void check(int rc, const char* what) {
            if (rc != DH_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + dh_last_error());
        }

        class Engine {
            public:
                Engine(int rrc, int demod, unsigned int sps, int proto, unsigned int flags, unsigned int maxSamples, unsigned int slotFilter = 3, unsigned int channels = 1) {
                    dh_engine_config cfg{};
                    cfg.struct_size = sizeof(cfg); cfg.device = 0; cfg.n_channels = channels; cfg.max_samples = maxSamples;
                    cfg.rrc = rrc; cfg.demod = demod; cfg.sps = sps; cfg.proto = proto; cfg.flags = flags; cfg.slot_filter = slotFilter;
                    cfg.stream = nullptr;
                    check(dh_engine_create(&cfg, &handle), "dh_engine_create");
                    max = maxSamples;
                }
                // a stand-alone FIR with the caller's table (DH_RRC_CUSTOM), filtered output kept
                Engine(const float* taps, unsigned int nZeros, double gain, unsigned int maxSamples) {
                    dh_engine_config cfg{};
                    cfg.struct_size = sizeof(cfg); cfg.device = 0; cfg.n_channels = 1; cfg.max_samples = maxSamples;
                    cfg.rrc = DH_RRC_CUSTOM; cfg.demod = DH_DEMOD_NONE; cfg.proto = DH_PROTO_NONE; cfg.flags = DH_FLAG_KEEP_FILTERED;
                    cfg.slot_filter = 3; cfg.rrc_taps = taps; cfg.rrc_nzeros = nZeros; cfg.rrc_gain = gain;
                    check(dh_engine_create(&cfg, &handle), "dh_engine_create");
                    max = maxSamples;
                }
                ~Engine() { dh_engine_destroy(handle); }
                Engine(const Engine&) = delete;
                Engine& operator=(const Engine&) = delete;
                dh_engine* get() { return handle; }
                unsigned int maxSamples() const { return max; }
            private:
                dh_engine* handle = nullptr;
                unsigned int max = 0;
        };

    }
}
