// nxdn_meta.hpp -- Digiham::Nxdn::MetaCollector on decoder events.
// Reference: include/nxdn_meta.hpp, src/nxdn_decoder/nxdn_meta.cpp:5-76 (`protocol:NXDN;sync:voice;type:..;source:..;
// destination:..` lines) and its call sites in nxdn_phase.cpp (:51, :115-121, :141, :157); SACCH superframe fields
// sacch.cpp:133-152, constants types.hpp:1-9.  PARITY UNPINNED (see meta.hpp).
#pragma once

#include <string>

#include "meta.hpp"

#define NXDN_MESSAGE_TYPE_VCALL 0x01
#define NXDN_CALL_TYPE_CONFERENCE 0b001
#define NXDN_CALL_TYPE_INDIVIDUAL 0b100

namespace Digiham {
    namespace Nxdn {

        class MetaCollector: public Digiham::MetaCollector {
            public:
                void consume(const dh_event& ev) override {
                    switch (ev.type) {
                        case DH_EV_NXDN_SYNC_VOICE:                         // nxdn_phase.cpp:141
                            setSync("voice");
                            break;
                        case DH_EV_NXDN_SACCH_SF:                           // :115-121 -> setFromSacch (nxdn_meta.cpp:52-66)
                            if (ev.len < 9 || (ev.payload[0] & 0x3F) != NXDN_MESSAGE_TYPE_VCALL) break;
                            {
                                const unsigned callType = ev.payload[2] >> 5;
                                if (callType == NXDN_CALL_TYPE_CONFERENCE) setType("conference");
                                else if (callType == NXDN_CALL_TYPE_INDIVIDUAL) setType("individual");
                                else setType("");
                                setSource((uint16_t) ((ev.payload[3] << 8) | ev.payload[4]));
                                setDestination((uint16_t) ((ev.payload[5] << 8) | ev.payload[6]));
                            }
                            break;
                        case DH_EV_NXDN_META_RESET:                         // :51, :157 -> reset (nxdn_meta.cpp:68-75)
                            hold();
                            setSync(""); setType(""); setSource(0); setDestination(0);
                            release();
                            break;
                        default:
                            break;
                    }
                }
            protected:
                std::string getProtocol() override { return "NXDN"; }
                std::map<std::string, std::string> collect() override {     // nxdn_meta.cpp:7-22
                    auto metadata = Digiham::MetaCollector::collect();
                    if (!sync.empty()) metadata["sync"] = sync;
                    if (!type.empty()) metadata["type"] = type;
                    if (source != 0) metadata["source"] = std::to_string(source);
                    if (destination != 0) metadata["destination"] = std::to_string(destination);
                    return metadata;
                }
            private:
                void setSync(const std::string& v) { if (sync == v) return; sync = v; sendMetaData(); }
                void setType(const std::string& v) { if (type == v) return; type = v; sendMetaData(); }
                void setSource(uint16_t v) { if (source == v) return; source = v; sendMetaData(); }
                void setDestination(uint16_t v) { if (destination == v) return; destination = v; sendMetaData(); }
                std::string sync, type;
                uint16_t source = 0, destination = 0;
        };

    }
}
