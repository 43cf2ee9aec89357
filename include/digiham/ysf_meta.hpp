// ysf_meta.hpp -- Digiham::Ysf::MetaCollector on decoder events.
// Reference: include/ysf_meta.hpp, src/ysf_decoder/ysf_meta.cpp:13-106 (`protocol:YSF;mode:..;source:..` lines) and
// the call sites in ysf_phase.cpp (:50, :73, :87, :112, :133, :139-157, :163, :270-287, :351-361).
// Not carried over: the GPS position of the V/D2 data frames 6/7 (data.cpp / gps.cpp, :289-303) and the radio
// model name -- `lat`, `lon` and `radio` never appear.  PARITY UNPINNED (see meta.hpp).
#pragma once

#include <cstring>
#include <string>

#include "meta.hpp"

namespace Digiham {
    namespace Ysf {

        class MetaCollector: public Digiham::MetaCollector {
            public:
                void consume(const dh_event& ev) override {
                    if (inHeader && ev.type != DH_EV_YSF_HEADER_DCH) endHeader();
                    switch (ev.type) {
                        case DH_EV_YSF_MODE: {                  // :73, :87, :112, :133 (FICH data type)
                            static const char* names[4] = { "V1", "FR data", "DN", "VW" };
                            setField(mode, names[ev.b & 3]);
                            break;
                        }
                        case DH_EV_YSF_DCH:                     // decodeV2DataChannel, :270-287 (a = frame number)
                            if (ev.len < 10) break;
                            switch (ev.a) {
                                case 0: setField(destination, treatYsfString((const char*) ev.payload)); break;
                                case 1: setField(source, treatYsfString((const char*) ev.payload)); break;
                                case 2: setField(down, treatYsfString((const char*) ev.payload)); break;
                                case 3: setField(up, treatYsfString((const char*) ev.payload)); break;
                                default: break;
                            }
                            break;
                        case DH_EV_YSF_META_RESET:
                            reset();                            // :50 (sync lost), :141 (header), :163 (terminator)
                            if (ev.b == 1) { hold(); inHeader = true; }      // header: the CSD fields go out as one line (:142-157)
                            break;
                        case DH_EV_YSF_HEADER_DCH:              // a = 0: CSD1 (dest, src), a = 1: CSD2 (down, up)
                            if (ev.len < 20) break;
                            if (ev.a == 0) {
                                setField(destination, treatYsfString((const char*) ev.payload));
                                setField(source, treatYsfString((const char*) ev.payload + 10));
                            } else {
                                setField(down, treatYsfString((const char*) ev.payload));
                                setField(up, treatYsfString((const char*) ev.payload + 10));
                            }
                            break;
                        default:
                            break;
                    }
                }
                void flush() override { if (inHeader) endHeader(); }
            protected:
                std::string getProtocol() override { return "YSF"; }
                std::map<std::string, std::string> collect() override {         // ysf_meta.cpp:13-45
                    auto result = Digiham::MetaCollector::collect();
                    if (!mode.empty()) result["mode"] = mode;
                    if (!destination.empty()) result["target"] = destination;
                    if (!source.empty()) result["source"] = source;
                    if (!up.empty()) result["up"] = up;
                    if (!down.empty()) result["down"] = down;
                    return result;
                }
            private:
                void endHeader() { inHeader = false; release(); }
                void reset() {                                                  // ysf_meta.cpp:47-57
                    hold();
                    setField(mode, ""); setField(destination, ""); setField(source, ""); setField(up, ""); setField(down, "");
                    release();
                }
                void setField(std::string& field, const std::string& value) {   // ysf_meta.cpp:59-93
                    if (field == value) return;
                    field = value;
                    sendMetaData();
                }
                static std::string treatYsfString(const char* input) {          // ysf_phase.cpp:351-361
                    size_t length = 10;
                    for (char c : { '\n', ' ' }) {
                        const char* end = (const char*) memchr(input, c, length);
                        if (end != nullptr) length = (size_t) (end - input);
                    }
                    return Converter::convertToUtf8(input, length);
                }
                std::string mode, destination, source, up, down;
                bool inHeader = false;
        };

    }
}
