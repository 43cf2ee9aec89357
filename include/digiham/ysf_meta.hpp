// ysf_meta.hpp -- Digiham::Ysf::MetaCollector on decoder events.
// Reference: include/ysf_meta.hpp, src/ysf_decoder/ysf_meta.cpp:13-106 (`protocol:YSF;mode:..;source:..` lines) and
// the call sites in ysf_phase.cpp (:50, :73, :87, :112, :133, :139-157, :163, :270-287, :351-361).
// `radio` never appears: the reference has a setter for it but no caller (ysf_phase.cpp never calls setRadio).
// PARITY UNPINNED (see meta.hpp).
#pragma once

#include <cstring>
#include <string>

#include "meta.hpp"

namespace Digiham {
    namespace Ysf {

        // short GPS position of the V/D2 data frames (gps.cpp:5-82).  The reference leaves `lon` uninitialised when
        // data[4] & 0xF0 is neither 0x50 nor 0x30 (:37-52); it starts from 0 here.
        struct Gps {
            static Coordinate* parse(const uint8_t* data) {
                for (int i = 0; i < 6; i++) if ((data[i] & 0x0F) > 9) return nullptr;
                float lat = (data[0] & 0x0F) * 10 + (data[1] & 0x0F) + (float) (data[2] & 0x0F) / 6 + (float) (data[3] & 0x0F) / 60 +
                            (float) (data[4] & 0x0F) / 600 + (float) (data[5] & 0x0F) / 6000;
                uint8_t direction = data[3] & 0xF0;
                if (direction == 0x50) {} else if (direction == 0x30) lat *= -1; else return nullptr;
                float lon = 0;
                uint8_t b = data[4] & 0xF0;
                const uint8_t c = data[6];
                if (b == 0x50) {
                    if (c >= 0x76 && c < 0x7f) lon = c - 0x76;
                    else if (c >= 0x6c && c < 0x75) lon = 100 + (c - 0x6c);
                    else if (c >= 0x26 && c < 0x6b) lon = 110 + (c - 0x26);
                    else return nullptr;
                } else if (b == 0x30) {
                    if (c >= 0x26 && c < 0x7f) lon = 10 + (c - 0x26);
                    else return nullptr;
                }
                b = data[7];
                if (b > 0x58 && b <= 0x61) lon += (float) (b - 0x58) / 60;
                else if (b >= 0x26 && b <= 0x57) lon += (float) (10 + (b - 0x26)) / 60;
                else return nullptr;
                b = data[8];
                if (b >= 0x1c && b < 0x7f) lon += (float) (b - 0x1c) / 6000;
                else return nullptr;
                direction = data[5] & 0xF0;
                if (direction == 0x50) lon *= -1; else if (direction == 0x30) {} else return nullptr;
                if (lat > 90 || lat < -90) return nullptr;
                if (lon > 180 || lon < -180) return nullptr;
                return new Coordinate(lat, lon);
            }
        };

        class MetaCollector: public Digiham::MetaCollector {
            public:
                ~MetaCollector() override { delete coord; }
                void consume(const dh_event& ev) override {
                    if (inHeader && ev.type != DH_EV_YSF_HEADER_DCH) endHeader();
                    switch (ev.type) {
                        case DH_EV_YSF_MODE: {                  // :73, :87, :112, :133 (FICH data type)
                            static const char* names[4] = { "V1", "FR data", "DN", "VW" };
                            setField(mode, names[ev.b & 3]);
                            break;
                        }
                        case DH_EV_YSF_DCH:                     // decodeV2DataChannel, :270-305 (a = frame number)
                            if (ev.len < 10) break;
                            if (ev.a < 6) {
                                switch (ev.a) {
                                    case 0: setField(destination, treatYsfString((const char*) ev.payload)); break;
                                    case 1: setField(source, treatYsfString((const char*) ev.payload)); break;
                                    case 2: setField(down, treatYsfString((const char*) ev.payload)); break;
                                    case 3: setField(up, treatYsfString((const char*) ev.payload)); break;
                                    default: break;
                                }
                                nextOffset = 0;                                 // dataCollector->reset()
                            } else if (ev.a < 8) {                              // DataCollector::collect (data.cpp:54-66)
                                const unsigned offset = ev.a - 6u;
                                if (offset != nextOffset) nextOffset = 0;
                                else { nextOffset = offset + 1; std::memcpy(dt + offset * 10, ev.payload, 10); }
                            }
                            if (nextOffset >= 2 && dt[18] == 0x03) {            // getDataFrame (data.cpp:72-87)
                                uint8_t checksum = 0;
                                for (int i = 0; i < 19; i++) checksum = (uint8_t) (checksum + dt[i]);
                                if (checksum == dt[19]) {
                                    const uint32_t command = (uint32_t) dt[1] << 16 | (uint32_t) dt[2] << 8 | dt[3];
                                    setGps(command == 0x22625fu ? Gps::parse(dt + 5) : nullptr);      // COMMAND_SHORT_GPS, data.cpp:28-35
                                }
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
                    if (coord != nullptr) {
                        result["lat"] = std::to_string(coord->lat);
                        result["lon"] = std::to_string(coord->lon);
                    }
                    return result;
                }
            private:
                void endHeader() { inHeader = false; release(); }
                void reset() {                                                  // ysf_meta.cpp:47-57
                    hold();
                    setField(mode, ""); setField(destination, ""); setField(source, ""); setField(up, ""); setField(down, "");
                    setGps(nullptr);
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
                void setGps(Coordinate* c) {                                    // ysf_meta.cpp:95-106
                    if (coord == c || (coord != nullptr && c != nullptr && *coord == *c)) { delete c; return; }
                    auto old = coord; coord = c; delete old;
                    sendMetaData();
                }
                std::string mode, destination, source, up, down;
                Coordinate* coord = nullptr;
                uint8_t dt[20] = { 0 };
                unsigned nextOffset = 0;
                bool inHeader = false;
        };

    }
}
