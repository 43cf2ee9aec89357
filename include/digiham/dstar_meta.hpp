// dstar_meta.hpp -- Digiham::DStar::MetaCollector on decoder events.
// Reference: include/dstar_meta.hpp, src/dstar_decoder/dstar_meta.cpp:9-135 (`protocol:DSTAR;sync:voice;departure:..;
// destination:..;ourcall:..;yourcall:..;message:..;dprs:..;lat:..;lon:..` lines), the header fields header.cpp:150-181,
// and the slow-data text handling the reference does inside its VoicePhase (dstar_phase.cpp:205-290): the simple-data
// string, `$$CRC` DPRS sentences and NMEA GGA positions are host work here too, driven by the events the frame
// parser on the GPU emits in the reference's call order.  PARITY UNPINNED (see meta.hpp).
#pragma once

#include <cstdlib>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "meta.hpp"

namespace Digiham {
    namespace DStar {

        // Crc::isCrcValid (src/dstar_decoder/crc.cpp:6-23)
        inline bool isCrcValid(const unsigned char* data, size_t len, uint16_t to_check) {
            uint16_t checksum = 0xFFFF;
            for (size_t k = 0; k < len; k++) {
                for (int i = 0; i < 8; i++) {
                    checksum ^= (data[k] >> i) & 1;
                    if (checksum & 1) checksum = (uint16_t) ((checksum >> 1) ^ 0x8408);
                    else checksum >>= 1;
                }
            }
            checksum ^= 0xFFFF;
            return checksum == to_check;
        }

        class MetaCollector: public Digiham::MetaCollector {
            public:
                ~MetaCollector() override { delete coord; }
                void consume(const dh_event& ev) override {
                    switch (ev.type) {
                        case DH_EV_DSTAR_HEADER:                            // dstar_phase.cpp:50 (radio header), :214 (slow data)
                            if (ev.a == 0) { std::memcpy(header, ev.payload, 24); break; }
                            std::memcpy(header + 24, ev.payload, 17);
                            setFromHeader();
                            break;
                        case DH_EV_DSTAR_VOICE_START:                       // a new VoicePhase: an empty simpleData (dstar_phase.hpp:77)
                            simpleData.clear();
                            break;
                        case DH_EV_DSTAR_SYNC_VOICE:                        // :110
                            setSync("voice");
                            break;
                        case DH_EV_DSTAR_MESSAGE:                           // :207
                            setMessage(Converter::convertToUtf8((const char*) ev.payload, 20));
                            break;
                        case DH_EV_DSTAR_SIMPLE:                            // :178
                            simpleData += std::string((const char*) ev.payload, ev.len);
                            break;
                        case DH_EV_DSTAR_FRAME_SYNC:                        // parseFrameData's sentence loop (:218-245)
                            parseSimpleData();
                            break;
                        case DH_EV_DSTAR_META_RESET:                        // :97, :105 -> reset (dstar_meta.cpp:83-94)
                            hold();
                            setSync(""); setMessage(""); setDeparture(""); setDestination(""); setOurCall(""); setYourCall("");
                            setDPRS(""); setGPS(nullptr);
                            release();
                            break;
                        default:
                            break;
                    }
                }
            protected:
                std::string getProtocol() override { return "DSTAR"; }
                std::map<std::string, std::string> collect() override {     // dstar_meta.cpp:100-135
                    auto metadata = Digiham::MetaCollector::collect();
                    if (!sync.empty()) metadata["sync"] = sync;
                    if (!departure.empty()) metadata["departure"] = departure;
                    if (!destination.empty()) metadata["destination"] = destination;
                    if (!ourCall.empty()) metadata["ourcall"] = ourCall;
                    if (!yourCall.empty()) metadata["yourcall"] = yourCall;
                    if (!message.empty()) metadata["message"] = message;
                    if (!dprs.empty()) metadata["dprs"] = dprs;
                    if (coord != nullptr) {
                        metadata["lat"] = std::to_string(coord->lat);
                        metadata["lon"] = std::to_string(coord->lon);
                    }
                    return metadata;
                }
            private:
                static std::string rtrim(std::string input) {               // header.cpp:154-157
                    input.erase(input.find_last_not_of(' ') + 1);
                    return input;
                }
                std::string field(int at, int len) const { return rtrim(Converter::convertToUtf8((const char*) header + at, (size_t) len)); }
                void setFromHeader() {                                      // dstar_meta.cpp:15-27, header.cpp:150-181
                    hold();
                    setSync(((header[0] >> 7) & 1) ? "data" : "voice");
                    setDeparture(field(11, 8));
                    setDestination(field(3, 8));
                    std::string own = field(27, 8);
                    const std::string suffix = field(35, 4);
                    if (suffix != "") own += "/" + suffix;
                    setOurCall(own);
                    setYourCall(field(19, 8));
                    release();
                }
                // `ss << std::hex << text; ss >> value` (dstar_phase.cpp:222-225, :262-265): leading hex digits, 0 when none
                static uint16_t hexValue(const std::string& text) {
                    std::stringstream ss;
                    uint16_t v = 0;
                    ss << std::hex << text;
                    ss >> v;
                    return ss.fail() ? (uint16_t) 0 : v;
                }
                void parseSimpleData() {
                    size_t pos;
                    while ((pos = simpleData.find('\r')) != std::string::npos) {
                        const std::string something = simpleData.substr(0, pos + 1);
                        if (something.length() >= 10 && something.substr(0, 5) == "$$CRC" && something.at(9) == ',') {
                            const uint16_t checksum = hexValue(something.substr(5, 4));
                            const std::string body = something.substr(10);
                            if (isCrcValid((const unsigned char*) body.c_str(), something.length() - 10, checksum))
                                setDPRS(something.substr(10, something.length() - 11));
                        } else if (something.length() > 5 && something.at(0) == '$') {
                            parseNMEAData(something);
                        }
                        // termination may be \r or \r\n (:243-244)
                        simpleData = simpleData.substr(pos + 1 + (simpleData.length() > pos + 1 && simpleData.at(pos + 1) == '\n'));
                    }
                }
                // dstar_phase.cpp:248-290.  Where the reference would throw out of a malformed sentence (substr / stof on
                // short or empty fields, :257, :281-286) and end the process, the sentence is dropped instead.
                void parseNMEAData(const std::string& input) {
                    const size_t checksum_pos = input.find_last_of("*");
                    if (checksum_pos == std::string::npos) return;
                    if (checksum_pos + 2 > input.length()) return;
                    if (checksum_pos < 1) return;
                    const std::string body = input.substr(1, checksum_pos - 1);
                    if (body.length() < 2) return;
                    const std::string sentence = body.substr(2, 3);
                    uint8_t checksum = 0;
                    for (size_t i = 0; i < body.length(); i++) checksum ^= (uint8_t) body.at(i);
                    if (checksum != hexValue(input.substr(checksum_pos + 1, 2))) return;
                    std::vector<std::string> fields;
                    std::stringstream splitter(body);
                    std::string item;
                    while (getline(splitter, item, ',')) fields.push_back(item);
                    if (sentence == "GGA") {
                        if (fields.size() < 6) return;
                        float lat_combined, lon_combined;
                        try { lat_combined = std::stof(fields[2]); lon_combined = std::stof(fields[4]); }
                        catch (const std::exception&) { return; }
                        float lat = (int) lat_combined / 100;
                        lat += (lat_combined - lat * 100) / 60;
                        if (fields[3] == "S") lat *= -1;
                        float lon = (int) lon_combined / 100;
                        lon += (lon_combined - lon * 100) / 60;
                        if (fields[5] == "W") lon *= -1;
                        setGPS(new Coordinate(lat, lon));
                    }
                }
                void setSync(const std::string& v) { if (sync == v) return; sync = v; sendMetaData(); }
                void setMessage(const std::string& v) { if (message == v) return; message = v; sendMetaData(); }
                void setDeparture(const std::string& v) { if (departure == v) return; departure = v; sendMetaData(); }
                void setDestination(const std::string& v) { if (destination == v) return; destination = v; sendMetaData(); }
                void setOurCall(const std::string& v) { if (ourCall == v) return; ourCall = v; sendMetaData(); }
                void setYourCall(const std::string& v) { if (yourCall == v) return; yourCall = v; sendMetaData(); }
                void setDPRS(const std::string& v) { if (dprs == v) return; dprs = v; sendMetaData(); }
                void setGPS(Coordinate* c) {                                // dstar_meta.cpp:71-81
                    if (coord == c || (coord != nullptr && c != nullptr && *coord == *c)) { delete c; return; }
                    auto old = coord; coord = c; delete old;
                    sendMetaData();
                }
                unsigned char header[41] = { 0 };
                std::string simpleData;
                std::string sync, departure, destination, ourCall, yourCall, message, dprs;
                Coordinate* coord = nullptr;
        };

    }
}
