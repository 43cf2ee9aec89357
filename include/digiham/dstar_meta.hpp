// dstar_meta.hpp -- Digiham::DStar::MetaCollector on decoder events.
// Reference: include/dstar_meta.hpp, src/dstar_decoder/dstar_meta.cpp:9-135 (`protocol:DSTAR;sync:voice;departure:..;
// destination:..;ourcall:..;yourcall:..;message:..;dprs:..;lat:..;lon:..` lines), the header fields header.cpp:150-181,
// and the slow-data text handling the reference does inside its VoicePhase (dstar_phase.cpp:205-290): the simple-data
// string, `$$CRC` DPRS sentences and NMEA GGA positions are host work here too, driven by the events the frame
// parser on the GPU emits in the reference's call order.  PARITY UNPINNED (see meta.hpp).
#pragma once

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <string>

#include "meta.hpp"

namespace Digiham {
    namespace DStar {

        // The D-Star checksum (reflected CRC-CCITT, polynomial 0x8408, preset and final inversion 0xFFFF) of `len` bytes against
        // `expected`; same value as Crc::isCrcValid (src/dstar_decoder/crc.cpp:6-23), which feeds the register bit by bit.
        inline bool isCrcValid(const unsigned char* data, size_t len, uint16_t expected) {
            unsigned reg = 0xFFFF;
            while (len--) {
                reg ^= *data++;
                for (int bit = 0; bit < 8; bit++) reg = (reg >> 1) ^ (0x8408u & (0u - (reg & 1u)));
            }
            return (uint16_t) ~reg == expected;
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
                    std::map<std::string, std::string> line = Digiham::MetaCollector::collect();
                    const std::pair<const char*, const std::string*> texts[] = {
                        {"sync", &sync}, {"departure", &departure}, {"destination", &destination}, {"ourcall", &ourCall},
                        {"yourcall", &yourCall}, {"message", &message}, {"dprs", &dprs}};
                    for (const auto& t: texts)
                        if (!t.second->empty()) line[t.first] = *t.second;
                    if (coord) {
                        line["lat"] = std::to_string(coord->lat);
                        line["lon"] = std::to_string(coord->lon);
                    }
                    return line;
                }
            private:
                static std::string rtrim(std::string text) {                // header.cpp:154-157: trailing blanks off
                    size_t keep = text.size();
                    while (keep > 0 && text[keep - 1] == ' ') keep--;
                    text.resize(keep);
                    return text;
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
                // ---- slow-data text (what the reference's VoicePhase does with strings and streams, dstar_phase.cpp:211-290) ----
                // Here: one pass over the received bytes with a read-only window; no streams, nothing thrown.

                // A window on received text.
                struct Text {
                    const char* p; size_t n;
                    char operator[](size_t i) const { return p[i]; }
                    Text from(size_t at) const { return at >= n ? Text{p + n, 0} : Text{p + at, n - at}; }
                    Text first(size_t len) const { return Text{p, len < n ? len : n}; }
                    bool is(const char* lit) const { return std::strlen(lit) == n && std::memcmp(p, lit, n) == 0; }
                    std::string str() const { return std::string(p, n); }
                };
                // What formatted extraction of a uint16_t in base 16 leaves in the variable (the reference reads both of its
                // checksums that way, :222-225 and :262-265): blanks skipped, a sign, zeros with at most one x among them, then
                // hex digits; nothing usable gives 0, too much gives 0xFFFF, a minus sign negates modulo 2^16.
                static uint16_t hex16(Text t) {
                    size_t i = 0;
                    while (i < t.n && std::strchr(" \t\n\v\f\r", t[i]) != nullptr && t[i] != 0) i++;
                    bool minus = false;
                    if (i < t.n && (t[i] == '+' || t[i] == '-')) minus = t[i++] == '-';
                    bool zero = false, x = false;
                    unsigned digits = 0;
                    for (; i < t.n; i++) {
                        if (t[i] == '0' && !zero) { zero = true; digits++; }
                        else if (zero && (t[i] == 'x' || t[i] == 'X') && !x) { zero = false; digits = 0; x = true; }
                        else break;
                    }
                    unsigned long v = 0;
                    for (; i < t.n; i++, digits++) {
                        const char c = t[i];
                        const int d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1;
                        if (d < 0) break;
                        if (v <= 0xFFFFFF) v = v * 16 + (unsigned) d;
                    }
                    if (digits == 0 && !zero) return 0;
                    if (v > 0xFFFF) return 0xFFFF;
                    return (uint16_t) (minus ? 0u - v : v);
                }
                // ddmm.mmmm -> degrees in the reference's float steps (:281-282, :285-286; the lines must agree to the last bit)
                static float degrees(float ddmm) {
                    float whole = (int) ddmm / 100;
                    whole += (ddmm - whole * 100) / 60;
                    return whole;
                }
                // A field as std::stof reads it (strtof's longest prefix); false where stof would have thrown, or where the
                // (int) of the value is not defined.
                static bool number(Text t, float& out) {
                    const std::string z = t.str();                                      // (strtof wants a terminator)
                    char* end = nullptr;
                    errno = 0;
                    out = std::strtof(z.c_str(), &end);
                    if (end == z.c_str() || errno == ERANGE) return false;
                    return out > -2147483648.0f && out < 2147483648.0f;                 // also refuses NaN
                }
                // Every complete sentence of the collected simple data (terminated by \r or \r\n), oldest first.
                void parseSimpleData() {
                    size_t done = 0;
                    for (;;) {
                        const void* cr = std::memchr(simpleData.data() + done, '\r', simpleData.size() - done);
                        if (cr == nullptr) break;
                        const size_t end = (size_t) ((const char*) cr - simpleData.data()) + 1;     // one past the \r
                        const Text s{simpleData.data() + done, end - done};
                        if (s.n >= 10 && s.first(5).is("$$CRC") && s[9] == ',') {
                            const Text covered = s.from(10);                            // the CRC covers the \r, the field does not
                            if (isCrcValid((const unsigned char*) covered.p, covered.n, hex16(s.from(5).first(4))))
                                setDPRS(covered.first(covered.n - 1).str());
                        } else if (s.n > 5 && s[0] == '$') {
                            nmea(s);
                        }
                        done = end + (end < simpleData.size() && simpleData[end] == '\n');
                    }
                    simpleData.erase(0, done);
                }
                // `$ttSSS,f1,f2,...*hh\r`: XOR of everything between $ and the LAST `*` against the two characters behind it; of all
                // sentences only GGA is used (fields 2..5 = latitude, N/S, longitude, E/W).  A sentence the reference could not
                // have survived (fields missing, no number where one belongs: an exception out of its process) is dropped.
                void nmea(Text s) {
                    size_t star = s.n;
                    while (star > 0 && s[star - 1] != '*') star--;
                    if (star == 0 || star + 1 > s.n) return;                            // no `*`, or nothing behind it
                    star--;
                    const Text body = s.from(1).first(star - 1);
                    if (body.n < 2) return;
                    uint8_t x = 0;
                    for (size_t i = 0; i < body.n; i++) x ^= (uint8_t) body[i];
                    if (x != hex16(s.from(star + 1).first(2))) return;
                    if (!body.from(2).first(3).is("GGA")) return;
                    Text f[6];
                    size_t count = 0, at = 0;
                    while (at < body.n) {                                               // (an empty field behind a trailing comma does not count)
                        size_t comma = at;
                        while (comma < body.n && body[comma] != ',') comma++;
                        if (count < 6) f[count] = body.from(at).first(comma - at);
                        count++;
                        at = comma + 1;
                    }
                    if (count < 6) return;
                    float la, lo;
                    if (!number(f[2], la) || !number(f[4], lo)) return;
                    const float lat = degrees(la), lon = degrees(lo);
                    setGPS(new Coordinate(f[3].is("S") ? -lat : lat, f[5].is("W") ? -lon : lon));
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
