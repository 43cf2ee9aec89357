// dmr_meta.hpp -- DMR call metadata from decoder events: `protocol:DMR;slot:N;sync:..;type:..;source:..;target:..;
// talkeralias:..;lat:..;lon:..` lines, one per change and slot.
//
// What the reference does while it parses bursts (call sites dmr_phase.cpp:80, :109-114, :178-184, :196-202, :233,
// :279-282, :295, :304-339 into dmr_meta.cpp:11-179, talkeralias.cpp:20-144, gps.cpp:7-17) happens here after the
// fact: the GPU decoder reports those calls as dh_events, and Dmr::MetaCollector replays them on two slot records.
// A slot record is a FieldRecord (meta.hpp): seven text fields in wire order, one change flag; an event is a handful
// of put()s and at most one line.
//
// Pinned against the reference's own classes (tests/golden/elements_ref.npz via tests/host_cpp/elements_test.cpp):
// Gps::parse, TalkerAliasCollector (all four formats, partial and out-of-order blocks), Lc getters.  The event ->
// line rules are restated from the cited lines (PARITY UNPINNED: dmr_meta.cpp / dmr_phase.cpp need csdr).
#pragma once

#include <cmath>
#include <cstring>
#include <string>

#include "lc.hpp"
#include "meta.hpp"

#define SYNCTYPE_DATA 1
#define SYNCTYPE_VOICE 2

namespace Digiham {
    namespace Dmr {

        // GPS Info LC, bytes 2..8 of the LC word (Lc::getData()): longitude sign in bit 0 of byte 0, 24-bit magnitude in
        // bytes 1-3, steps of 360 / 2^25 degrees; latitude sign in bit 7 of byte 4, 23-bit magnitude below it, steps of
        // 180 / 2^24 degrees.  Both step sizes are exact binary fractions times 45, so magnitude * 45 scaled by a power
        // of two rounds once -- the same float as the reference's `180.0f / (1 << 24) * (float) bits`.
        struct Gps {
            static Coordinate* parse(const unsigned char* d) {
                const int32_t lat = (int32_t) (d[4] & 0x7F) << 16 | d[5] << 8 | d[6];
                const int32_t lon = (int32_t) d[1] << 16 | d[2] << 8 | d[3];
                return new Coordinate(std::ldexp((float) (45 * (d[4] & 0x80 ? -lat : lat)), -22),
                                      std::ldexp((float) ((int64_t) 45 * (d[0] & 0x01 ? -lon : lon)), -22));
            }
        };

        // Talker alias: up to four 7-byte LC payloads (header block + 3), 28 bytes read as ONE bit string.
        // Byte 0: format (2 bits), length in characters (5 bits), and the first payload bit.
        class TalkerAliasCollector {
            public:
                enum Format { Bits7 = 0, Latin1 = 1, Utf8 = 2, Utf16 = 3 };
                void reset() { have = 0; }
                void setBlock(int block, const unsigned char* payload) {
                    std::memcpy(bytes + 7 * block, payload, 7);
                    have |= 1u << block;
                }
                // enough consecutive blocks (from the header on) for the announced length?  The per-format arithmetic is the
                // reference's (talkeralias.cpp:32-51), including its 7-bit estimate (bytes * 7 / 8 - 1 characters)
                bool isComplete() {
                    if (!(have & 1u)) return false;
                    const int n = usable(), want = length();
                    switch (format()) {
                        case Bits7: return n * 7 / 8 - 1 >= want;
                        case Latin1: return n - 1 >= want;
                        case Utf8: return (int) getContents().size() >= want;
                        default: return (n - 1) / 2 >= want;
                    }
                }
                std::string getContents() {
                    if (!(have & 1u)) return std::string();
                    const int n = usable();
                    std::string text;
                    switch (format()) {
                        case Bits7:                            // n * 8 / 7 seven-bit characters, MSB first; the first one is header bits
                            for (int bit = 7; bit + 7 <= 8 * n; bit += 7) text += (char) take(bit, 7);
                            break;
                        case Latin1: text = Converter::convertToUtf8((const char*) bytes + 1, (size_t) n - 1); break;
                        case Utf8: text.assign((const char*) bytes + 1, (size_t) n - 1); break;
                        default:                               // UTF-16BE code units; a surrogate pair is one character
                            for (int at = 1; at + 2 <= n; at += 2) {
                                uint32_t cp = take(8 * at, 16);
                                if ((cp & 0xFC00) == 0xD800 && at + 4 <= n && (take(8 * at + 16, 16) & 0xFC00) == 0xDC00) {
                                    cp = 0x10000 + ((cp & 0x3FF) << 10) + (take(8 * at + 16, 16) & 0x3FF);
                                    at += 2;
                                }
                                appendUtf8(text, cp);
                            }
                    }
                    if ((int) text.size() > length()) text.resize((size_t) length());          // bytes, not characters (talkeralias.cpp:104-106)
                    return text;
                }
            private:
                int format() const { return bytes[0] >> 6; }
                int length() const { return (bytes[0] >> 1) & 0x1F; }
                int usable() const {                           // bytes of the blocks present without a gap, header first
                    int blocks = 0;
                    while (blocks < 4 && (have >> blocks & 1u)) blocks++;
                    return 7 * blocks;
                }
                uint32_t take(int bit, int count) const {      // `count` bits of the 224-bit string from position `bit`, MSB first
                    uint32_t v = 0;
                    for (int i = bit; i < bit + count; i++) v = v << 1 | (bytes[i >> 3] >> (7 - (i & 7)) & 1u);
                    return v;
                }
                unsigned char bytes[28] = { 0 };
                unsigned have = 0;
        };

        class MetaCollector: public Digiham::MetaCollector {
            public:
                void consume(const dh_event& ev) override {
                    const int s = ev.a & 1;
                    Call& call = calls[s];
                    switch (ev.type) {
                        case DH_EV_DMR_SYNC:                   // a burst with a sync pattern (dmr_phase.cpp:109-114)
                            settleAlias(s, ev.sym_index, ev.b == SYNCTYPE_VOICE);
                            call.fields.put(SYNC, ev.b == SYNCTYPE_DATA ? "data" : ev.b == SYNCTYPE_VOICE ? "voice" : "unknown");
                            if (ev.len > 0 && ev.payload[0]) call.endCall();       // voice -> data on this slot
                            if (ev.b != SYNCTYPE_VOICE) call.alias.reset();        // :233: every non-voice burst
                            publish(s);
                            break;
                        case DH_EV_DMR_SLOTTYPE:               // a data burst (:233)
                            settleAlias(s, ev.sym_index, false);
                            call.alias.reset();
                            break;
                        case DH_EV_DMR_SLOT_RESET:             // the slot lost its sync (:80, :178-180, :196-198, :295)
                            if (ev.b == 1) {
                                // :80, the OTHER slot after a TACT slot switch: Slot::reset() leaves the talker alias collector alone; the
                                // collector is cleared by that slot's next burst (:233) -- unless the burst brings a voice sync
                                call.loseKeepingAlias();
                                aliasDeadline[s] = ev.sym_index + 144u; aliasPending[s] = true;
                            } else {
                                settleAlias(s, ev.sym_index, false);
                                call.lose();
                            }
                            publish(s);
                            break;
                        case DH_EV_DMR_META_RESET:             // the decoder fell back to its sync search (:184, :202)
                            for (Call& c : calls) { c.fields.wipe(); c.alias.reset(); }     // a new FramePhase starts with empty alias collectors
                            aliasPending[0] = aliasPending[1] = false;
                            publish(0); publish(1);
                            break;
                        case DH_EV_DMR_SOFT_RESET:             // terminator LC / idle burst (:279-282)
                            call.endCall();
                            publish(s);
                            break;
                        case DH_EV_DMR_LC:                     // a link control word, voice header or embedded (:304-339)
                            settleAlias(s, ev.sym_index, false);
                            if (ev.len >= 9) { linkControl(call, Lc(ev.payload)); publish(s); }
                            break;
                        default:
                            break;
                    }
                }
            protected:
                std::string getProtocol() override { return "DMR"; }
            private:
                enum Field { LAT, LON, SOURCE, SYNC, ALIAS, TARGET, TYPE, N_FIELDS };          // wire (= key) order
                struct Call {
                    FieldRecord<N_FIELDS> fields { { { "lat", "lon", "source", "sync", "talkeralias", "target", "type" } } };
                    TalkerAliasCollector alias;
                    void endCall() { for (Field f : { TYPE, SOURCE, TARGET, ALIAS, LAT, LON }) fields.put(f, std::string()); }   // Slot::softReset
                    void loseKeepingAlias() { endCall(); fields.put(SYNC, std::string()); }                                  // Slot::reset
                    void lose() { loseKeepingAlias(); alias.reset(); }                                                       // ... in a burst that then runs :233
                };
                static std::string number(uint32_t v) { return v ? std::to_string(v) : std::string(); }      // 0 = not known
                void linkControl(Call& call, Lc lc) {
                    const unsigned op = lc.getOpCode();
                    if (op == LC_OPCODE_GROUP || op == LC_OPCODE_UNIT_TO_UNIT) {
                        call.fields.put(TYPE, op == LC_OPCODE_GROUP ? "group" : "direct");
                        call.fields.put(TARGET, number(lc.getTarget()));
                        call.fields.put(SOURCE, number(lc.getSource()));
                    } else if (op >= LC_TALKER_ALIAS_HDR && op <= LC_TALKER_ALIAS_BLK3) {
                        call.alias.setBlock((int) (op - LC_TALKER_ALIAS_HDR), lc.getData());
                        if (call.alias.isComplete()) {
                            std::string text = call.alias.getContents();
                            text.erase(text.find_last_not_of('\0') + 1);           // padding NULs (dmr_phase.cpp:325-327)
                            call.fields.put(ALIAS, text);
                        }
                    } else if (op == LC_GPS_INFO) {
                        std::unique_ptr<Coordinate> c(Gps::parse(lc.getData()));
                        call.fields.put(LAT, std::to_string(c->lat));
                        call.fields.put(LON, std::to_string(c->lon));
                    }
                }
                void publish(int s) {                          // one line for this slot if anything changed (dmr_meta.cpp:157-170)
                    if (!calls[s].fields.takeChanged()) return;
                    MetaMap m = collect();
                    m["slot"] = std::to_string(s);
                    calls[s].fields.addTo(m);
                    sendMetaData(std::move(m));
                }
                // a slot reset by the TACT switch keeps its collected alias blocks until its next burst (144 symbols on) has been
                // seen not to carry a voice sync
                void settleAlias(int s, uint32_t sym_index, bool voice_sync) {
                    if (!aliasPending[s]) return;
                    aliasPending[s] = false;
                    if (!(voice_sync && sym_index == aliasDeadline[s])) calls[s].alias.reset();
                }
                Call calls[2];
                bool aliasPending[2] = { false, false };
                uint32_t aliasDeadline[2] = { 0, 0 };
        };

    }
}
