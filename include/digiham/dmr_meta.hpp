// dmr_meta.hpp -- Digiham::Dmr::{Slot, MetaCollector, TalkerAliasCollector, Gps} on decoder events.
// Reference: include/dmr_meta.hpp, src/dmr_decoder/dmr_meta.cpp:11-179 (slot state + `protocol:DMR;slot:N;...`
// lines), talkeralias.cpp:20-144, gps.cpp:7-17, and the call sites in dmr_phase.cpp (:80, :111-114, :178-184,
// :196-202, :233, :280, :295, :304-339).  PARITY UNPINNED (see meta.hpp).
#pragma once

#include <cstring>
#include <string>

#include "lc.hpp"
#include "meta.hpp"

#define META_TYPE_DIRECT 1
#define META_TYPE_GROUP 2
#define SYNCTYPE_DATA 1
#define SYNCTYPE_VOICE 2

#define TALKER_ALIAS_FORMAT_7BIT 0
#define TALKER_ALIAS_FORMAT_8BIT 1
#define TALKER_ALIAS_FORMAT_UTF8 2
#define TALKER_ALIAS_FORMAT_UTF16 3

namespace Digiham {
    namespace Dmr {

        struct Gps {                                            // gps.cpp:7-17
            static Coordinate* parse(const unsigned char* data) {
                int32_t latitudeBits = ((data[4] & 0x7F) << 16) | (data[5] << 8) | data[6];
                if (data[4] & 0x80) latitudeBits *= -1;
                int32_t longitudeBits = (data[1] << 16) | (data[2] << 8) | data[3];
                if (data[0] & 0x01) longitudeBits *= -1;
                return new Coordinate(180.0f / (float) (1 << 24) * (float) latitudeBits,
                                      360.0f / (float) (1 << 25) * (float) longitudeBits);
            }
        };

        class TalkerAliasCollector {                            // talkeralias.cpp:20-144
            public:
                void reset() { blocks = 0; }
                void setBlock(int block, const unsigned char* d) {
                    std::memcpy(data + (size_t) block * 7, d, 7);
                    blocks |= 1 << block;
                }
                bool isComplete() {
                    if (!hasHeader()) return false;
                    const unsigned char bytes = collectedBytes();
                    switch (getDataFormat()) {
                        case TALKER_ALIAS_FORMAT_7BIT: return ((bytes * 7) / 8) - 1 >= getLength();
                        case TALKER_ALIAS_FORMAT_8BIT: return bytes - 1 >= getLength();
                        case TALKER_ALIAS_FORMAT_UTF8: return getContents().length() >= getLength();
                        case TALKER_ALIAS_FORMAT_UTF16: return (bytes - 1) / 2 >= getLength();
                    }
                    return false;
                }
                std::string getContents() {
                    if (!hasHeader()) return "";
                    const unsigned char bytes = collectedBytes();
                    std::string result;
                    switch (getDataFormat()) {
                        case TALKER_ALIAS_FORMAT_7BIT: {
                            std::string all;
                            for (size_t i = 0; i < bytes; i += 7) all += convert7BitData(data + i);
                            result = all.substr(1);     // first character is built from the header bits
                            break;
                        }
                        case TALKER_ALIAS_FORMAT_8BIT:
                            result = Converter::convertToUtf8((const char*) data + 1, bytes - 1);
                            break;
                        case TALKER_ALIAS_FORMAT_UTF8:
                            result = std::string((const char*) data + 1, bytes - 1);
                            break;
                        case TALKER_ALIAS_FORMAT_UTF16: {
                            // big-endian UTF-16 code units to UTF-8 (std::codecvt_utf8_utf16 in the reference, :96-108)
                            const unsigned int chars = (bytes - 1) / 2;
                            const unsigned char* src = data + 1;
                            for (unsigned int k = 0; k < chars; k++) {
                                uint32_t cp = (uint32_t) (src[k * 2] << 8) | src[k * 2 + 1];
                                if (cp >= 0xD800 && cp < 0xDC00 && k + 1 < chars) {
                                    const uint32_t lo = (uint32_t) (src[k * 2 + 2] << 8) | src[k * 2 + 3];
                                    if (lo >= 0xDC00 && lo < 0xE000) { cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); k++; }
                                }
                                appendUtf8(result, cp);
                            }
                            break;
                        }
                    }
                    if (result.length() > getLength()) result = result.substr(0, getLength());
                    return result;
                }
            private:
                bool hasHeader() const { return blocks & 1; }
                unsigned char getDataFormat() const { return data[0] >> 6; }
                unsigned char getLength() const { return (data[0] & 0x3E) >> 1; }
                unsigned char collectedBytes() const {
                    int i;
                    for (i = 0; i < 4; i++) {
                        const unsigned char mask = (unsigned char) ((1 << (i + 1)) - 1);
                        if ((blocks & mask) != mask) break;
                    }
                    return (unsigned char) (i * 7);
                }
                static std::string convert7BitData(const unsigned char* s) {   // :131-142
                    unsigned char r[8];
                    r[0] = (s[0] & 0xFE) >> 1;
                    r[1] = (unsigned char) ((s[0] & 0x01) << 6 | (s[1] & 0xFC) >> 2);
                    r[2] = (unsigned char) ((s[1] & 0x03) << 5 | (s[2] & 0xF8) >> 3);
                    r[3] = (unsigned char) ((s[2] & 0x07) << 4 | (s[3] & 0xF0) >> 4);
                    r[4] = (unsigned char) ((s[3] & 0x0F) << 3 | (s[4] & 0xE0) >> 5);
                    r[5] = (unsigned char) ((s[4] & 0x1F) << 2 | (s[5] & 0xC0) >> 6);
                    r[6] = (unsigned char) ((s[5] & 0x3F) << 1 | (s[6] & 0x80) >> 7);
                    r[7] = s[6] & 0x7F;
                    return std::string((const char*) r, 8);
                }
                static void appendUtf8(std::string& out, uint32_t cp) {
                    if (cp < 0x80) out.push_back((char) cp);
                    else if (cp < 0x800) { out.push_back((char) (0xC0 | (cp >> 6))); out.push_back((char) (0x80 | (cp & 0x3F))); }
                    else if (cp < 0x10000) {
                        out.push_back((char) (0xE0 | (cp >> 12))); out.push_back((char) (0x80 | ((cp >> 6) & 0x3F)));
                        out.push_back((char) (0x80 | (cp & 0x3F)));
                    } else {
                        out.push_back((char) (0xF0 | (cp >> 18))); out.push_back((char) (0x80 | ((cp >> 12) & 0x3F)));
                        out.push_back((char) (0x80 | ((cp >> 6) & 0x3F))); out.push_back((char) (0x80 | (cp & 0x3F)));
                    }
                }
                unsigned char data[28] = { 0 };
                unsigned char blocks = 0;
        };

        class Slot {                                            // dmr_meta.cpp:7-131
            public:
                ~Slot() { delete coordinate; }
                void setSync(int v) { if (sync == v) return; sync = v; dirty = true; }
                void setType(int v) { if (type == v) return; type = v; dirty = true; }
                void setSource(uint32_t v) { if (source == v) return; source = v; dirty = true; }
                void setTarget(uint32_t v) { if (target == v) return; target = v; dirty = true; }
                void setFromLc(Lc* lc) {
                    switch (lc->getOpCode()) {
                        case LC_OPCODE_GROUP: setType(META_TYPE_GROUP); break;
                        case LC_OPCODE_UNIT_TO_UNIT: setType(META_TYPE_DIRECT); break;
                        default: break;
                    }
                    setTarget(lc->getTarget());
                    setSource(lc->getSource());
                }
                void setTalkerAlias(const std::string& alias) { if (talkerAlias == alias) return; talkerAlias = alias; dirty = true; }
                void setCoordinate(Coordinate* coord) {
                    if (coordinate == coord || (coordinate != nullptr && coord != nullptr && *coordinate == *coord)) { delete coord; return; }
                    auto old = coordinate; coordinate = coord; delete old;
                    dirty = true;
                }
                bool isDirty() const { return dirty; }
                void setClean() { dirty = false; }
                void softReset() { setType(-1); setSource(0); setTarget(0); setTalkerAlias(""); setCoordinate(nullptr); }
                void reset() { softReset(); setSync(-1); }
                std::map<std::string, std::string> collect() {
                    std::map<std::string, std::string> result;
                    if (sync > 0) result["sync"] = sync == SYNCTYPE_DATA ? "data" : sync == SYNCTYPE_VOICE ? "voice" : "unknown";
                    if (type > 0) result["type"] = type == META_TYPE_DIRECT ? "direct" : type == META_TYPE_GROUP ? "group" : "unknown";
                    if (source > 0) result["source"] = std::to_string(source);
                    if (target > 0) result["target"] = std::to_string(target);
                    if (!talkerAlias.empty()) result["talkeralias"] = talkerAlias;
                    if (coordinate != nullptr) {
                        result["lat"] = std::to_string(coordinate->lat);
                        result["lon"] = std::to_string(coordinate->lon);
                    }
                    return result;
                }
            private:
                bool dirty = false;
                int sync = -1, type = -1;
                uint32_t source = 0, target = 0;
                std::string talkerAlias;
                Coordinate* coordinate = nullptr;
        };

        class MetaCollector: public Digiham::MetaCollector {    // dmr_meta.cpp:133-179 + the call sites in dmr_phase.cpp
            public:
                void consume(const dh_event& ev) override {
                    const int slot = ev.a & 1;
                    switch (ev.type) {
                        case DH_EV_DMR_SYNC:                    // dmr_phase.cpp:109-114
                            slots[slot].setSync(ev.b);
                            if (ev.len > 0 && ev.payload[0]) slots[slot].softReset();
                            if (ev.b != SYNCTYPE_VOICE) aliases[slot].reset();     // :233 (every non-voice burst of the slot)
                            sendMetaDataForSlot(slot);
                            break;
                        case DH_EV_DMR_SLOT_RESET:              // :80, :178-180, :196-198, :295
                            slots[slot].reset();
                            aliases[slot].reset();
                            sendMetaDataForSlot(slot);
                            break;
                        case DH_EV_DMR_SLOTTYPE:                // a data burst: :233
                            aliases[slot].reset();
                            break;
                        case DH_EV_DMR_META_RESET:              // :184, :202
                            for (int i = 0; i < 2; i++) slots[i].reset();
                            for (int i = 0; i < 2; i++) sendMetaDataForSlot(i);
                            break;
                        case DH_EV_DMR_SOFT_RESET:              // :279-282 (terminator LC / idle)
                            slots[slot].softReset();
                            sendMetaDataForSlot(slot);
                            break;
                        case DH_EV_DMR_LC:                      // handleLc, :304-339
                            if (ev.len >= 9) handleLc(slot, ev.payload);
                            break;
                        default:
                            break;
                    }
                }
            protected:
                std::string getProtocol() override { return "DMR"; }
            private:
                void handleLc(int slot, const unsigned char* bytes) {
                    Lc lc(bytes);
                    const unsigned char opcode = lc.getOpCode();
                    switch (opcode) {
                        case LC_OPCODE_GROUP:
                        case LC_OPCODE_UNIT_TO_UNIT:
                            slots[slot].setFromLc(&lc);
                            sendMetaDataForSlot(slot);
                            break;
                        case LC_TALKER_ALIAS_HDR: case LC_TALKER_ALIAS_BLK1: case LC_TALKER_ALIAS_BLK2: case LC_TALKER_ALIAS_BLK3:
                            aliases[slot].setBlock(opcode - LC_TALKER_ALIAS_HDR, lc.getData());
                            if (aliases[slot].isComplete()) {
                                std::string alias = aliases[slot].getContents();
                                const auto end = alias.find_last_not_of('\0');
                                alias = end == std::string::npos ? "" : alias.substr(0, end + 1);
                                slots[slot].setTalkerAlias(alias);
                                sendMetaDataForSlot(slot);
                            }
                            break;
                        case LC_GPS_INFO:
                            slots[slot].setCoordinate(Gps::parse(lc.getData()));
                            sendMetaDataForSlot(slot);
                            break;
                        default:
                            break;
                    }
                }
                void sendMetaDataForSlot(int i) {               // dmr_meta.cpp:157-170
                    if (!slots[i].isDirty()) return;
                    auto metadata = Digiham::MetaCollector::collect();
                    metadata["slot"] = std::to_string(i);
                    auto slotMetadata = slots[i].collect();
                    metadata.insert(slotMetadata.begin(), slotMetadata.end());
                    Digiham::MetaCollector::sendMetaData(metadata);
                    slots[i].setClean();
                }
                Slot slots[2];
                TalkerAliasCollector aliases[2];
        };

    }
}
