// lc.hpp -- Digiham::Dmr::Lc: field view of a 9-byte DMR full link control word, as delivered in
// DH_EV_DMR_LC events (reference: src/dmr_decoder/lc.hpp:16-31, lc.cpp:8-43; the RS(12,9) parity of a voice
// header is not checked there either, lc.cpp:9).
#pragma once

#include <cstdint>
#include <cstring>

#define LC_OPCODE_GROUP 0
#define LC_OPCODE_UNIT_TO_UNIT 3
#define LC_TALKER_ALIAS_HDR 4
#define LC_TALKER_ALIAS_BLK1 5
#define LC_TALKER_ALIAS_BLK2 6
#define LC_TALKER_ALIAS_BLK3 7
#define LC_GPS_INFO 8

namespace Digiham {
    namespace Dmr {

        class Lc {
            public:
                static Lc* parseFromVoiceHeader(unsigned char* data) { return new Lc(data); }
                explicit Lc(const unsigned char* data) { std::memcpy(this->data, data, 9); }
                unsigned char getOpCode() const { return data[0] & 0x3F; }
                unsigned char getFeatureSetId() const { return data[1]; }
                uint32_t getSource() const { return (uint32_t) data[6] << 16 | (uint32_t) data[7] << 8 | data[8]; }
                uint32_t getTarget() const { return (uint32_t) data[3] << 16 | (uint32_t) data[4] << 8 | data[5]; }
                unsigned char* getData() { return data + 2; }
            private:
                unsigned char data[9];
        };

    }
}
