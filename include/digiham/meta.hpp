// meta.hpp -- metadata plumbing of the decoders, host side: Serializer / MetaWriter / MetaCollector with the
// reference's names and wire format (include/meta.hpp:11-78, src/lib/meta.cpp:8-110): one line per change,
// `key:value;key:value\n`, keys in std::map order.
//
// In the reference the protocol state machines call into a MetaCollector while they parse frames.  Here the
// frames are parsed on the GPU and every such call arrives as a dh_event (include/digiham_amd.h); the
// collectors below replay them in order (`consume`).  The text lines have no golden vectors in the reference
// (it has no tests) and the reference cannot be built here (csdr is absent), so these lines are PARITY UNPINNED:
// they are checked against the formats and state rules cited at each method.
#pragma once

#include <cstdio>
#include <map>
#include <sstream>
#include <string>

#include "../digiham_amd.h"

namespace Digiham {

    // Converter::convertToUtf8 (src/lib/charset.cpp:10-27) for its only charset in use, ISO-8859-1 (charset.hpp:9):
    // ICU maps byte b to U+00b; the result is cut at the first NUL because the reference builds a std::string
    // from a C string (charset.cpp:23).
    struct Converter {
        static std::string convertToUtf8(const char* input, size_t length) {
            std::string out;
            for (size_t i = 0; i < length; i++) {
                const unsigned char b = (unsigned char) input[i];
                if (b == 0) break;
                if (b < 0x80) out.push_back((char) b);
                else { out.push_back((char) (0xC0 | (b >> 6))); out.push_back((char) (0x80 | (b & 0x3F))); }
            }
            return out;
        }
    };

    class Coordinate {                      // src/lib/coordinate.cpp:5-9
        public:
            Coordinate(float lat, float lon): lat(lat), lon(lon) {}
            bool operator==(const Coordinate& other) const { return other.lat == lat && other.lon == lon; }
            float lat, lon;
    };

    class Serializer {
        public:
            virtual ~Serializer() = default;
            virtual std::string serializeMetaData(std::map<std::string, std::string> metadata) = 0;
    };

    class StringSerializer: public Serializer {            // src/lib/meta.cpp:8-17
        public:
            std::string serializeMetaData(std::map<std::string, std::string> metadata) override {
                std::stringstream ss;
                for (auto it = metadata.begin(); it != metadata.end(); it++) {
                    if (it != metadata.begin()) ss << ";";
                    ss << it->first << ":" << it->second;
                }
                ss << "\n";
                return ss.str();
            }
    };

    class MetaWriter {                                      // src/lib/meta.cpp:19-33
        public:
            explicit MetaWriter(Serializer* serializer): serializer(serializer) {}
            MetaWriter(): MetaWriter(new StringSerializer()) {}
            virtual ~MetaWriter() { delete serializer; }
            virtual void sendMetaData(std::map<std::string, std::string> metadata) = 0;
            void setSerializer(Serializer* s) {
                if (s == serializer) return;
                auto old = serializer; serializer = s; delete old;
            }
        protected:
            Serializer* serializer;
    };

    class FileMetaWriter: public MetaWriter {               // src/lib/meta.cpp:35-47
        public:
            explicit FileMetaWriter(FILE* out): MetaWriter(), file(out) {}
            FileMetaWriter(FILE* out, Serializer* serializer): MetaWriter(serializer), file(out) {}
            ~FileMetaWriter() override { if (file) fclose(file); }
            void sendMetaData(std::map<std::string, std::string> metadata) override {
                if (!file) return;
                const std::string s = serializer->serializeMetaData(std::move(metadata));
                fwrite(s.c_str(), 1, s.length(), file);
                fflush(file);
            }
        private:
            FILE* file;
    };

    class MetaCollector {                                   // src/lib/meta.cpp:58-110
        public:
            MetaCollector() = default;
            virtual ~MetaCollector() { delete writer; }
            void setWriter(MetaWriter* w) { delete writer; writer = w; }
            void hold() { held++; }
            void release() {
                held--;
                if (held == 0) { if (dirty) sendMetaData(); dirty = false; }
            }
            // one decoder event = one call the reference's frame parser makes into its collector
            virtual void consume(const dh_event& ev) = 0;
            // end of a batch of events (a decoder call): frames never straddle batches, so a header group still held
            // open is complete
            virtual void flush() {}
        protected:
            virtual std::string getProtocol() = 0;
            virtual std::map<std::string, std::string> collect() { return std::map<std::string, std::string> { {"protocol", getProtocol()} }; }
            void sendMetaData(std::map<std::string, std::string> metadata) { if (writer) writer->sendMetaData(std::move(metadata)); }
            virtual void sendMetaData() {
                if (writer == nullptr) return;
                if (held) { dirty = true; return; }
                sendMetaData(collect());
            }
        private:
            MetaWriter* writer = nullptr;
            int held = 0;
            bool dirty = false;
    };

}
