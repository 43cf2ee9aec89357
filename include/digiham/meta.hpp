// meta.hpp -- where the decoders' metadata goes, host side.
//
// Boundary (what OpenWebRX / pycsdr plug into, reference include/meta.hpp:11-78): a decoder owns a MetaCollector, the
// caller hands it a MetaWriter (FileMetaWriter for the command line tools' --fifo, PipelineMetaWriter as a
// Csdr::Source<unsigned char> inside a csdr pipeline), and every change of the call's state leaves as one text line
//     key:value;key:value\n          keys in std::map order (src/lib/meta.cpp:8-17)
// The class names, the ownership rules (a collector owns its writer, a writer owns its serializer: meta.cpp:23-33,
// :62-69) and the hold() / release() batching contract (meta.cpp:71-81, :97-104) are the reference's; everything
// behind them is built around this engine instead: frames are parsed on the GPU, every call the reference's frame
// parser would make into its collector arrives as a dh_event (include/digiham_amd.h), and the protocol collectors
// (dmr_meta.hpp, ysf_meta.hpp, nxdn_meta.hpp, dstar_meta.hpp) are replay machines over those events that keep their
// call state in a FieldRecord -- a fixed, key-ordered set of text fields with one change flag.
//
// Pinning: Coordinate / the Latin-1 converter / GPS / talker alias / YSF data frames are checked against the
// reference's own classes (tests/golden/elements_ref.npz, produced from oracle/_ref); the line-level state machines
// (src/*/..._meta.cpp + the call sites in *_phase.cpp need csdr) are restated from the cited lines and PARITY UNPINNED.
#pragma once

#include <array>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <utility>

#include "../digiham_amd.h"
#include "csdr_compat.hpp"

namespace Digiham {

    typedef std::map<std::string, std::string> MetaMap;

    // ISO-8859-1 -> UTF-8, the only conversion the reference asks ICU for (Converter::convertToUtf8, charset.hpp:9):
    // byte b is U+00b.  The reference copies ICU's output out of a C string (charset.cpp:22), so a NUL ends the text.
    struct Converter {
        static std::string convertToUtf8(const char* input, size_t length) {
            std::string text;
            text.reserve(length + length / 2);
            for (const char* p = input, *end = input + length; p != end && *p != '\0'; ++p) {
                const unsigned char b = (unsigned char) *p;
                if (b & 0x80) { text += (char) (0xC0 | (b >> 6)); text += (char) (0x80 | (b & 0x3F)); }
                else text += (char) b;
            }
            return text;
        }
    };

    // one Unicode scalar value as UTF-8 (talker alias UTF-16 blocks, dmr_meta.hpp)
    inline void appendUtf8(std::string& text, uint32_t cp) {
        const int extra = cp < 0x80 ? 0 : cp < 0x800 ? 1 : cp < 0x10000 ? 2 : 3;
        static const unsigned char lead[4] = { 0x00, 0xC0, 0xE0, 0xF0 };
        text += (char) (lead[extra] | (cp >> (6 * extra)));
        for (int k = extra - 1; k >= 0; k--) text += (char) (0x80 | ((cp >> (6 * k)) & 0x3F));
    }

    struct Coordinate {                                    // src/lib/coordinate.hpp:5-13
        Coordinate(float lat, float lon): lat(lat), lon(lon) {}
        bool operator==(const Coordinate& o) const { return o.lat == lat && o.lon == lon; }
        float lat, lon;
    };

    // ------------------------------------------------------------------ serializer / writers
    class Serializer {
        public:
            virtual ~Serializer() = default;
            virtual std::string serializeMetaData(MetaMap metadata) = 0;
    };

    class StringSerializer: public Serializer {
        public:
            std::string serializeMetaData(MetaMap metadata) override {
                std::string line;
                const char* sep = "";
                for (const auto& kv : metadata) { ((line += sep) += kv.first) += ':'; line += kv.second; sep = ";"; }
                return line += '\n';
            }
    };

    class MetaWriter {
        public:
            MetaWriter(): MetaWriter(new StringSerializer()) {}
            explicit MetaWriter(Serializer* s): serializer(s) {}
            virtual ~MetaWriter() { delete serializer; }
            virtual void sendMetaData(MetaMap metadata) = 0;
            void setSerializer(Serializer* s) {               // takes ownership; handing back the current one is a no-op
                if (s != serializer) { delete serializer; serializer = s; }
            }
        protected:
            std::string render(MetaMap& metadata) { return serializer->serializeMetaData(std::move(metadata)); }
            Serializer* serializer;
    };

    // lines to a stdio stream the writer owns (the tools' --fifo: src/lib/cli.cpp:127-131); a stream that failed to
    // open swallows the lines instead of crashing the decoder
    class FileMetaWriter: public MetaWriter {
        public:
            explicit FileMetaWriter(FILE* out): out(out) {}
            FileMetaWriter(FILE* out, Serializer* s): MetaWriter(s), out(out) {}
            ~FileMetaWriter() override { if (out != nullptr) fclose(out); }
            void sendMetaData(MetaMap metadata) override {
                if (out == nullptr) return;
                const std::string line = render(metadata);
                fwrite(line.data(), 1, line.size(), out);
                fflush(out);
            }
        private:
            FILE* out;
    };

    // lines into a csdr pipeline: the writer is a Csdr::Source<unsigned char>, whoever owns the pipeline attaches a
    // Csdr::Writer to it.  A line that does not fit the downstream buffer in one piece is dropped whole (the
    // reference's "can't write...", meta.cpp:51-56); so is everything sent before a writer is attached.
    class PipelineMetaWriter: public MetaWriter, public Csdr::Source<unsigned char> {
        public:
            explicit PipelineMetaWriter(Serializer* s): MetaWriter(s) {}
            void sendMetaData(MetaMap metadata) override {
                const std::string line = render(metadata);
                if (writer == nullptr || writer->writeable() < line.size()) return;
                std::memcpy(writer->getWritePointer(), line.data(), line.size());
                writer->advance(line.size());
            }
    };

    // ------------------------------------------------------------------ call state as text fields
    // N named text fields in key order (the order std::map gives them on the wire); an empty value is an absent key.
    // put() reports whether the value changed and raises the record's change flag.
    template <size_t N> class FieldRecord {
        public:
            explicit FieldRecord(const std::array<const char*, N>& keys): keys(keys) {}
            bool put(size_t i, const std::string& value) {
                if (values[i] == value) return false;
                values[i] = value;
                return changed = true;
            }
            void wipe() { for (size_t i = 0; i < N; i++) put(i, std::string()); }
            const std::string& get(size_t i) const { return values[i]; }
            bool takeChanged() { const bool was = changed; changed = false; return was; }
            void addTo(MetaMap& m) const { for (size_t i = 0; i < N; i++) if (!values[i].empty()) m[keys[i]] = values[i]; }
        private:
            std::array<const char*, N> keys;
            std::array<std::string, N> values;
            bool changed = false;
    };

    // ------------------------------------------------------------------ collectors
    class MetaCollector {
        public:
            MetaCollector() = default;
            explicit MetaCollector(MetaWriter* w): writer(w) {}
            virtual ~MetaCollector() = default;
            void setWriter(MetaWriter* w) { writer.reset(w); }
            // hold() ... release(): changes in between leave as ONE line when the outermost release() comes
            void hold() { ++depth; }
            void release() { if (--depth == 0 && std::exchange(pending, false)) sendMetaData(); }
            // one decoder event = one call the reference's frame parser makes into its collector
            virtual void consume(const dh_event& ev) = 0;
            // end of a batch of events (one decoder call): frames never straddle batches, so a group still held open is complete
            virtual void flush() {}
        protected:
            virtual std::string getProtocol() = 0;
            virtual MetaMap collect() { return MetaMap { { "protocol", getProtocol() } }; }
            void sendMetaData(MetaMap metadata) { if (writer) writer->sendMetaData(std::move(metadata)); }
            virtual void sendMetaData() {
                if (!writer) return;
                if (depth > 0) pending = true; else sendMetaData(collect());
            }
        private:
            std::unique_ptr<MetaWriter> writer;
            int depth = 0;
            bool pending = false;
    };

}
