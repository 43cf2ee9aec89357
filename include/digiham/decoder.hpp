// decoder.hpp -- Digiham::Decoder, Digiham::Dmr::Decoder, Digiham::Ysf::Decoder on the MI355X engine
// (reference: include/decoder.hpp:17-30, include/dmr_decoder.hpp:9-17, include/ysf_decoder.hpp:9-12).
//
// Output bytes are the reference's.  Every call the reference's frame parsers make into their MetaCollector
// arrives from the GPU as a dh_event (FEC-corrected LC / FICH / DCH bytes); with a MetaWriter set, the host-side
// collectors of dmr_meta.hpp / ysf_meta.hpp replay them into the reference's `k:v;k:v\n` lines.  The raw events
// are also available through the event callback.
#pragma once

#include <functional>
#include <memory>
#include <vector>

#include "csdr_compat.hpp"
#include "engine_handle.hpp"
#include "shared_engine.hpp"
#include "dmr_meta.hpp"
#include "ysf_meta.hpp"
#include "nxdn_meta.hpp"
#include "dstar_meta.hpp"

#define BUF_SIZE 128
#define RINGBUFFER_SIZE 1024

namespace Digiham {

    class Decoder: public Csdr::Module<unsigned char, unsigned char> {
        public:
            ~Decoder() override { if (bank) bank->detach(slot); dh_device_free(dSyms); dh_device_free(dCount); delete metaCollector; }
            bool canProcess() override {
                std::lock_guard<std::mutex> lock(processMutex);
                const bool fresh = reader->available() > 0 && writer->writeable() >= maxOutputPerCall;
                if (!shared()) return fresh;
                const bool queued = bank->hasOutput(slot);
                return bank->hasPending(slot) || (queued && writer->writeable() > 0) || (!queued && fresh);
            }
            void process() override {
                std::lock_guard<std::mutex> lock(processMutex);
                if (shared()) {                        // one engine for every decoder instance of the process (shared_engine.hpp)
                    const bool want = (bool) onEvent || metaCollector != nullptr;
                    if (want != wantsEvents) { bank->setWantEvents(slot, want); wantsEvents = want; }
                    deliver();
                    if (bank->hasPending(slot)) { bank->settle(slot); deliver(); return; }
                    if (bank->hasOutput(slot)) return;
                    if (!(reader->available() > 0 && writer->writeable() >= maxOutputPerCall)) return;
                    const size_t n = std::min(reader->available(), Amd::SharedEngine::chunk);
                    bank->deposit(slot, reader->getReadPointer(), n);
                    reader->advance(n);
                    deliver();
                    return;
                }
                ensure();
                size_t n = reader->available();
                if (n > chunk) n = chunk;
                Amd::check(dh_copy_to_device(dSyms, reader->getReadPointer(), n), "dh_copy_to_device");
                uint32_t cnt = (uint32_t) n;
                Amd::check(dh_copy_to_device(dCount, &cnt, sizeof(cnt)), "dh_copy_to_device");
                Amd::check(dh_engine_push_symbols(engine->get(), (const uint8_t*) dSyms, chunk, (const uint32_t*) dCount), "dh_engine_push_symbols");
                reader->advance(n);
                size_t got = writer->writeable();
                Amd::check(dh_engine_read_frames(engine->get(), 0, writer->getWritePointer(), &got), "dh_engine_read_frames");
                writer->advance(got);
                if (onEvent || metaCollector) {
                    events.resize(chunk / 20 + 64);
                    size_t ne = events.size();
                    Amd::check(dh_engine_read_events(engine->get(), 0, events.data(), &ne), "dh_engine_read_events");
                    for (size_t i = 0; i < ne; i++) {
                        if (metaCollector) metaCollector->consume(events[i]);
                        if (onEvent) onEvent(events[i]);
                    }
                    if (metaCollector) metaCollector->flush();
                }
            }
            // takes ownership of the writer, as the reference does (src/lib/decoder.cpp:34-40)
            void setMetaWriter(MetaWriter* meta) {
                std::lock_guard<std::mutex> lock(processMutex);
                if (proto == DH_PROTO_POCSAG) { delete meta; return; }        // no collector: the writer is released (decoder.cpp:34-38)
                if (!metaCollector) metaCollector = proto == DH_PROTO_DMR ? (MetaCollector*) new Dmr::MetaCollector()
                                                  : proto == DH_PROTO_YSF ? (MetaCollector*) new Ysf::MetaCollector()
                                                  : proto == DH_PROTO_DSTAR ? (MetaCollector*) new DStar::MetaCollector()
                                                  : (MetaCollector*) new Nxdn::MetaCollector();
                metaCollector->setWriter(meta);
            }
            void setEventCallback(std::function<void(const dh_event&)> cb) { onEvent = std::move(cb); }
        protected:
            explicit Decoder(int proto): proto(proto) {}
            void ensure() {
                if (engine) return;
                engine.reset(new Amd::Engine(DH_RRC_NONE, DH_DEMOD_NONE, 0, proto, 0, chunk, slotFilter));
                Amd::check(dh_device_alloc(0, chunk, &dSyms), "dh_device_alloc");
                Amd::check(dh_device_alloc(0, sizeof(uint32_t), &dCount), "dh_device_alloc");
            }
            std::unique_ptr<Amd::Engine> engine;
            unsigned char slotFilter = 3;
            bool shared() {
                if (!decided) {
                    decided = true;
                    if (Amd::SharedEngine::enabled()) {
                        bank = Amd::SharedEngine::join(Amd::SharedEngine::DECODER, DH_RRC_NONE, DH_DEMOD_NONE, 0, proto, 0, slot);
                        if (bank && proto == DH_PROTO_DMR && slotFilter != 3) bank->setSlotFilter(slot, slotFilter);
                    }
                }
                return (bool) bank;
            }
            std::shared_ptr<Amd::SharedEngine> bank;
            int slot = -1;
        private:
            void deliver() {
                const size_t n = bank->take(slot, writer->getWritePointer(), writer->writeable());
                if (n) writer->advance(n);
                // (always taken, consumer or not: events left in the slot would count as undelivered output for ever)
                const std::vector<dh_event> evs = bank->takeEvents(slot);
                if (onEvent || metaCollector) {
                    for (const dh_event& e : evs) {
                        if (metaCollector) metaCollector->consume(e);
                        if (onEvent) onEvent(e);
                    }
                    if (metaCollector) metaCollector->flush();
                }
            }
            bool decided = false, wantsEvents = false;
            static constexpr size_t chunk = 16384;
            // a call may emit one voice payload per 144-symbol burst (DMR, 27 bytes), 95 bytes per 480-symbol frame (YSF)
            // or 36 bytes per 192-symbol frame (NXDN); a POCSAG page line can take up to about half a byte per input bit
            static constexpr size_t maxOutputPerCall = chunk / 2 + 512;      // POCSAG lines are the densest output
            int proto;
            void* dSyms = nullptr;
            void* dCount = nullptr;
            std::vector<dh_event> events;
            std::function<void(const dh_event&)> onEvent;
            MetaCollector* metaCollector = nullptr;
    };

    namespace Dmr {

        class Decoder: public Digiham::Decoder {
            public:
                Decoder(): Digiham::Decoder(DH_PROTO_DMR) {}
                void setSlotFilter(unsigned char filter) {
                    std::lock_guard<std::mutex> lock(processMutex);   // the reference races here (dmr_cli.cpp:57-69)
                    slotFilter = filter;
                    if (bank) bank->setSlotFilter(slot, filter);
                    if (engine) Amd::check(dh_engine_set_slot_filter(engine->get(), filter), "dh_engine_set_slot_filter");
                }
        };

    }

    namespace Ysf {

        class Decoder: public Digiham::Decoder {
            public:
                Decoder(): Digiham::Decoder(DH_PROTO_YSF) {}
        };

    }

    namespace Pocsag {

        // include/pocsag_decoder.hpp: the decoded pages are the OUTPUT of this module (`address:<n>;message:<text>\n` lines
        // from its serializer on the writer), there is no metadata side channel
        class Decoder: public Digiham::Decoder {
            public:
                Decoder(): Digiham::Decoder(DH_PROTO_POCSAG) {}
                explicit Decoder(Serializer* serializer): Decoder() {
                    // the engine writes the reference's StringSerializer format; another serializer cannot be honoured
                    StringSerializer* s = dynamic_cast<StringSerializer*>(serializer);
                    if (s == nullptr) { delete serializer; throw std::invalid_argument("Digiham::Pocsag::Decoder: only the StringSerializer format is available"); }
                    delete s;
                }
        };

    }

    namespace DStar {

        class Decoder: public Digiham::Decoder {          // include/dstar_decoder.hpp, dstar_decoder.cpp:7-9
            public:
                Decoder(): Digiham::Decoder(DH_PROTO_DSTAR) {}
        };

    }

    namespace Nxdn {

        class Decoder: public Digiham::Decoder {          // include/nxdn_decoder.hpp:9-14
            public:
                Decoder(): Digiham::Decoder(DH_PROTO_NXDN) {}
        };

    }

}
