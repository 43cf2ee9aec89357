// shared_engine.hpp -- one launch for MANY module instances of a process.
//
// The reference is one process per channel and module (src/lib/cli.cpp:19-37); a receiver such as OpenWebRX that hosts N
// channels in one process creates N x (WideRrcFilter, GfskDemodulator, Dmr::Decoder).  With a 1-channel engine behind every
// instance (the default) that is N x 3 launches, copies and synchronisations per round.  After
//     Digiham::Amd::SharedEngine::enable(capacity);
// instances of the same kind (stage + parameters) attach to ONE engine of `capacity` channels instead:
//   * process() of an instance DEPOSITS what its reader holds into its row of a host staging block (the reader advances) and
//     delivers the output of the previous round to its writer;
//   * the deposit that completes the set -- or the first instance that comes round again with its deposit still pending --
//     runs the TICK: one dh_engine_push_host_ragged / dh_engine_push_symbols for every row that holds something, then the
//     outputs of those rows are fetched.
// The byte streams are the reference's (the tests run 64 triples against the oracle); an instance's output merely arrives one
// call later.  canProcess() is also true while an instance has a deposit pending or output undelivered, so that the usual
// `while (canProcess()) process()` drivers drain the pipeline.  All entry points take the bank's mutex: instances may live on
// different threads (csdr's AsyncRunner).
#pragma once

#include <algorithm>
#include <atomic>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <vector>

#include "engine_handle.hpp"

namespace Digiham {
    namespace Amd {

        class SharedEngine {
            public:
                enum Stage { RRC, SLICER, DECODER };
                static constexpr size_t chunk = 16384;           // samples / symbols a slot deposits per tick at most

                static void enable(unsigned int capacity = 64) { std::lock_guard<std::mutex> l(registryMutex()); capacityRef() = capacity; }
                static void disable() { std::lock_guard<std::mutex> l(registryMutex()); capacityRef() = 0; }
                static bool enabled() { std::lock_guard<std::mutex> l(registryMutex()); return capacityRef() > 0; }
                // launches (ticks) of every bank of the process so far: what a test counts
                static unsigned long ticksTotal() { std::lock_guard<std::mutex> l(registryMutex()); unsigned long t = 0; for (auto& kv : registry()) if (auto p = kv.second.lock()) t += p->ticks; return t; }

                // The bank for this kind of module (created on first use, destroyed with its last instance) AND a slot in it, taken
                // under the registry lock: two threads asking at once cannot both be handed the last free slot of a bank, and a
                // disable() in between cannot make this construct an engine of capacity 0 (null bank: the caller runs its own
                // 1-channel engine, as without sharing).
                static std::shared_ptr<SharedEngine> join(Stage stage, int rrc, int demod, unsigned int sps, int proto, unsigned int flags, int& slot) {
                    std::lock_guard<std::mutex> l(registryMutex());
                    slot = -1;
                    if (capacityRef() == 0) return nullptr;
                    const auto key = std::make_tuple((int) stage, rrc, demod, sps, proto, flags);
                    auto& weak = registry()[key];
                    auto bank = weak.lock();
                    if (!bank || bank->full()) {             // (a full bank stays with its instances; newcomers get the next one)
                        bank.reset(new SharedEngine(stage, rrc, demod, sps, proto, flags, capacityRef()));
                        weak = bank;
                    }
                    slot = bank->attach();
                    return bank;
                }

                ~SharedEngine() { if (dSyms) dh_device_free(dSyms); if (dCounts) dh_device_free(dCounts); }

                int attach() {
                    std::lock_guard<std::mutex> l(mutex);
                    for (unsigned int s = 0; s < slots.size(); s++) if (!slots[s].used) {
                        if (slots[s].dirty) check(dh_engine_reset_channel(engine.get(), s), "dh_engine_reset_channel");
                        slots[s] = Slot(); slots[s].used = true; slots[s].dirty = true;
                        attached++;
                        return (int) s;
                    }
                    throw std::runtime_error("Digiham::Amd::SharedEngine: no free slot");
                }
                void detach(int s) { std::lock_guard<std::mutex> l(mutex); if (slots[s].pending) pendingCount--; slots[s].used = false; slots[s].pending = 0; slots[s].out.clear(); slots[s].events.clear(); attached--; }

                bool hasPending(int s) { std::lock_guard<std::mutex> l(mutex); return slots[s].pending > 0; }
                bool hasOutput(int s) { std::lock_guard<std::mutex> l(mutex); return !slots[s].out.empty() || !slots[s].events.empty(); }

                // deposit n elements (floats for RRC / SLICER, symbols for DECODER); the slot must have no deposit pending
                void deposit(int s, const void* data, size_t n) {
                    std::lock_guard<std::mutex> l(mutex);
                    Slot& sl = slots[s];
                    if (n > chunk || sl.pending) throw std::runtime_error("Digiham::Amd::SharedEngine::deposit: more than one chunk, or a deposit still pending");
                    if (n == 0) return;
                    const size_t w = stage == DECODER ? 1 : sizeof(float);
                    std::memcpy(staging.data() + (size_t) s * chunk * w, data, n * w);
                    sl.pending = (uint32_t) n;
                    pendingCount++;
                    if (pendingCount >= attached) tickLocked();
                }
                // run the tick if this slot's deposit is still waiting for it
                void settle(int s) { std::lock_guard<std::mutex> l(mutex); if (slots[s].pending) tickLocked(); }

                // hand over up to `room` output elements (floats for RRC, bytes otherwise); returns how many
                size_t take(int s, void* dst, size_t room) {
                    std::lock_guard<std::mutex> l(mutex);
                    Slot& sl = slots[s];
                    const size_t w = stage == RRC ? sizeof(float) : 1;
                    const size_t have = sl.out.size() / w, n = std::min(have, room);
                    std::memcpy(dst, sl.out.data(), n * w);
                    sl.out.erase(sl.out.begin(), sl.out.begin() + (long) (n * w));
                    return n;
                }
                std::vector<dh_event> takeEvents(int s) { std::lock_guard<std::mutex> l(mutex); std::vector<dh_event> e; e.swap(slots[s].events); return e; }
                size_t outputSize(int s) { std::lock_guard<std::mutex> l(mutex); return slots[s].out.size() / (stage == RRC ? sizeof(float) : 1); }

                void setSlotFilter(int s, unsigned int filter) {
                    std::lock_guard<std::mutex> l(mutex);
                    check(dh_engine_set_slot_filter_channel(engine.get(), (uint32_t) s, filter), "dh_engine_set_slot_filter_channel");
                }
                // DECODER banks: fetch this slot's event row as well (an instance with a meta writer or an event callback).  Per slot:
                // events fetched for an instance that never takes them would make hasOutput() true for ever.
                void setWantEvents(int s, bool want) { std::lock_guard<std::mutex> l(mutex); slots[s].wantEvents = want; if (!want) slots[s].events.clear(); }

            private:
                struct Slot { bool used = false, dirty = false, wantEvents = false; uint32_t pending = 0; std::vector<unsigned char> out; std::vector<dh_event> events; };

                SharedEngine(Stage stage, int rrc, int demod, unsigned int sps, int proto, unsigned int flags, unsigned int capacity):
                    stage(stage),
                    engine(rrc, demod, sps, proto, flags, (unsigned int) chunk, 3, capacity),
                    slots(capacity), staging((size_t) capacity * chunk * (stage == DECODER ? 1 : sizeof(float))), counts(capacity) {
                    if (stage == DECODER) {
                        check(dh_device_alloc(0, (size_t) capacity * chunk, &dSyms), "dh_device_alloc");
                        check(dh_device_alloc(0, sizeof(uint32_t) * capacity, &dCounts), "dh_device_alloc");
                    }
                }
                bool full() { std::lock_guard<std::mutex> l(mutex); return attached >= slots.size(); }

                void tickLocked() {
                    const unsigned int B = (unsigned int) slots.size();
                    uint32_t most = 0;
                    for (unsigned int s = 0; s < B; s++) { counts[s] = slots[s].pending; most = std::max(most, counts[s]); }
                    if (most == 0) return;
                    if (stage == DECODER) {
                        check(dh_copy_to_device(dSyms, staging.data(), staging.size()), "dh_copy_to_device");
                        check(dh_copy_to_device(dCounts, counts.data(), sizeof(uint32_t) * B), "dh_copy_to_device");
                        check(dh_engine_push_symbols(engine.get(), (const uint8_t*) dSyms, chunk, (const uint32_t*) dCounts), "dh_engine_push_symbols");
                    } else {
                        check(dh_engine_push_host_ragged(engine.get(), reinterpret_cast<const float*>(staging.data()), chunk, counts.data(), most), "dh_engine_push_host_ragged");
                    }
                    ticks++;
                    // The deposits are in the engine now, whatever happens below: a read that throws (DH_ECAPACITY, DH_EDEVICE) must
                    // not leave rows marked pending, or the next tick would push the same staging rows a second time.
                    for (unsigned int s = 0; s < B; s++) slots[s].pending = 0;
                    pendingCount = 0;
                    // Every slot that pushed is read, whatever another slot's read returns: the next tick overwrites the engine's buffers, so a
                    // slot skipped here would have a hole in its byte stream.  The first failure is reported once all of them are through.
                    int firstRc = DH_OK; const char* firstWhat = "";
                    for (unsigned int s = 0; s < B; s++) {
                        Slot& sl = slots[s];
                        const uint32_t pushed = counts[s];
                        if (!pushed) continue;
                        const size_t w = stage == RRC ? sizeof(float) : 1;
                        const size_t old = sl.out.size();
                        size_t got = stage == RRC ? pushed : stage == SLICER ? pushed / 2 + 8 : chunk / 2 + 512;
                        sl.out.resize(old + got * w);
                        const int rc = stage == RRC ? dh_engine_read_filtered(engine.get(), s, reinterpret_cast<float*>(sl.out.data() + old), &got)
                                     : stage == SLICER ? dh_engine_read_symbols(engine.get(), s, sl.out.data() + old, &got)
                                     : dh_engine_read_frames(engine.get(), s, sl.out.data() + old, &got);
                        sl.out.resize(old + (rc == DH_OK ? got : 0) * w);
                        if (rc != DH_OK && firstRc == DH_OK) { firstRc = rc; firstWhat = stage == RRC ? "dh_engine_read_filtered" : stage == SLICER ? "dh_engine_read_symbols" : "dh_engine_read_frames"; }
                        if (stage == DECODER && sl.wantEvents) {
                            const size_t eold = sl.events.size();
                            size_t ne = chunk / 20 + 64;
                            sl.events.resize(eold + ne);
                            const int rce = dh_engine_read_events(engine.get(), s, sl.events.data() + eold, &ne);
                            sl.events.resize(eold + (rce == DH_OK ? ne : 0));
                            if (rce != DH_OK && firstRc == DH_OK) { firstRc = rce; firstWhat = "dh_engine_read_events"; }
                        }
                    }
                    check(firstRc, firstWhat);
                }

                static std::mutex& registryMutex() { static std::mutex m; return m; }
                static unsigned int& capacityRef() { static unsigned int c = 0; return c; }
                typedef std::tuple<int, int, int, unsigned int, int, unsigned int> Key;
                static std::map<Key, std::weak_ptr<SharedEngine>>& registry() { static std::map<Key, std::weak_ptr<SharedEngine>> r; return r; }

                Stage stage;
                Engine engine;
                std::mutex mutex;
                std::vector<Slot> slots;
                std::vector<unsigned char> staging;              // [capacity][chunk] floats or symbols
                std::vector<uint32_t> counts;
                void* dSyms = nullptr; void* dCounts = nullptr;
                unsigned int attached = 0, pendingCount = 0;
                std::atomic<unsigned long> ticks { 0 };      // (read by ticksTotal() without the bank's mutex)
        };

    }
}
