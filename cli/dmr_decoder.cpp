// dmr_decoder -- dibits in, 27-byte AMBE bursts out, metadata lines to --fifo, slot filter from --control-fifo
// (reference: src/dmr_decoder/dmr_cli.cpp:5-78)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <poll.h>
#include <unistd.h>

#include "digiham/cli.hpp"
#include "digiham/dmr_decoder.hpp"

namespace {
    class Tool: public Digiham::DecoderCli {
        public:
            ~Tool() override {
                quit = true;
                if (control.joinable()) control.join();
            }
        protected:
            std::string getName() override { return "dmr_decoder"; }
            void declareOptions(std::vector<Digiham::CliOption>& table) override {
                Digiham::DecoderCli::declareOptions(table);
                table.push_back({ 'c', "control-fifo", "path", "read control messages from this file", [this] (const char* path) {
                    if (control.joinable()) return true;        // a repeated -c: the first one stands (assigning to a running std::thread terminates)
                    control = std::thread([this, p = std::string(path)] () { controlLoop(p); });
                    return true;
                } });
            }
            Csdr::Module<unsigned char, unsigned char>* buildModule() override {
                auto module = new Digiham::Dmr::Decoder();
                if (metaWriter) module->setMetaWriter(metaWriter);
                std::lock_guard<std::mutex> lock(decoderMutex);
                decoder = module;
                if (filter >= 0) module->setSlotFilter((unsigned char) filter);     // a command that arrived before the module existed
                return module;
            }
            void releaseModule(Csdr::Module<unsigned char, unsigned char>* module) override {
                { std::lock_guard<std::mutex> lock(decoderMutex); decoder = nullptr; }  // the control thread must not reach it any more
                delete module;
            }
        private:
            // "<digit>\n" = slot filter (bit 0: slot 1, bit 1: slot 2).  The fifo is opened here, not while the options are
            // parsed: opening a fifo for reading blocks until there is a writer.  When the writer goes away (EOF) the loop
            // keeps listening -- every later `echo N > control_fifo` must still arrive (dmr_cli.cpp:57-78) -- with a short
            // sleep instead of the reference's busy spin.
            void controlLoop(const std::string& path) {
                // non-blocking open + poll with a timeout: neither a fifo without a writer nor an idle writer can keep this thread
                // from seeing `quit` (with fopen / fread the tool hung at exit, as the reference does: dmr_cli.cpp:57-78)
                const int fd = open(path.c_str(), O_RDONLY | O_NONBLOCK);
                if (fd < 0) return;
                char prev = 0;
                bool have_prev = false;
                while (!quit) {
                    struct pollfd p = { fd, POLLIN, 0 };
                    const int r = poll(&p, 1, 50);
                    if (r <= 0 || !(p.revents & POLLIN)) {
                        if (p.revents & (POLLHUP | POLLERR)) std::this_thread::sleep_for(std::chrono::milliseconds(20));     // no writer at the moment: keep listening
                        continue;
                    }
                    char buf[64];
                    const ssize_t got = read(fd, buf, sizeof(buf));
                    if (got <= 0) { std::this_thread::sleep_for(std::chrono::milliseconds(20)); continue; }
                    for (ssize_t i = 0; i < got; i++) {          // "<digit>\n" pairs, as the reference reads them two bytes at a time
                        if (!have_prev) { prev = buf[i]; have_prev = true; continue; }
                        have_prev = false;
                        if (buf[i] != '\n') continue;
                        std::lock_guard<std::mutex> lock(decoderMutex);
                        filter = prev - '0';
                        if (decoder != nullptr) decoder->setSlotFilter((unsigned char) filter);
                    }
                }
                close(fd);
            }
            std::mutex decoderMutex;                    // guards `decoder` and `filter` (control thread vs. main thread)
            Digiham::Dmr::Decoder* decoder = nullptr;
            int filter = -1;
            std::atomic<bool> quit { false };
            std::thread control;
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
