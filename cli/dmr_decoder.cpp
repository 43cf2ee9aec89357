// dmr_decoder -- dibits in, 27-byte AMBE bursts out, metadata lines to --fifo, slot filter from --control-fifo
// (reference: src/dmr_decoder/dmr_cli.cpp:5-78)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <thread>

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
                FILE* fifo = fopen(path.c_str(), "r");
                if (fifo == nullptr) return;
                char line[2];
                while (!quit && !ferror(fifo)) {
                    if (fread(line, sizeof(char), 2, fifo) < 2) {
                        clearerr(fifo);
                        std::this_thread::sleep_for(std::chrono::milliseconds(20));
                        continue;
                    }
                    if (line[1] != '\n') continue;
                    std::lock_guard<std::mutex> lock(decoderMutex);
                    filter = line[0] - '0';
                    if (decoder != nullptr) decoder->setSlotFilter((unsigned char) filter);
                }
                fclose(fifo);
            }
            std::mutex decoderMutex;                    // guards `decoder` and `filter` (control thread vs. main thread)
            Digiham::Dmr::Decoder* decoder = nullptr;
            int filter = -1;
            std::atomic<bool> quit { false };
            std::thread control;
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
