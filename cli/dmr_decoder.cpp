// dmr_decoder -- dibits in, 27-byte AMBE bursts out, metadata lines to --fifo, slot filter from --control-fifo
// (reference: src/dmr_decoder/dmr_cli.cpp:5-78)
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "digiham/cli.hpp"
#include "digiham/dmr_decoder.hpp"

namespace {
    class Cli: public Digiham::DecoderCli {
        public:
            ~Cli() override {
                if (fifo != nullptr) fclose(fifo);
                if (fifoReader != nullptr) { fifoReader->join(); delete fifoReader; }
            }
        protected:
            std::string getName() override { return "dmr_decoder"; }
            Csdr::Module<unsigned char, unsigned char>* buildModule() override {
                decoder = new Digiham::Dmr::Decoder();
                if (metaWriter) decoder->setMetaWriter(metaWriter);
                if (pendingFilter >= 0) decoder->setSlotFilter((unsigned char) pendingFilter);
                return decoder;
            }
            std::stringstream getUsageString() override {
                std::stringstream ss = Digiham::DecoderCli::getUsageString();
                ss << " -c, --control-fifo  read control messages from this file\n";
                return ss;
            }
            std::vector<struct option> getOptions() override {
                std::vector<struct option> options = Digiham::DecoderCli::getOptions();
                options.push_back({"control-fifo", required_argument, NULL, 'c'});
                return options;
            }
            bool receiveOption(int c, char* optarg) override {
                if (c != 'c') return Digiham::DecoderCli::receiveOption(c, optarg);
                fifo = fopen(optarg, "r");
                if (fifo != nullptr) fifoReader = new std::thread([this] () { fifoLoop(); });
                return true;
            }
        private:
            // "<digit>\n" = slot filter (bit 0: slot 1, bit 1: slot 2), dmr_cli.cpp:57-78
            void fifoLoop() {
                char line[2];
                while (fifo != nullptr && !ferror(fifo) && fread(line, sizeof(char), 2, fifo) >= 2) {
                    if (line[1] != '\n') continue;
                    const int filter = line[0] - '0';
                    if (decoder != nullptr) decoder->setSlotFilter((unsigned char) filter);
                    else pendingFilter = filter;          // the option is parsed before the module exists
                }
            }
            Digiham::Dmr::Decoder* decoder = nullptr;
            FILE* fifo = nullptr;
            std::thread* fifoReader = nullptr;
            int pendingFilter = -1;
    };
}

int main(int argc, char** argv) { Cli runner; return runner.main(argc, argv); }
