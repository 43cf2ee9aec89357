// cli.hpp -- Digiham::Cli<T, U> / Digiham::DecoderCli: the stdin -> module -> stdout driver of the reference's
// command line tools (src/lib/cli.hpp:14-37, src/lib/cli.cpp:19-137), so that `rrc_filter | gfsk_demodulator |
// dmr_decoder --fifo meta` from examples/dmr-decoder.sh runs on the MI355X engine unchanged: same option letters,
// same wire formats (raw float32 / uint8 / int16 on the pipes, `k:v;k:v\n` lines on the fifo).
#pragma once

#include <getopt.h>

#include <algorithm>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "csdr_compat.hpp"
#include "meta.hpp"
#include "../digiham_amd.h"

#ifndef BUF_SIZE
#define BUF_SIZE 128
#endif
#ifndef RINGBUFFER_SIZE
#define RINGBUFFER_SIZE 1024
#endif

namespace Digiham {

    template <typename T, typename U>
    class Cli {
        public:
            Cli(): ringbuffer(new Csdr::Ringbuffer<T>(ringbufferSize())) {}
            virtual ~Cli() { delete ringbuffer; }
            int main(int argc, char** argv) {
                if (!parseOptions(argc, argv)) return 0;
                int rc = 0;
                try {
                    auto module = buildModule();
                    module->setReader(new Csdr::RingbufferReader<T>(ringbuffer));
                    module->setWriter(new Csdr::StdoutWriter<U>());
                    while (read()) {
                        while (module->canProcess()) module->process();
                    }
                    delete module;
                } catch (const std::exception& e) {         // no MI355X, library missing, ...: fail loudly
                    std::cerr << getName() << ": " << e.what() << "\n";
                    rc = 1;
                }
                return rc;
            }
        protected:
            virtual std::string getName() = 0;
            virtual Csdr::Module<T, U>* buildModule() = 0;
            // The reference reads up to BUF_SIZE = 128 items per fread() into a 1024-item ring (cli.cpp:102-106); a GPU
            // launch per 128 samples would be all overhead, so the tools here read up to 4096 items at a time.  Pipes
            // deliver what is there, so latency is unchanged; the outputs do not depend on how the stream is cut.
            virtual size_t readSize() { return 4096; }
            virtual size_t ringbufferSize() { return 16384; }
            virtual std::vector<struct option> getOptions() {
                return { {"version", no_argument, NULL, 'v'}, {"help", no_argument, NULL, 'h'} };
            }
            virtual std::stringstream getUsageString() {
                std::stringstream result;
                result << getName() << " version " << dh_version() << "\n\n"
                       << "Usage: " << getName() << " [options]\n\n"
                       << "Available options:\n"
                       << " -h, --help          show this message\n"
                       << " -v, --version       print version and exit\n";
                return result;
            }
            virtual void printVersion() { std::cout << getName() << " version " << dh_version() << "\n"; }
            virtual bool parseOptions(int argc, char** argv) {
                std::vector<struct option> long_options = getOptions();
                std::string short_options;
                for (const auto& opt : long_options) {
                    short_options += (char) opt.val;
                    if (opt.has_arg == required_argument) short_options += ":";
                }
                long_options.push_back({ NULL, 0, NULL, 0 });
                int c;
                while ((c = getopt_long(argc, argv, short_options.c_str(), long_options.data(), NULL)) != -1) {
                    if (!receiveOption(c, optarg)) return false;
                }
                return true;
            }
            virtual bool receiveOption(int c, char* optarg) {
                (void) optarg;
                switch (c) {
                    case 'v':
                        printVersion();
                        return false;
                    case 'h':
                    default:
                        std::cerr << getUsageString().str();
                        return false;
                }
            }
        private:
            bool read() {
                const size_t r = fread(ringbuffer->getWritePointer(), sizeof(T), std::min(ringbuffer->writeable(), readSize()), stdin);
                ringbuffer->advance(r);
                return r > 0;
            }
            Csdr::Ringbuffer<T>* ringbuffer;
    };

    class DecoderCli: public Cli<unsigned char, unsigned char> {       // src/lib/cli.cpp:108-137
        protected:
            std::stringstream getUsageString() override {
                std::stringstream result = Cli<unsigned char, unsigned char>::getUsageString();
                result << " -f, --fifo          send metadata to this file\n";
                return result;
            }
            std::vector<struct option> getOptions() override {
                std::vector<struct option> options = Cli<unsigned char, unsigned char>::getOptions();
                options.push_back({"fifo", required_argument, NULL, 'f'});
                return options;
            }
            bool receiveOption(int c, char* optarg) override {
                switch (c) {
                    case 'f': {
                        std::cerr << "meta fifo: " << optarg << "\n";
                        metaWriter = new FileMetaWriter(fopen(optarg, "w"));
                        break;
                    }
                    default:
                        return Cli<unsigned char, unsigned char>::receiveOption(c, optarg);
                }
                return true;
            }
            MetaWriter* metaWriter = nullptr;
    };

}
