// cli.hpp -- the stdin -> module -> stdout driver behind the command line tools (cli/*.cpp), so that
//     rtl_fm ... | rrc_filter | gfsk_demodulator | dmr_decoder --fifo meta | mbe_synthesizer ...
// from examples/dmr-decoder.sh runs on the MI355X engine unchanged: the reference's tool names, option letters and wire
// formats (raw float32 / uint8 / int16 on the pipes, `k:v;k:v\n` lines on the fifo; src/lib/cli.cpp:19-137 and the
// src/*/*_cli.cpp files are the specification of that surface).
//
// A tool is a Cli<T, U> subclass that names itself, builds its Csdr::Module<T, U>, and DECLARES its switches as rows
// of an option table (letter, long name, value placeholder, help text, action).  Parsing, the usage text and the
// --help / --version rows all derive from that one table.
#pragma once

#include <getopt.h>

#include <algorithm>
#include <cstdio>
#include <functional>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "csdr_compat.hpp"
#include "meta.hpp"
#include "../digiham_amd.h"

namespace Digiham {

    struct CliOption {
        char letter;
        const char* name;
        const char* value;                          // placeholder of the argument in the usage text; nullptr = a plain flag
        const char* help;
        std::function<bool(const char*)> action;    // false = stop (after --help / --version): main() returns 0
    };

    template <typename T, typename U>
    class Cli {
        public:
            virtual ~Cli() = default;
            int main(int argc, char** argv) {
                std::vector<CliOption> table = {
                    { 'h', "help", nullptr, "show this message", [this, &table] (const char*) { std::cerr << usage(table); return false; } },
                    { 'v', "version", nullptr, "print version and exit", [this] (const char*) { std::cout << banner() << "\n"; return false; } },
                };
                declareOptions(table);
                if (!parse(argc, argv, table)) return 0;
                try {
                    return pump();
                } catch (const std::exception& e) {     // no MI355X, library missing, ...: fail loudly
                    std::cerr << getName() << ": " << e.what() << "\n";
                    return 1;
                }
            }
        protected:
            virtual std::string getName() = 0;
            virtual Csdr::Module<T, U>* buildModule() = 0;
            virtual void declareOptions(std::vector<CliOption>&) {}
            virtual void releaseModule(Csdr::Module<T, U>* module) { delete module; }       // end of the pipe: the module goes away
            // The reference reads up to 128 items per fread() into a 1024-item ring (cli.cpp:102-106); one GPU launch
            // per 128 samples would be all overhead, so up to 4096 items are taken at a time.  A pipe delivers what is
            // there, so latency is unchanged; the outputs do not depend on how the stream is cut.
            virtual size_t readSize() { return 4096; }
            virtual size_t ringbufferSize() { return 16384; }
        private:
            std::string banner() { return getName() + " version " + dh_version(); }
            std::string usage(const std::vector<CliOption>& table) {
                std::string text = banner() + "\n\nUsage: " + getName() + " [options]\n\nAvailable options:\n";
                for (const CliOption& o : table) {
                    std::string left = std::string(" -") + o.letter + ", --" + o.name;
                    left.resize(std::max<size_t>(left.size() + 1, 21), ' ');
                    text += left + o.help + "\n";
                }
                return text;
            }
            bool parse(int argc, char** argv, const std::vector<CliOption>& table) {
                std::string letters;
                std::vector<struct option> longs;
                for (const CliOption& o : table) {
                    letters += o.letter;
                    if (o.value != nullptr) letters += ':';
                    longs.push_back({ o.name, o.value != nullptr ? required_argument : no_argument, nullptr, o.letter });
                }
                longs.push_back({ nullptr, 0, nullptr, 0 });
                for (int c; (c = getopt_long(argc, argv, letters.c_str(), longs.data(), nullptr)) != -1;) {
                    const auto row = std::find_if(table.begin(), table.end(), [c] (const CliOption& o) { return o.letter == c; });
                    if (row == table.end()) { std::cerr << usage(table); return false; }      // unknown switch: getopt already complained
                    if (!row->action(optarg)) return false;
                }
                return true;
            }
            int pump() {
                Csdr::Ringbuffer<T> ring(ringbufferSize());
                std::unique_ptr<Csdr::Module<T, U>, std::function<void(Csdr::Module<T, U>*)>> module(buildModule(), [this] (Csdr::Module<T, U>* m) { releaseModule(m); });
                Csdr::RingbufferReader<T> reader(&ring);
                Csdr::StdoutWriter<U> writer;
                module->setReader(&reader);
                module->setWriter(&writer);
                for (;;) {
                    const size_t got = fread(ring.getWritePointer(), sizeof(T), std::min(ring.writeable(), readSize()), stdin);
                    if (got == 0) break;                // end of the pipe
                    ring.advance(got);
                    while (module->canProcess()) module->process();
                }
                return 0;
            }
    };

    // decoders additionally take --fifo: where their metadata lines go (src/lib/cli.cpp:108-137)
    class DecoderCli: public Cli<unsigned char, unsigned char> {
        protected:
            void declareOptions(std::vector<CliOption>& table) override {
                table.push_back({ 'f', "fifo", "path", "send metadata to this file", [this] (const char* path) {
                    std::cerr << "meta fifo: " << path << "\n";
                    metaWriter = new FileMetaWriter(fopen(path, "w"));
                    return true;
                } });
            }
            MetaWriter* metaWriter = nullptr;           // handed to (and then owned by) the decoder in buildModule()
    };

}
