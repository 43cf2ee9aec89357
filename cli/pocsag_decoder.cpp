// pocsag_decoder -- bits in (fsk_demodulator -i -s 40), `address:<n>;message:<text>` lines out
// (reference: src/pocsag_decoder/pocsag_cli.cpp:5-19, examples/pocsag-decoder.sh)
#include "digiham/cli.hpp"
#include "digiham/pocsag_decoder.hpp"

namespace {
    class Tool: public Digiham::Cli<unsigned char, unsigned char> {
        protected:
            std::string getName() override { return "pocsag_decoder"; }
            Csdr::Module<unsigned char, unsigned char>* buildModule() override { return new Digiham::Pocsag::Decoder(); }
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
