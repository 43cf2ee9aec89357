// dstar_decoder -- bits (fsk_demodulator -s 10) in, 9-byte AMBE voice frames out, metadata lines to --fifo (reference: src/dstar_decoder/dstar_cli.cpp, examples/dstar-decoder.sh)
#include "digiham/cli.hpp"
#include "digiham/dstar_decoder.hpp"

namespace {
    class Tool: public Digiham::DecoderCli {
        protected:
            std::string getName() override { return "dstar_decoder"; }
            Csdr::Module<unsigned char, unsigned char>* buildModule() override {
                auto module = new Digiham::DStar::Decoder();
                if (metaWriter) module->setMetaWriter(metaWriter);
                return module;
            }
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
