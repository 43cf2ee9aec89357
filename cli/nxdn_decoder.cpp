// nxdn_decoder -- dibits in, 18-byte AMBE voice blocks out, metadata lines to --fifo (reference: src/nxdn_decoder/nxdn_cli.cpp:5-19)
#include "digiham/cli.hpp"
#include "digiham/nxdn_decoder.hpp"

namespace {
    class Tool: public Digiham::DecoderCli {
        protected:
            std::string getName() override { return "nxdn_decoder"; }
            Csdr::Module<unsigned char, unsigned char>* buildModule() override {
                auto module = new Digiham::Nxdn::Decoder();
                if (metaWriter) module->setMetaWriter(metaWriter);
                return module;
            }
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
