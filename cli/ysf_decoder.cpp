// ysf_decoder -- dibits in, codec frames out, metadata lines to --fifo (reference: src/ysf_decoder/ysf_cli.cpp:5-17)
#include "digiham/cli.hpp"
#include "digiham/ysf_decoder.hpp"

namespace {
    class Tool: public Digiham::DecoderCli {
        protected:
            std::string getName() override { return "ysf_decoder"; }
            Csdr::Module<unsigned char, unsigned char>* buildModule() override {
                auto module = new Digiham::Ysf::Decoder();
                if (metaWriter) module->setMetaWriter(metaWriter);
                return module;
            }
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
