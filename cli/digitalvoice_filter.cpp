// digitalvoice_filter -- int16 in, int16 out (reference: src/digitalvoice_filter/digitalvoice_filter_cli.cpp:6-16)
#include "digiham/cli.hpp"
#include "digiham/digitalvoice_filter.hpp"

namespace {
    class Tool: public Digiham::Cli<short, short> {
        protected:
            std::string getName() override { return "digitalvoice_filter"; }
            Csdr::Module<short, short>* buildModule() override { return new Digiham::DigitalVoice::DigitalVoiceFilter(); }
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
