// gfsk_demodulator -- float32 in, one dibit per byte out (reference: src/gfsk_demodulator/gfsk_demodulator_cli.cpp:5-39)
#include <cstdlib>

#include "digiham/cli.hpp"
#include "digiham/gfsk_demodulator.hpp"

namespace {
    class Tool: public Digiham::Cli<float, unsigned char> {
        protected:
            std::string getName() override { return "gfsk_demodulator"; }
            void declareOptions(std::vector<Digiham::CliOption>& table) override {
                table.push_back({ 's', "samples", "n", "samples per symbol ( = audio sample rate / symbol rate; default: 10)",
                                  [this] (const char* v) { samplesPerSymbol = (unsigned int) std::strtoul(v, nullptr, 10); return true; } });
            }
            Csdr::Module<float, unsigned char>* buildModule() override { return new Digiham::Fsk::GfskDemodulator(samplesPerSymbol); }
        private:
            unsigned int samplesPerSymbol = 10;
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
