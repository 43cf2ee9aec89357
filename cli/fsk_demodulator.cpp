// fsk_demodulator -- float32 in, one bit per byte out (reference: src/fsk_demodulator/fsk_demodulator_cli.cpp:5-45)
#include <cstdlib>

#include "digiham/cli.hpp"
#include "digiham/fsk_demodulator.hpp"

namespace {
    class Tool: public Digiham::Cli<float, unsigned char> {
        protected:
            std::string getName() override { return "fsk_demodulator"; }
            void declareOptions(std::vector<Digiham::CliOption>& table) override {
                table.push_back({ 's', "samples", "n", "samples per symbol ( = audio sample rate / symbol rate; default: 40)",
                                  [this] (const char* v) { samplesPerSymbol = (unsigned int) std::strtoul(v, nullptr, 10); return true; } });
                table.push_back({ 'i', "invert", nullptr, "invert bits (used e.g in pocsag)", [this] (const char*) { invert = true; return true; } });
            }
            Csdr::Module<float, unsigned char>* buildModule() override { return new Digiham::Fsk::FskDemodulator(samplesPerSymbol, invert); }
        private:
            unsigned int samplesPerSymbol = 40;
            bool invert = false;
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
