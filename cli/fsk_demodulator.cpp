// fsk_demodulator -- float32 in, one bit per byte out (reference: src/fsk_demodulator/fsk_demodulator_cli.cpp:5-45)
#include <cstdlib>

#include "digiham/cli.hpp"
#include "digiham/fsk_demodulator.hpp"

namespace {
    class FskCli: public Digiham::Cli<float, unsigned char> {
        protected:
            std::string getName() override { return "fsk_demodulator"; }
            Csdr::Module<float, unsigned char>* buildModule() override { return new Digiham::Fsk::FskDemodulator(samplesPerSymbol, invert); }
            std::stringstream getUsageString() override {
                std::stringstream result = Digiham::Cli<float, unsigned char>::getUsageString();
                result << " -s, --samples       samples per symbol ( = audio sample rate / symbol rate; default: 40)\n"
                       << " -i, --invert        invert bits (used e.g in pocsag)\n";
                return result;
            }
            std::vector<struct option> getOptions() override {
                std::vector<struct option> options = Digiham::Cli<float, unsigned char>::getOptions();
                options.push_back({"samples", required_argument, NULL, 's'});
                options.push_back({"invert", no_argument, NULL, 'i'});
                return options;
            }
            bool receiveOption(int c, char* optarg) override {
                switch (c) {
                    case 's': samplesPerSymbol = (unsigned int) std::strtoul(optarg, NULL, 10); return true;
                    case 'i': invert = true; return true;
                    default: return Digiham::Cli<float, unsigned char>::receiveOption(c, optarg);
                }
            }
        private:
            unsigned int samplesPerSymbol = 40;
            bool invert = false;
    };
}

int main(int argc, char** argv) { FskCli runner; return runner.main(argc, argv); }
