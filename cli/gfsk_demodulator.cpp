// gfsk_demodulator -- float32 in, one dibit per byte out (reference: src/gfsk_demodulator/gfsk_demodulator_cli.cpp:5-39)
#include <cstdlib>

#include "digiham/cli.hpp"
#include "digiham/gfsk_demodulator.hpp"

namespace {
    class GfskCli: public Digiham::Cli<float, unsigned char> {
        protected:
            std::string getName() override { return "gfsk_demodulator"; }
            Csdr::Module<float, unsigned char>* buildModule() override { return new Digiham::Fsk::GfskDemodulator(samplesPerSymbol); }
            std::stringstream getUsageString() override {
                std::stringstream result = Digiham::Cli<float, unsigned char>::getUsageString();
                result << " -s, --samples       samples per symbol ( = audio sample rate / symbol rate; default: 10)\n";
                return result;
            }
            std::vector<struct option> getOptions() override {
                std::vector<struct option> options = Digiham::Cli<float, unsigned char>::getOptions();
                options.push_back({"samples", required_argument, NULL, 's'});
                return options;
            }
            bool receiveOption(int c, char* optarg) override {
                if (c == 's') { samplesPerSymbol = (unsigned int) std::strtoul(optarg, NULL, 10); return true; }
                return Digiham::Cli<float, unsigned char>::receiveOption(c, optarg);
            }
        private:
            unsigned int samplesPerSymbol = 10;
    };
}

int main(int argc, char** argv) { GfskCli runner; return runner.main(argc, argv); }
