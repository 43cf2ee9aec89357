// rrc_filter -- float32 in, float32 out (reference: src/rrc_filter/rrc_filter_cli.cpp:6-44)
#include "digiham/cli.hpp"
#include "digiham/rrc_filter.hpp"

namespace {
    class Cli: public Digiham::Cli<float, float> {
        protected:
            std::string getName() override { return "rrc_filter"; }
            std::stringstream getUsageString() override {
                std::stringstream result = Digiham::Cli<float, float>::getUsageString();
                result << " -n, --narrow        use narrow (6.25kHz) filter version (default: wide / 12.5kHz)\n";
                return result;
            }
            std::vector<struct option> getOptions() override {
                std::vector<struct option> options = Digiham::Cli<float, float>::getOptions();
                options.push_back({"narrow", no_argument, NULL, 'n'});
                return options;
            }
            bool receiveOption(int c, char* optarg) override {
                if (c == 'n') { narrow = true; return true; }
                return Digiham::Cli<float, float>::receiveOption(c, optarg);
            }
            Csdr::Module<float, float>* buildModule() override {
                if (narrow) return new Digiham::RrcFilter::NarrowRrcFilter();
                return new Digiham::RrcFilter::WideRrcFilter();
            }
        private:
            bool narrow = false;
    };
}

int main(int argc, char** argv) { Cli runner; return runner.main(argc, argv); }
