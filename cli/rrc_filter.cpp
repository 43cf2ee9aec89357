// rrc_filter -- float32 in, float32 out (reference: src/rrc_filter/rrc_filter_cli.cpp:6-44)
#include "digiham/cli.hpp"
#include "digiham/rrc_filter.hpp"

namespace {
    class Tool: public Digiham::Cli<float, float> {
        protected:
            std::string getName() override { return "rrc_filter"; }
            void declareOptions(std::vector<Digiham::CliOption>& table) override {
                table.push_back({ 'n', "narrow", nullptr, "use narrow (6.25kHz) filter version (default: wide / 12.5kHz)",
                                  [this] (const char*) { narrow = true; return true; } });
            }
            Csdr::Module<float, float>* buildModule() override {
                if (narrow) return new Digiham::RrcFilter::NarrowRrcFilter();
                return new Digiham::RrcFilter::WideRrcFilter();
            }
        private:
            bool narrow = false;
    };
}

int main(int argc, char** argv) { Tool tool; return tool.main(argc, argv); }
