// ysf_decoder.hpp -- Digiham::Ysf::Decoder (reference: include/ysf_decoder.hpp:9-12).
#pragma once
#include "decoder.hpp"
