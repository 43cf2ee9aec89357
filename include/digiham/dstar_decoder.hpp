// dstar_decoder.hpp -- Digiham::DStar::Decoder (reference: include/dstar_decoder.hpp, src/dstar_decoder/dstar_decoder.cpp:7-9);
// see decoder.hpp
#pragma once
#include "decoder.hpp"
