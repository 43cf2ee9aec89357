// pocsag_decoder.hpp -- Digiham::Pocsag::Decoder (reference: include/pocsag_decoder.hpp); see decoder.hpp
#pragma once
#include "decoder.hpp"
