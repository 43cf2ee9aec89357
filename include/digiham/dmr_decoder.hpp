// dmr_decoder.hpp -- Digiham::Dmr::Decoder (reference: include/dmr_decoder.hpp:9-17).
#pragma once
#include "decoder.hpp"
