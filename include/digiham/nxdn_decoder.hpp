// nxdn_decoder.hpp -- Digiham::Nxdn::Decoder (reference: include/nxdn_decoder.hpp:9-14); see decoder.hpp
#pragma once
#include "decoder.hpp"
