// fsk_demodulator.hpp -- Digiham::Fsk::FskDemodulator (reference: include/fsk_demodulator.hpp:12-33).
#pragma once
#include "gfsk_demodulator.hpp"
