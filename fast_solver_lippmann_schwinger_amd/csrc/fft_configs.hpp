// Line factorizations used by the pruned pipeline (padded length L = 2n).
#pragma once
#include "fft_core.hpp"
namespace lsfc { namespace fft {
//                L    T   R0  R1  R2
using Cfg32   = Cfg<32,   4,  8,  4>;
using Cfg64   = Cfg<64,   8,  8,  8>;
using Cfg128  = Cfg<128, 16,  8,  4, 4>;
using Cfg256  = Cfg<256, 32,  8,  8, 4>;
using Cfg512  = Cfg<512, 64,  8,  8, 8>;
using Cfg1024 = Cfg<1024, 64, 16, 8, 8>;
using Cfg2048 = Cfg<2048, 128, 16, 16, 8>;
// four-stage, 8 elements per thread: half the registers of Cfg1024 for one more LDS exchange.  Measured slower on
// MI355X (profiles/r01_experiment_e8_four_stage.log); kept as the tested instance of the 4-stage machinery.
using Cfg1024S = Cfg<1024, 128, 8, 8, 4, 4>;
}} // namespace
