// Line factorizations used by the pruned pipeline (padded length L = 2n).
#pragma once
#include "fft_core.hpp"
namespace lsfc { namespace fft {
//                L    T   R0  R1  R2
using Cfg32   = Cfg<32,   4,  8,  4>;
using Cfg64   = Cfg<64,   8,  8,  8>;
using Cfg128  = Cfg<128, 16,  8,  4, 4>;
using Cfg256  = Cfg<256, 32,  4,  8, 8>;     // (round 3; 8.8.4 before: the 8.8 pair last lets the fused pass exchange through the lanes, 128^3 0.184 -> 0.178 ms)
using Cfg512  = Cfg<512, 64,  8,  8, 8>;
using Cfg1024 = Cfg<1024, 64, 16, 8, 8>;
using Cfg2048 = Cfg<2048, 128, 8, 16, 16>;   // (round 3; 16.16.8 before: the 16.16 pair last runs through the lanes in 4-line workgroups -- 2D n = 1024 58.2 -> 55.4 us, 1024^3 110 -> 107.5 ms)
// four-stage, 8 elements per thread: half the registers of Cfg1024 for one more LDS exchange.  Measured slower on
// MI355X (profiles/r01_experiment_e8_four_stage.log; again in round 3 with the persistent fused pass and its 4.4 exchange through
// the lanes: fused pass 5.0 against 4.63 ms at 512^3, x and y passes equal, profiles/r03_experiment_lane_exchange_radix4.log);
// kept as the tested instance of the 4-stage machinery.
using Cfg1024S = Cfg<1024, 128, 8, 8, 4, 4>;
// lines with one factor 3 (L = 3 * 2^k): the radix carrying it is the first one, 12 or 24 elements per thread
using Cfg48   = Cfg<48,    4, 12, 4>;
using Cfg96   = Cfg<96,    8, 12, 4, 2>;
using Cfg192  = Cfg<192,  16, 12, 4, 4>;
using Cfg384  = Cfg<384,  16, 24, 4, 4>;
// 768 points: 12 elements per thread in four stages (round 3; before: 24.8.4 on 32 threads).  Twice the threads per line put two
// waves on every SIMD instead of one, and the last exchange (4.4) runs through the lanes: 384^3 apply 5.90 -> 5.50 ms (fused pass
// 2.62 -> 2.34, yfwd 1.08 -> 0.93).  The same step on the 384-point line (12.4.4.2 on 32 threads) is a tie (192^3 0.74 ms
// either way) and was not taken (profiles/r03_experiment_lane_exchange_radix4.log)
using Cfg768  = Cfg<768,  64, 12, 4, 4, 4>;
using Cfg1536 = Cfg<1536, 64, 24, 8, 8>;     // (24.4.4.4 with two lane exchanges, as the 1280-point line: 768^3 49.4 -> 52.1 ms, not taken)
// lines with one factor 5 (L = 5 * 2^k): first radix 20 (10 at 640 points), 20 elements per thread
using Cfg80   = Cfg<80,    4, 20, 4>;
using Cfg160  = Cfg<160,   8, 20, 4, 2>;
using Cfg320  = Cfg<320,  16, 20, 4, 4>;
// 640 points: first radix 10 (two half-size butterflies per thread) and three radix-4 stages instead of 20.4.4.2 (round 3, same-box A/B of
// two builds: 320^3 apply 3.52 -> 3.38 ms, yfwd 0.60 -> 0.555, fused pass 1.65 -> 1.57); the same change is a loss on the 320-point line
// (10.4.4.2: 160^3 0.40 -> 0.46 ms) and a tie on the 160-point line, which stay (profiles/r03_experiment_lane_exchange_radix4.log)
using Cfg640  = Cfg<640,  32, 10, 4, 4, 4>;
using Cfg1280 = Cfg<1280, 64, 20, 4, 4, 4>;
}} // namespace
