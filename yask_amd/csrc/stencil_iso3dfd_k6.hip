// stencil_iso3dfd_k6.hip -- kernel instantiations for solution 'iso3dfd', group 6: the shipping 128x32 tile on 1024 threads (one
// row per thread, <= 128 VGPRs, 4 waves per SIMD instead of 2: twice the independent instruction streams per CU).
#include "gen/iso3dfd_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {
using namespace ykh_gen_iso3dfd;
void iso3dfd_variants_k6(PartImpl& p) {
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(starlin_variant<part_1, 4, 32, 32, 1, ROT_MOVE, 1, 4, 2, 0>());     // 124 VGPRs
    p.variants.push_back(starlin_variant<part_1, 4, 32, 32, 1, ROT_TRIP2, 1, 4, 2, 0>());    // 128 VGPRs (planes two ahead: spills)
    // (both measured slower than the 512-thread default: 1024^3 3.08 / 2.96 vs 2.84 ms, 512^3 0.391 / 0.393 vs 0.383 ms; gpurun_out/r03o)
    // the 256x16 tile with the queue renaming of the default (in `_m` form it beat the 128x32 `_m`: 2.876 vs 2.941 ms)
    p.variants.push_back(starlin_variant<part_1, 4, 64, 8, 2, ROT_TRIP, 9, 2, 2, 0>());      // 242 VGPRs
    p.variants.push_back(starlin_variant<part_1, 4, 64, 8, 2, ROT_TRIP2, 9, 2, 2, 0>());     // 256 VGPRs, no spill
#endif
}
}  // namespace ykh
