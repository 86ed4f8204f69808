// stencil_iso3dfd_k4.hip -- kernel instantiations for solution 'iso3dfd', group 4 (split over several
// translation units so that hipcc compiles the tile shapes in parallel).
#include "gen/iso3dfd_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {
using namespace ykh_gen_iso3dfd;
void iso3dfd_variants_k4(PartImpl& p) {
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 3, 2, 4, 0>());   // halos after row 0
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 5, 2, 4, 0>());   // halos at the end
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 1, 2, 2, 0>());    // LDS batches of 2
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 17, 2, 2, 0>());   // + operands two planes ahead
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 9, 2, 2, 0>());    // + star planes two planes ahead
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_TRIP, 9, 2, 2, 0>());    // + queue renaming within the 2-plane trip
    p.variants.push_back(starlin_variant_planned<part_1, 4, 32, 16, 2, ROT_TRIP2, 9, 2, 2>());     // (+ twin for planned launches)
    // + cheap tail planes (_tl): the main loop is the plain shape's, the block's last XH planes take a path without halo / slab / barrier
#endif
    p.variants.push_back(starlin_variant_planned<part_1, 4, 32, 16, 2, ROT_TRIP2, 9 | 64, 2, 2>());      // 256 VGPRs, no scratch
    p.variants.push_back(starlin_variant_planned<part_1, 4, 32, 16, 2, ROT_TRIP, 9 | 64, 2, 2>());
#ifdef YKH_PROFILING      // round 5 A/B: the default shape with the XCD lock-step every 16 / 64 planes (ykh_starlin.hpp xcd_sync)
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_TRIP2, 9 | 64 | (5 << 8), 2, 2>());
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_TRIP2, 9 | 64 | (7 << 8), 2, 2>());
#endif
#ifdef YKH_PROFILING      // write-through output stores: measured 1-12 % slower, profiles/r4_wt
    p.variants.push_back(starlin_variant_planned<part_1, 4, 32, 16, 2, ROT_TRIP2, 9 | 64 | 128, 2, 2>());
#endif
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 9 | 64, 2, 2>());   // + 4-plane trips
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 29, 2, 2, 0>());   // + both, halos at the end
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 13, 2, 2, 0>());   // planes two ahead, halos at the end
    p.variants.push_back(starlin_variant<part_1, 4, 64, 8, 2, ROT_MOVE, 9, 2, 2, 0>());     // 256x16 tile, planes two ahead
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 1, ROT_UNROLL, 1, 2, 4, 0>());  // 128x16, queue rotation by renaming
    // 256-thread workgroups, two per CU: independent barriers, one computes while the other waits on memory
    p.variants.push_back(starlin_variant<part_1, 4, 32, 8, 2, ROT_MOVE, 1, 2, 4, 0>());    // tile 128x16
    p.variants.push_back(starlin_variant<part_1, 4, 64, 8, 2, ROT_MOVE, 1, 2, 4, 0>());
    p.variants.push_back(starlin_variant<part_1, 4, 16, 32, 2, ROT_MOVE, 1, 2, 4, 0>());
#endif
}
}  // namespace ykh
