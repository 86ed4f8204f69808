// stencil_3axis_k4.hip -- kernel instantiations for solution '3axis', group 4: register queues renamed inside 2-/4-plane
// trips (ROT_TRIP / ROT_TRIP2, ykh_starlin.hpp) on the two shipping tile shapes, with and without planes two ahead.
#include "gen/3axis_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"
namespace ykh {
using namespace ykh_gen_3axis;
void s3axis_variants_k4(PartImpl& p) {
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_TRIP, 1, 2, 4>());      // tile 128x32 (the large-grid shape)
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_TRIP2, 1, 2, 4>());
    // (planes two ahead on this shape: 256 VGPRs + 44 ... 92 B of scratch per lane with either rotation: not instantiated)
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_TRIP2, 1, 2, 4>());    // tile 64x32 (the default shape)
    p.variants.push_back(starlin_variant<part_1, 2, 64, 16, 2, ROT_MOVE, 1, 4, 4>());     // tile 128x32 on 1024 threads: 120 VGPRs, 4 waves per SIMD
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 2, ROT_MOVE, 1, 4, 4>());      // tile 128x16, 126 VGPRs: two workgroups per CU
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_TRIP2, 9, 2, 4>());
    // round 3 (fp64 is latency-bound at 2 waves per SIMD): more waves instead of more rows per thread --
    // one row per thread on 1024 threads, tile 128 x 16 (94-102 VGPRs: 4-5 waves per SIMD), with queue renaming and planes two ahead
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_TRIP2, 1, 2, 2>());     // the large-grid shape, LDS batches of 2 (246 VGPRs)
    p.variants.push_back(starlin_variant<part_1, 2, 64, 16, 1, ROT_TRIP2, 9, 4, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 64, 16, 1, ROT_TRIP2, 1, 4, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 32, 32, 1, ROT_TRIP2, 9, 4, 4>());    // tile 64 x 32 on 1024 threads
    p.variants.push_back(starlin_variant<part_1, 2, 64, 16, 1, ROT_TRIP2, 25, 4, 4>());   // + operands two planes ahead
#endif
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_UNROLL, 1 | 64, 2, 4>());   // the default shape + cheap tail planes (_tl)
#ifdef YKH_PROFILING      // round 5 A/B: + XCD lock-step every 16 / 32 / 64 planes (512^3: x-chunks of 128 planes)
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_UNROLL, 1 | 64 | (5 << 8), 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_UNROLL, 1 | 64 | (6 << 8), 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_UNROLL, 1 | 64 | (7 << 8), 2, 4>());
#endif
#ifdef YKH_PROFILING      // (halo-late A/B, see stencil_3axis_k2.hip)
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_UNROLL, 1 | 64 | 2, 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_UNROLL, 1 | 64 | 4, 2, 4>());
#endif
#ifdef YKH_PROFILING      // write-through output stores: measured 1-12 % slower, profiles/r4_wt
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_UNROLL, 1 | 64 | 128, 2, 4>());
#endif
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | 64, 2, 4>());      // the large-grid shape + cheap tail planes
#endif
}
}  // namespace ykh
