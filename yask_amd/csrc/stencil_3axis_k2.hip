// stencil_3axis_k2.hip -- more tile shapes for '3axis'.
#include "gen/3axis_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"
namespace ykh {
using namespace ykh_gen_3axis;
void s3axis_variants_k2(PartImpl& p) {
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1, 2, 4>());
    // + lock-step of an XCD's workgroups every 64 planes: 1024^3 4.1 % faster, FETCH_SIZE -4.4 %, bit-identical (profiles/r5_3axis_lockstep)
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | (7 << 8), 2, 4>());
#ifdef YKH_PROFILING      // round 4 A/B: halo vectors of the next plane requested LATE (after row 0 / at the end of the iteration): 4-24 % slower, FETCH unchanged (profiles/r4_3axis_fetch)
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | 2, 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | 4, 2, 4>());
#endif
#ifdef YKH_PROFILING      // round 5 A/B: soft lock-step of an XCD's workgroups every 1 ... 64 planes (profiles/r5_3axis_lockstep)
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | (1 << 8), 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | (2 << 8), 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | (3 << 8), 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | (4 << 8), 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | (5 << 8), 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | (6 << 8), 2, 4>());
#endif
#ifdef YKH_PROFILING      // write-through output stores: measured 1-12 % slower, profiles/r4_wt
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 4, ROT_MOVE, 1 | 128, 2, 4>());
#endif
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(starlin_variant<part_1, 2, 64, 16, 1, ROT_MOVE, 1, 4, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_UNROLL, 1, 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_MOVE, 0, 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 4, ROT_MOVE, 1, 2, 4>());    // tile 64x64
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_UNROLL, 9, 2, 4>());  // planes two ahead
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_MOVE, 9, 2, 4>());
#endif

}
}  // namespace ykh
