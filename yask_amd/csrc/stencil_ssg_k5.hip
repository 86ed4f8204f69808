// stencil_ssg_k5.hip -- instruction diet of the shipping ssg shape (tile 128x16, 16-byte lanes, nt, halo rings), stage 1:
// packed subtractions (_ps), reciprocal-based divisions (_fd), queue renaming in trips (_t2 / _t4); see ykh_march.hpp.
#include "gen/ssg_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"
namespace ykh {
using namespace ykh_gen_ssg;
void ssg_variants_k5(PartImpl& p) {
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(march_variant<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4>());             // 144 v_sub_f32 -> 72 v_pk_fma_f32
    p.variants.push_back(march_variant<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 8>());             // 3 divisions per point: ~130 -> ~18 instructions
    p.variants.push_back(march_variant<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8>());
#endif
    p.variants.push_back(march_variant_planned<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 16>());        // exact arithmetic, trips of 2
    p.variants.push_back(march_variant_planned<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 16>());
#ifdef YKH_PROFILING      // round 5 A/B: + XCD lock-step every 16 / 64 planes (ykh_device.hpp xcd_lockstep)
    p.variants.push_back(march_variant<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 16 | (5 << 9)>());
    p.variants.push_back(march_variant<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 16 | (7 << 9)>());
#endif
#ifdef YKH_PROFILING      // write-through output stores: measured 1-12 % slower, profiles/r4_wt
    p.variants.push_back(march_variant<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 16 | 256>());
#endif
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(march_variant<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 32>());    // trips of 4: 256 VGPRs, no spill
    p.variants.push_back(march_variant<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 16 | 128>());   // + late refill of the centre-only operands
    p.variants.push_back(march_variant<part_1, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 32 | 128>());
#endif
}
}  // namespace ykh
