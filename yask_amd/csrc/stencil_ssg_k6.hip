// stencil_ssg_k6.hip -- instruction diet of the shipping ssg shape, stage 2 (256 VGPRs already: the shapes that fit).
#include "gen/ssg_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"
namespace ykh {
using namespace ykh_gen_ssg;
void ssg_variants_k6(PartImpl& p) {
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 4>());             // (spills 2 registers)
#endif
#ifdef YKH_PROFILING      // round 5 A/B: + XCD lock-step every 16 / 64 planes
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 8 | (5 << 9)>());
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 8 | (7 << 9)>());
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 8 | 128 | (5 << 9)>());      // (on the _lo shape: the plain one spills 2 registers with the lock-step)
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 8 | 128 | (7 << 9)>());
#endif
    p.variants.push_back(march_variant_planned<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 8>());             // 5 divisions per point: ~200 -> ~30 instructions
#ifdef YKH_PROFILING      // write-through output stores: measured 1-12 % slower, profiles/r4_wt
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 8 | 256>());
#endif
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8>());         // (spills 2 registers)
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 1 | 4 | 8>());         // without the halo rings: fits
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 8 | 16>());        // trips of 2 (spills 6)
    // late refill of the nine centre-only operands (_lo: each held once instead of twice, 256 -> 232 VGPRs): everything fits
#endif
    p.variants.push_back(march_variant_planned<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 8 | 128>());
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 128>());
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 16 | 128>());
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 8 | 32 | 128>());
#endif
    p.variants.push_back(march_variant_planned<part_2, 4, 32, 16, 2, 1, false, 1, 3 | 4 | 16 | 128>());      // exact arithmetic
}
}  // namespace ykh
