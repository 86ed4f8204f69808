// stencil_ssg_k2.hip -- marching-kernel tile shapes for ssg part_2.
#include "gen/ssg_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"
namespace ykh {
using namespace ykh_gen_ssg;
void ssg_variants_k2(PartImpl& p) {
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(march_variant<part_2, 4, 32, 8, 2>());
#endif
    p.variants.push_back(march_variant_planned<part_2, 2, 64, 8, 2>());      // (+ twin for planned launches: the multi-rank tests' shape)
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(march_variant<part_2, 2, 32, 16, 2>());
    p.variants.push_back(march_variant<part_2, 2, 64, 4, 2>());
    p.variants.push_back(march_variant<part_2, 2, 64, 8, 2, 1, true>());   // strict program order (pin() after every temporary)
    p.variants.push_back(march_variant<part_2, 2, 32, 16, 2, 1, false, 2>());   // two planes ahead
    p.variants.push_back(march_variant<part_2, 2, 64, 8, 2, 1, false, 2>());
#endif
}
}  // namespace ykh
