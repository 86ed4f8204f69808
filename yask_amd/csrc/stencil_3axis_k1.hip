// stencil_3axis_k1.hip -- tile shapes for '3axis' (fp64: a 16-byte z-vector is 2 doubles).
#include "gen/3axis_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"
namespace ykh {
using namespace ykh_gen_3axis;
void s3axis_variants_k1(PartImpl& p) {
    p.variants.push_back(star_variant<part_1, 32, 16, 1, ROT_UNROLL>());
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 2, ROT_MOVE, 1, 2, 4>());
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 2, ROT_MOVE, 1, 2, 4>());
    p.variants.push_back(starlin_variant<part_1, 2, 32, 16, 1, ROT_MOVE, 1, 4, 4>());
#endif
}
}  // namespace ykh
