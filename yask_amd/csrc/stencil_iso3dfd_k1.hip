// stencil_iso3dfd_k1.hip -- kernel instantiations for solution 'iso3dfd', group 1 (split over several
// translation units so that hipcc compiles the tile shapes in parallel).
#include "gen/iso3dfd_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {
using namespace ykh_gen_iso3dfd;
void iso3dfd_variants_k1(PartImpl& p) {
    p.variants.push_back(star_variant<part_1, 32, 16, 1, ROT_UNROLL>());
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(star_variant<part_1, 32, 16, 1, ROT_MOVE>());
#endif
    p.variants.push_back(star_variant<part_1, 32, 8, 1, ROT_UNROLL>());
}
}  // namespace ykh
