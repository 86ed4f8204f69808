// stencil_iso3dfd_k3.hip -- kernel instantiations for solution 'iso3dfd', group 3 (split over several
// translation units so that hipcc compiles the tile shapes in parallel).
#include "gen/iso3dfd_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {
using namespace ykh_gen_iso3dfd;
void iso3dfd_variants_k3(PartImpl& p) {
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(starlin_variant<part_1, 2, 64, 16, 1, ROT_MOVE, 1, 4, 8, 0>());
    p.variants.push_back(starlin_variant<part_1, 2, 64, 8, 1, ROT_MOVE, 1, 4, 8, 0>());
    p.variants.push_back(starlin_variant<part_1, 4, 64, 4, 1, ROT_MOVE, 1, 3, 4, 0>());
#endif
}
}  // namespace ykh
