// stencil_iso3dfd_k5.hip -- kernel instantiations for solution 'iso3dfd', group 5 (split over several
// translation units so that hipcc compiles the tile shapes in parallel).
#include "gen/iso3dfd_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {
using namespace ykh_gen_iso3dfd;
// Profiling ablations (no halo loads / no operand loads / no stores): they compute WRONG results on purpose, so the shipped
// library does not contain them -- `make YKH_PROFILING=1` (-DYKH_PROFILING) builds a library that does.
void iso3dfd_variants_k5(PartImpl& p) {
#ifdef YKH_PROFILING
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 1, 2, 4, 1>());
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 1, 2, 4, 4>());
    p.variants.push_back(starlin_variant<part_1, 4, 32, 16, 2, ROT_MOVE, 1, 2, 4, 7>());
#else
    (void)p;
#endif
}
}  // namespace ykh
