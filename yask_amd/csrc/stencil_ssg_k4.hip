// stencil_ssg_k4.hip -- more marching-kernel shapes for ssg part_2 (rows per thread, 16-byte lanes, deeper prefetch).
#include "gen/ssg_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"
namespace ykh {
using namespace ykh_gen_ssg;
void ssg_variants_k4(PartImpl& p) {
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(march_variant<part_2, 2, 64, 8, 2, 2, false, 1>());   // two rows per thread: tile 128x16
    p.variants.push_back(march_variant<part_2, 2, 64, 8, 2, 2, false, 2>());
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1>());  // 16-byte lanes, tile 128x16
    p.variants.push_back(march_variant<part_2, 4, 64, 8, 2, 1, false, 1>());   // 16-byte lanes, tile 256x8
    p.variants.push_back(march_variant<part_2, 2, 64, 8, 2, 1, false, 3>());   // three planes ahead
    p.variants.push_back(march_variant<part_2, 2, 64, 8, 4, 1, false, 1>());   // <= 128 VGPRs: two workgroups per CU
    p.variants.push_back(march_variant<part_2, 4, 64, 8, 2, 1, false, 1, 1>());   // + non-temporal one-touch streams
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 1>());
    p.variants.push_back(march_variant<part_2, 2, 64, 8, 2, 1, false, 2, 1>());
    p.variants.push_back(march_variant<part_2, 4, 16, 32, 2, 1, false, 1, 1>());   // tile 64x32: less y halo, more z halo
    p.variants.push_back(march_variant<part_2, 2, 64, 8, 2, 2, false, 1, 1>());    // 8-byte lanes, two rows per thread, nt
    p.variants.push_back(march_variant<part_2, 2, 32, 16, 2, 2, false, 1, 1>());   // tile 64x32 with 8-byte lanes
#endif
    p.variants.push_back(march_variant<part_2, 4, 32, 16, 2, 1, false, 1, 3>());   // 128x16, nt + halo rings
#ifdef YKH_PROFILING      // sweep shapes: measured, documented, never selected -- built with `make YKH_PROFILING=1` only
    p.variants.push_back(march_variant<part_2, 2, 64, 8, 2, 1, false, 1, 3>());    // 8-byte lanes, nt + halo rings
#endif
}
}  // namespace ykh
