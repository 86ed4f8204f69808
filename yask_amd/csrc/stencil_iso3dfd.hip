// stencil_iso3dfd.hip -- kernel instantiations for solution 'iso3dfd' (16th-order isotropic 3-D
// finite-difference wave propagation; DSL: src/stencils/Iso3dfdStencil.cpp of the reference).
// Links with the generic runtime into libyask_kernel.iso3dfd.cdna4_hip.so.
#include "gen/iso3dfd_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {

const SolnImpl& ykh_solution_impl() {
    using namespace ykh_gen_iso3dfd;
    static const SolnImpl impl = [] {
        SolnImpl s;
        s.meta = &soln;
        PartImpl p;
        p.meta = &parts[0];
        p.variants.push_back(naive_variant<part_1>());
        // <lanes along z (x4 floats), thread rows, rows per thread, queue rotation>
        p.variants.push_back(star_variant<part_1, 32, 8, 1, ROT_MOVE>());
        p.variants.push_back(star_variant<part_1, 32, 8, 1, ROT_UNROLL>());
        p.variants.push_back(star_variant<part_1, 64, 4, 1, ROT_MOVE>());
        p.variants.push_back(star_variant<part_1, 64, 4, 1, ROT_UNROLL>());
        p.variants.push_back(star_variant<part_1, 32, 16, 1, ROT_MOVE>());
        p.variants.push_back(star_variant<part_1, 32, 16, 1, ROT_UNROLL>());
        p.variants.push_back(star_variant<part_1, 64, 8, 1, ROT_MOVE>());
        p.variants.push_back(star_variant<part_1, 64, 8, 1, ROT_UNROLL>());
        p.variants.push_back(star_variant<part_1, 32, 8, 2, ROT_MOVE>());
        p.variants.push_back(star_variant<part_1, 32, 8, 2, ROT_UNROLL>());
        p.variants.push_back(star_variant<part_1, 16, 16, 1, ROT_MOVE>());
        p.variants.push_back(star_variant<part_1, 32, 16, 2, ROT_MOVE>());
        p.variants.push_back(star_variant<part_1, 32, 16, 2, ROT_UNROLL>());
        p.variants.push_back(star_variant<part_1, 64, 8, 2, ROT_MOVE>());
        p.variants.push_back(star_variant<part_1, 64, 8, 2, ROT_UNROLL>());
        p.variants.push_back(star_variant<part_1, 64, 4, 2, ROT_UNROLL>());
        p.variants.push_back(star_variant<part_1, 16, 32, 1, ROT_UNROLL>());
        p.variants.push_back(star_variant<part_1, 32, 16, 1, ROT_UNROLL, 1>());
        p.variants.push_back(star_variant<part_1, 32, 16, 1, ROT_UNROLL, 2>());
        p.variants.push_back(star_variant<part_1, 32, 16, 1, ROT_UNROLL, 4>());
        p.variants.push_back(star_variant<part_1, 32, 16, 1, ROT_UNROLL, 7>());
        p.default_variant = 6;   // star25d_z128_y16_r1_u
        s.parts.push_back(p);
        return s;
    }();
    return impl;
}

}  // namespace ykh
