// stencil_3axis.hip -- kernel registry of solution '3axis' (AxisStencil, radius 4, fp64: the
// "heat3d"-like double-precision bandwidth validation case; DSL: src/stencils/SimpleStencils.cpp:39-115
// of the reference).  Links with the generic runtime into libyask_kernel.3axis.cdna4_hip.so.
#include "gen/3axis_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {
void s3axis_variants_k1(PartImpl&);
void s3axis_variants_k2(PartImpl&);
void s3axis_variants_k3(PartImpl&);   // two steps per pass
void s3axis_variants_k4(PartImpl&);   // queue renaming inside trips

const SolnImpl& ykh_solution_impl() {
    using namespace ykh_gen_3axis;
    static const SolnImpl impl = [] {
        SolnImpl s;
        s.meta = &soln;
        PartImpl p;
        p.meta = &parts[0];
        p.variants.push_back(naive_variant<part_1>());
        s3axis_variants_k1(p);
        s3axis_variants_k2(p);
        s3axis_variants_k3(p);
        s3axis_variants_k4(p);
        p.set_default("starlin_v2_z64_y32_r2_u_nt_tl_w2_c4");      // cheap tail planes: 512^3 0.395 vs 0.412 ms (gpurun_out/r3j), bit-identical
        p.set_large_grid("starlin_v2_z128_y32_r4_m_nt_ls64_w2_c4");     // (round 4: the same shape without the lock-step)
        s.parts.push_back(p);
        return s;
    }();
    return impl;
}
}  // namespace ykh
