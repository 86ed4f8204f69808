// stencil_ssg.hip -- kernel registry of solution 'ssg' (staggered-grid 3-D elastic wave propagation,
// 2 stages: 3 velocity updates, then 6 stress updates; DSL: src/stencils/SSGElasticStencil.cpp and
// src/stencils/ElasticStencil/ElasticStencil.hpp of the reference).
#include "gen/ssg_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {
void ssg_variants_k1(PartImpl&);   // marching kernels, stage 1
void ssg_variants_k2(PartImpl&);   // marching kernels, stage 2
void ssg_variants_k3(PartImpl&);   // more shapes, stage 1
void ssg_variants_k4(PartImpl&);   // more shapes, stage 2
void ssg_variants_k5(PartImpl&);   // instruction diet, stage 1
void ssg_variants_k6(PartImpl&);   // instruction diet, stage 2

const SolnImpl& ykh_solution_impl() {
    using namespace ykh_gen_ssg;
    static const SolnImpl impl = [] {
        SolnImpl s;
        s.meta = &soln;
        {
            PartImpl p;
            p.meta = &parts[0];
            p.variants.push_back(naive_variant<part_1>());
            p.variants.push_back(vecpt_variant<part_1, 4, 64, 4, 1>());
            p.variants.push_back(vecpt_variant<part_1, 4, 64, 4, 2>());
            p.variants.push_back(vecpt_variant<part_1, 4, 32, 8, 2>());
            p.variants.push_back(vecpt_variant<part_1, 4, 64, 4, 4>());
            p.variants.push_back(vecpt_variant<part_1, 2, 64, 4, 2>());
            ssg_variants_k1(p);
            ssg_variants_k3(p);
            ssg_variants_k5(p);
            // 512^3, same box (gpurun_out/r03f): nt_hr 1.266 ms -> + packed subtractions 1.231 -> + reciprocal divisions 1.188 ->
            // + trips of 2 1.178 (trips of 4: 1.177); exact arithmetic only (_ps_t2): 1.197
            p.set_default("march_v4_z128_y16_nt_hr_ps_fd_t2_w2");
            p.set_exact_div("march_v4_z128_y16_nt_hr_ps_t2_w2");
            s.parts.push_back(p);
        }
        {
            PartImpl p;
            p.meta = &parts[1];
            p.variants.push_back(naive_variant<part_2>());
            p.variants.push_back(vecpt_variant<part_2, 4, 64, 4, 1>());
            p.variants.push_back(vecpt_variant<part_2, 4, 64, 4, 2>());
            p.variants.push_back(vecpt_variant<part_2, 4, 32, 8, 2>());
            p.variants.push_back(vecpt_variant<part_2, 4, 64, 4, 4>());
            p.variants.push_back(vecpt_variant<part_2, 2, 64, 4, 2>());
            ssg_variants_k2(p);
            ssg_variants_k4(p);
            ssg_variants_k6(p);
            // 512^3: nt_hr 1.671 ms -> reciprocal divisions 1.602; packed subtractions / trips do not fit 256 VGPRs here
            // (with the late refill of the nine centre-only operands, _lo, everything fits 256 VGPRs -- and changes nothing: _fd
            //  alone already moves 6.27 TB/s; the exact shape gains 1.9 %: 1.674 -> 1.643 ms, gpurun_out/r03m)
            p.set_default("march_v4_z128_y16_nt_hr_fd_w2");
            p.set_exact_div("march_v4_z128_y16_nt_hr_ps_t2_lo_w2");
            // planned launches of a decomposed rank (descriptor-reading twins, ~5 more SGPRs): the default's twin spills (it sits at
            // 256 VGPRs); the late-refill form of the same arithmetic has room (232 VGPRs)
            p.set_planned("march_v4_z128_y16_nt_hr_fd_lo_w2");
            s.parts.push_back(p);
        }
        return s;
    }();
    return impl;
}
}  // namespace ykh
