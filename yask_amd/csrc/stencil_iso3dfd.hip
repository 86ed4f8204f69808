// stencil_iso3dfd.hip -- kernel registry of solution 'iso3dfd' (16th-order isotropic 3-D
// finite-difference wave propagation; DSL: src/stencils/Iso3dfdStencil.cpp of the reference).
// The tile-shape instantiations live in stencil_iso3dfd_k*.hip; everything links with the generic
// runtime into libyask_kernel.iso3dfd.cdna4_hip.so.
#include "gen/iso3dfd_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {
void iso3dfd_variants_k1(PartImpl&);   // star25d (gather) shapes
void iso3dfd_variants_k2(PartImpl&);   // starlin (gather past / scatter future) shapes
void iso3dfd_variants_k3(PartImpl&);
void iso3dfd_variants_k4(PartImpl&);
void iso3dfd_variants_k5(PartImpl&);   // profiling ablations ("abl*": wrong results on purpose)
void iso3dfd_variants_k6(PartImpl&);   // 1024-thread shapes

const SolnImpl& ykh_solution_impl() {
    using namespace ykh_gen_iso3dfd;
    static const SolnImpl impl = [] {
        SolnImpl s;
        s.meta = &soln;
        PartImpl p;
        p.meta = &parts[0];
        p.variants.push_back(naive_variant<part_1>());
        iso3dfd_variants_k1(p);
        iso3dfd_variants_k2(p);
        iso3dfd_variants_k3(p);
        iso3dfd_variants_k4(p);
        iso3dfd_variants_k5(p);
        iso3dfd_variants_k6(p);
        // same box A/B (gpurun_out/r03c): _t2 358.3 vs 349.0 Gpoints/s for _m; round 3 (gpurun_out/r3j, bit-identical shapes):
        // + cheap tail planes 1024^3 2.911 vs 2.979 ms, 512^3 0.418 vs 0.428 ms, 1024x1024x512 1.535 vs 1.536 ms
        p.set_default("starlin_v4_z128_y32_r2_t2_nt_pd2_tl_w2_c2");
        // planned launches of a decomposed rank: the twin of the 1-trip renaming (242 VGPRs; the default's twin takes all 256).  Two
        // waves of 256 VGPRs fill a SIMD's register file, and a CU that holds any other wave -- a transport's flag waiter, a
        // set-flag kernel -- cannot take a marching workgroup: the launch loses that CU and runs one more round for the block left
        // over (tools/microbench/waiter_cost.hip: 256 blocks 0.20 -> 0.40 ms beside ONE polling wave, no loss at 240 VGPRs).
        // Same bits as the default (every r2 shape is, tests), 0-3 % slower on whole blocks.
        p.set_planned("starlin_v4_z128_y32_r2_t_nt_pd2_tl_w2_c2");
        s.parts.push_back(p);
        return s;
    }();
    return impl;
}

}  // namespace ykh
