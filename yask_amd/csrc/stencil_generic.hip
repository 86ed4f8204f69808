// stencil_generic.hip -- kernel registry for ANY solution the `cdna4_hip` compiler target can render:
// every part gets the always-legal naive kernel and, for 3-D solutions, the vector-per-thread kernel.
// Compiled once per stencil with -DYKH_GEN_HEADER="gen/<name>_cdna4_hip.hpp" -DYKH_GEN_NS=ykh_gen_<name>
// (Makefile: GENERIC_STENCILS).  Hand-tuned registries (stencil_iso3dfd.hip, stencil_3axis.hip,
// stencil_ssg.hip) add the marching kernels for the hot-path stencils.
#include YKH_GEN_HEADER
#include "ykh_stencil_tu.hpp"

namespace ykh {
const SolnImpl& ykh_solution_impl() {
    using namespace YKH_GEN_NS;
    static const SolnImpl impl = [] {
        SolnImpl s;
        s.meta = &soln;
        int ndd = 0;
        for (int i = 0; i < soln.ndims; i++) ndd += (dims[i].type == DIM_DOMAIN);
        int pi = 0;
#define YKH_ADD_PART(PART)                                                                      \
        {                                                                                           \
            PartImpl p;                                                                             \
            p.meta = &parts[pi++];                                                                  \
            p.variants.push_back(naive_variant<PART>());                                            \
            if (ndd == 3 && !PART::has_domain_cond) {                                               \
                p.variants.push_back(vecpt_variant<PART, 16 / (int)sizeof(real_t), 64, 4, 1>());    \
                p.default_variant = 1;                                                              \
            }                                                                                       \
            s.parts.push_back(p);                                                                   \
        }
        YKH_FOR_EACH_PART(YKH_ADD_PART)
#undef YKH_ADD_PART
        return s;
    }();
    return impl;
}
}  // namespace ykh
