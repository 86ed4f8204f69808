// stencil_ssg.hip -- kernel registry of solution 'ssg' (staggered-grid 3-D elastic wave propagation,
// 2 stages: 3 velocity updates, then 6 stress updates; DSL: src/stencils/SSGElasticStencil.cpp and
// src/stencils/ElasticStencil/ElasticStencil.hpp of the reference).
#include "gen/ssg_cdna4_hip.hpp"
#include "ykh_stencil_tu.hpp"

namespace ykh {
const SolnImpl& ykh_solution_impl() {
    using namespace ykh_gen_ssg;
    static const SolnImpl impl = [] {
        SolnImpl s;
        s.meta = &soln;
        {
            PartImpl p;
            p.meta = &parts[0];
            p.variants.push_back(naive_variant<part_1>());
            p.default_variant = 0;
            s.parts.push_back(p);
        }
        {
            PartImpl p;
            p.meta = &parts[1];
            p.variants.push_back(naive_variant<part_2>());
            p.default_variant = 0;
            s.parts.push_back(p);
        }
        return s;
    }();
    return impl;
}
}  // namespace ykh
