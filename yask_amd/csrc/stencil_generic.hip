// stencil_generic.hip -- kernel registry for ANY solution the `cdna4_hip` compiler target can render.
// Every part gets the always-legal point kernel; 3-D parts without sub-domain conditions also get, when
// eligible (decided at compile time from the generated part): the vector-per-thread kernel, the plane-ring kernel
// (ykh_box.hpp: parts with many mixed-offset reads), the generic
// marching kernel (all offset-read and written groups are full-dim vars, slabs fit the LDS, <= 48 groups) and
// the linear-star kernel (the compiler found the linear star form).  Default = the most specialised one.
// Every marching and plane-ring shape is registered WITH its descriptor-reading twin (march_variant_planned / box_variant_planned): a
// decomposed rank runs a one-part stage on such a shape as a planned launch or as pipelined half-exchanges instead of exterior slabs +
// interior (profiles/r6_generic_twins: awp blocks 1.23 -> 1.04 x the undivided sweep at 256^3); a twin that spills is never used.
// Compiled once per stencil with -DYKH_GEN_HEADER="gen/<name>_cdna4_hip.hpp" -DYKH_GEN_NS=ykh_gen_<name>
// (Makefile: GENERIC_STENCILS).  The hand-tuned registries (stencil_iso3dfd.hip, stencil_3axis.hip,
// stencil_ssg.hip) list more tile shapes for the hot-path stencils.
#include YKH_GEN_HEADER
#include "ykh_stencil_tu.hpp"
#include <deque>

namespace ykh {

template <class P>
constexpr bool starlin_eligible() {
    if constexpr (!P::has_lin) return false;
    else {
        if (!P::group_full[P::lin_group]) return false;
        for (int i = 0; i < P::n_writes; i++)
            if (!P::group_full[P::writes[i]]) return false;
        return P::n_groups <= 16;
    }
}

// The plane-ring shapes of a part (ykh_box.hpp) -- one list for the two translation units that register them: this one, and (for
// solutions whose generated header carries the build hint "max-mixed-reads" > MAX_MIXED, csrc/Makefile) a second one compiled WITHOUT
// packed fp32 instructions, where Q = NoPk<P> gives every kernel a symbol of its own and the shapes get the suffix "_np".  Whether a
// chain of fp32 additions runs faster as v_pk_add_f32 (~10 VALU cycles per wave) or as two v_add_f32 (4 each) depends on the part
// (cube, 3axis_with_diags, tti: plain adds; 3plane: packed; profiles/r5_box): prepare_solution()'s timing decides per part -- there is
// no list of solution names in the build any more.
template <class P>
struct NoPk : P {};
inline const char* keep_name(const std::string& n) {
    static std::deque<std::string> names;          // (deque: pointers stay valid)
    names.push_back(n);
    return names.back().c_str();
}
template <class Q>
void add_box_family(PartImpl& p, const char* suffix, bool set_default) {
    typedef typename Q::real_t T;
    constexpr int VZ = 16 / (int)sizeof(T);
    const size_t first = p.variants.size();
    // box / plane neighbourhoods (more mixed-offset reads than the marching kernel prefetches): planes in an LDS ring
    // (ykh_box.hpp), for as many groups as the budget holds -- registered where the rings serve at least half of the reads.
    // Tile 128 x 16 points (fp32; 16-byte lanes): 512 threads with one row each, or 256 threads with two rows each evaluated as one
    // wide vector (shared LDS rows are loaded once; needs ~230 VGPRs); _p2 = planes requested two iterations ahead; _w1 = one
    // wave per SIMD with the whole register file, for parts like tti (340 VGPRs on any kernel).  prepare_solution() times them.
    if constexpr (box_eligible<Q>() && box_wanted<Q>()) {
        constexpr int TZL = 32;          // (z tile = 32 lanes of 16 bytes)
        if constexpr (2 * BoxCfg<Q, VZ, TZL, 16, 1>::ring_reads() >= Q::n_reads) {
            p.variants.push_back(box_variant_planned<Q, VZ, TZL, 16, 1, 2, 1>());
            if (set_default) p.default_variant = (int)p.variants.size() - 1;
            p.variants.push_back(box_variant_planned<Q, VZ, TZL, 16, 1, 2, 1 | 4>());
            p.variants.push_back(box_variant_planned<Q, VZ, TZL, 8, 2, 2, 1>());
            p.variants.push_back(box_variant_planned<Q, VZ, TZL, 8, 2, 2, 1 | 4>());
        }
        if constexpr (2 * BoxCfg<Q, VZ, TZL, 8, 1, 80>::ring_reads() >= Q::n_reads && Q::n_reads > 130)
            p.variants.push_back(box_variant_planned<Q, VZ, TZL, 8, 1, 1, 1 | 4, 80>());
        // where the 128 x 16 tile leaves groups without a ring, the half-height tile holds more of them in the same LDS
        if constexpr (BoxCfg<Q, VZ, TZL, 8, 1>::ring_reads() > BoxCfg<Q, VZ, TZL, 16, 1>::ring_reads() &&
                      2 * BoxCfg<Q, VZ, TZL, 8, 1>::ring_reads() >= Q::n_reads) {
            p.variants.push_back(box_variant_planned<Q, VZ, TZL, 8, 1, 2, 1 | 4>());
            if constexpr (Q::n_reads > 130) p.variants.push_back(box_variant_planned<Q, VZ, TZL, 8, 1, 1, 1 | 4>());
            // ... and with 8-byte lanes (two points per thread) such a part fits 256 VGPRs without spilling: 512 threads, two waves
            // per SIMD where the 16-byte-lane shape above runs one (tti: 251 VGPRs, no scratch; 256 + 68 B without packed fp32)
            if constexpr (Q::n_reads > 130 && VZ == 4) p.variants.push_back(box_variant_planned<Q, 2, 64, 8, 1, 2, 1 | 4>());
        }
        // parts that read small tables over a subset of the domain dims at offsets (kind 4; test_partial_3d: 59 of 79 reads): every such
        // read is live in registers from its load to its use -- with 16-byte lanes 950+ bytes of scratch per thread; 8-byte lanes halve it
        if constexpr (box_has_tables<Q>() && VZ == 4) {
            p.variants.push_back(box_variant_planned<Q, 2, 64, 8, 1, 2, 1 | 4>());
            // ... or one wave per SIMD with the whole register file (16-byte lanes, 128 x 8 tile: no scratch)
            p.variants.push_back(box_variant_planned<Q, VZ, 32, 8, 1, 1, 1 | 4>());
        }
        // (the no-packed unit also twins the vector point kernel: tti's ran 8.29 -> 5.39 ms without packed adds)
        if (suffix[0]) p.variants.push_back(vecpt_variant<Q, VZ, 64, 4, 1>());
    }
    if (suffix[0])
        for (size_t i = first; i < p.variants.size(); i++) p.variants[i].name = keep_name(std::string(p.variants[i].name) + suffix);
}

#ifdef YKH_NOPK_TU
// ---- the second translation unit of a solution with box / plane neighbourhoods: compiled with -target-feature -packed-fp32-ops
template <class P>
void add_np_part(PartImpl& p, int ndd) {
    if (ndd == 3 || ndd == 4) add_box_family<NoPk<P>>(p, "_np", false);
}
void ykh_add_np_variants(SolnImpl& s, int ndd) {
    using namespace YKH_GEN_NS;
    int pi = 0;
#define YKH_ADD_PART(PART) add_np_part<PART>(s.parts[pi++], ndd);
    YKH_FOR_EACH_PART(YKH_ADD_PART)
#undef YKH_ADD_PART
}
#else
void ykh_add_np_variants(SolnImpl& s, int ndd);       // (defined in the no-packed unit when the build has one: YKH_HAS_NOPK_TU)

template <class P>
void add_part(SolnImpl& s, const PartMeta* meta, int ndd) {
    typedef typename P::real_t T;
    constexpr int VZ = 16 / (int)sizeof(T);
    PartImpl p;
    p.meta = meta;
    p.variants.push_back(naive_variant<P>());
    if constexpr (P::has_domain_cond) { p.cond_bb = &launch_cond_bb<P>; p.cond_profile = &launch_cond_profile<P>; }
    // Sub-domain parts get the same kernels: they do not evaluate the condition, and prepare_solution() selects
    // them only when the condition holds at every point of its bounding box (a "solid" box, e.g. awp's
    // below-the-surface updates); otherwise the point kernel runs.
    // (four domain dims -- test_4d: the outermost one is a loop of launches with shifted base pointers, Solution::launch_part; each launch
    //  sweeps the inner three with any family below, the access groups telling the outer offsets apart.  Until the end of round 6 such
    //  solutions got the point kernels only.)
    if (ndd == 3 || ndd == 4) {
        {
            p.variants.push_back(vecpt_variant<P, VZ, 64, 4, 1>());
            p.default_variant = (int)p.variants.size() - 1;
            // parts with many reads and no other vector family (test_partial_3d: 79 operands, 346 VGPRs with 16-byte lanes = one wave per
            // SIMD): the same kernel with 8-byte lanes needs half the registers per operand
            if constexpr (VZ == 4 && P::n_reads > 64 && P::n_writes < 12 && !march_eligible<P>())
                p.variants.push_back(vecpt_variant<P, 2, 64, 4, 1>());
            if constexpr (march_eligible<P>() && P::n_groups <= 48) {
                // 8-byte lanes keep the per-thread queue state small (ykh_march.hpp); tile 128 x 8
                if constexpr (MarchCfg<P, 2, 64, 8>::lds_bytes <= 160 * 1024) {
                    p.variants.push_back(march_variant_planned<P, 2, 64, 8, 2>());
                    p.default_variant = (int)p.variants.size() - 1;
                }
                // 16-byte lanes, tile 128 x 16, one-touch streams non-temporal: fastest where the state still fits
                // 256 VGPRs (ssg: +12 %); prepare_solution() steps back to the shape above when this one spilled
                if constexpr (VZ > 2 && MarchCfg<P, VZ, 32, 16>::lds_bytes <= 160 * 1024) {
                    p.variants.push_back(march_variant_planned<P, VZ, 32, 16, 2, 1, false, 1, 1>());
                    p.default_variant = (int)p.variants.size() - 1;
                    // + packed subtractions (exact: a - b as fma(b, -1, a); ssg stage 1: -2.8 %) -- another candidate for
                    // the timing pass of prepare_solution(), which also drops it where the different schedule spills
                    p.variants.push_back(march_variant_planned<P, VZ, 32, 16, 2, 1, false, 1, 1 | 4>());
                }
                // + halo rings where a group has both a queue reaching ahead and a slab (ykh_march.hpp, HR)
                if constexpr (VZ > 2 && MarchCfg<P, VZ, 32, 16, 1, true>::RING_TOT > 0 &&
                              MarchCfg<P, VZ, 32, 16, 1, true>::lds_bytes <= 160 * 1024) {
                    p.variants.push_back(march_variant_planned<P, VZ, 32, 16, 2, 1, false, 1, 3>());
                    p.default_variant = (int)p.variants.size() - 1;
                    p.variants.push_back(march_variant_planned<P, VZ, 32, 16, 2, 1, false, 1, 3 | 4>());
                }
                // + late refill of the centre-only operands (_lo: each is held once instead of twice, ykh_march.hpp FL & 128) for parts
                // that have several: awp's velocity part 256 VGPRs + 20 B of scratch -> 231, 1.61 -> 1.34 ms at 512^3; awp_elastic's
                // stress part (8-byte lanes only without it) 2.62 -> 1.83; ssg2 -5 / -2 %; awp's stress part with 8-byte lanes 212 ->
                // 166 VGPRs, 3.92 -> 3.77 (job r6zd; planes two ahead, two workgroups per CU, stores inside eval(): no gain, r6zc / r6zd)
                if constexpr (march_once_count<P>() >= 3) {
                    if constexpr (MarchCfg<P, 2, 64, 8>::lds_bytes <= 160 * 1024 && VZ > 2)
                        p.variants.push_back(march_variant_planned<P, 2, 64, 8, 2, 1, false, 1, 1 | 128>());
                    if constexpr (VZ > 2 && MarchCfg<P, VZ, 32, 16>::lds_bytes <= 160 * 1024) {
                        p.variants.push_back(march_variant_planned<P, VZ, 32, 16, 2, 1, false, 1, 1 | 128>());
                        p.variants.push_back(march_variant_planned<P, VZ, 32, 16, 2, 1, false, 1, 1 | 4 | 128>());
                    }
                    if constexpr (VZ > 2 && MarchCfg<P, VZ, 32, 16, 1, true>::RING_TOT > 0 &&
                                  MarchCfg<P, VZ, 32, 16, 1, true>::lds_bytes <= 160 * 1024) {
                        p.variants.push_back(march_variant_planned<P, VZ, 32, 16, 2, 1, false, 1, 3 | 128>());
                        p.variants.push_back(march_variant_planned<P, VZ, 32, 16, 2, 1, false, 1, 3 | 4 | 128>());
                    }
                }
            }
            add_box_family<P>(p, "", true);       // plane-ring shapes for box / plane neighbourhoods (above)
            // big bundles (fsg: 12 and 24 equations, 296 / 435 reads): the part as K clusters of equations, one launch each
            // (ykh_subpart.hpp) -- on the point kernel (a cluster needs a third of the registers: three waves per SIMD instead
            // of one) and, where every cluster's slabs fit the LDS, on the marching kernel.
            if constexpr (P::n_writes >= 12) {
                if constexpr (clusters_legal<P, 4>()) {
                    p.variants.push_back(vecpt_cluster_variant<P, 4, VZ, 64, 4, 1>());
                    if constexpr (march_clusters_fit<P, 4, 2, 64, 8, 0>()) p.variants.push_back(march_cluster_variant<P, 4, 2, 64, 8, 2, 0>());
                }
                if constexpr (clusters_legal<P, 2>()) p.variants.push_back(vecpt_cluster_variant<P, 2, VZ, 64, 4, 1>());
                // (8-byte lanes: 6 % faster on fsg's stress clusters; K = 8, two x planes per thread, 128 x 8 / 64 x 16 tiles: 1-55 % slower, jobs r5p / r5v)
                if constexpr (P::n_writes >= 24 && VZ == 4 && clusters_legal<P, 4>()) p.variants.push_back(vecpt_cluster_variant<P, 4, 2, 64, 4, 1>());
            }
            // sub-domain parts run box by box (Solution::find_part_boxes), and the boxes of a shell include slabs that are thin in z:
            // a point-kernel tile of 32 z x 32 y keeps 5 of its 8 z lanes busy on a 20-point slab where the 256 x 4 tile keeps 5 of 64
            if constexpr (P::has_domain_cond) {
                p.variants.push_back(vecpt_variant<P, VZ, 8, 32, 1>());
                if constexpr (P::n_writes >= 12) {
                    if constexpr (clusters_legal<P, 4>()) p.variants.push_back(vecpt_cluster_variant<P, 4, VZ, 8, 32, 1>());
                    if constexpr (clusters_legal<P, 2>()) p.variants.push_back(vecpt_cluster_variant<P, 2, VZ, 8, 32, 1>());
                }
            }
            if constexpr (starlin_eligible<P>()) {
                p.variants.push_back(starlin_variant<P, VZ, 32, 16, 1, ROT_MOVE, 1, 2, 4>());
                p.default_variant = (int)p.variants.size() - 1;
                p.variants.push_back(starlin_variant<P, VZ, 32, 16, 2, ROT_MOVE, 1, 2, 4>());
                // the shape that won iso3dfd's sweeps (2-plane trips with queue renaming, star planes two ahead, cheap tail planes,
                // LDS batches of 2; stencil_iso3dfd_k4.hip): any fp32 star with an x range gets it as a candidate for the timing
                // (iso3dfd_sponge: the same star + three 1-D sponge profiles)
                // -- with two rows per thread (the headline's own shape: 256 VGPRs there, spills as soon as the part has further
                // operands: iso3dfd_sponge 0.54 ms against 0.46 on the plain shapes) and with one row per thread (172 VGPRs on the sponge)
                if constexpr (VZ == 4 && lin_range<P>().xhi > 0) {
                    p.variants.push_back(starlin_variant<P, VZ, 32, 16, 2, ROT_TRIP2, 9 | 64, 2, 2>());
                    p.variants.push_back(starlin_variant<P, VZ, 32, 16, 1, ROT_TRIP2, 9 | 64, 2, 2>());
                }
            }
        }
    } else if (ndd == 1) {
        // one domain dim: the vector point kernel on the part lifted to one row of one plane (ykh_lift2d.hpp, SHIFT = 2)
        if constexpr (lift2d_shape<P, 2>()) {
            KernelVariant kv = vecpt_variant<Lift2D<P, 2>, VZ, 256, 1, 1>();
            kv.lift1d = true;
            p.variants.push_back(kv);
            p.default_variant = (int)p.variants.size() - 1;
        }
    } else if (ndd == 2) {
        // two domain dims: the 3-D families on the part lifted to one x plane (ykh_lift2d.hpp) -- 16-byte vectors along the unit-stride
        // dim instead of the scalar point kernel's 4-byte loads (wave2d, swe2d: 15 / 65 full-domain sweeps per step); parts with many
        // mixed-offset reads (image filters) also get the plane-ring kernel, which on one plane is an LDS-tiled 2-D kernel
        if constexpr (lift2d_shape<P>()) {
            typedef Lift2D<P> L;
            auto lifted = [&](KernelVariant kv) { kv.lift2d = true; p.variants.push_back(kv); };
            lifted(vecpt_variant<L, VZ, 64, 4, 1>());
            p.default_variant = (int)p.variants.size() - 1;
            lifted(vecpt_variant<L, VZ, 16, 16, 1>());          // (64 x 16-point tile: thin boxes, short rows)
            if constexpr (box_eligible<L>() && box_wanted<L>()) {
                if constexpr (2 * BoxCfg<L, VZ, 32, 16, 1>::ring_reads() >= L::n_reads) lifted(box_variant<L, VZ, 32, 16, 1, 2, 1>());
            }
        }
    }
    s.parts.push_back(p);
}

// the generated header's var table, for the compile-time plans of ykh_fused.hpp
struct GenTraits {
    typedef YKH_GEN_NS::real_t real_t;
    static constexpr const VarMeta* vars = YKH_GEN_NS::vars;
    static constexpr const DimMeta* dims = YKH_GEN_NS::dims;
    static constexpr int n_vars = YKH_GEN_NS::soln.n_vars;
};

const SolnImpl& ykh_solution_impl() {
    using namespace YKH_GEN_NS;
    static const SolnImpl impl = [] {
        SolnImpl s;
        s.meta = &soln;
        s.select_by_timing = true;      // the per-part defaults below are a static guess (e.g. awp_abc: the point kernel wins)
        int ndd = 0;
        // (a solution with 4 domain dims -- DIM_OUTER -- only gets the point kernel: ndd counts 4)
        for (int i = 0; i < soln.ndims; i++) ndd += (dims[i].type == DIM_DOMAIN || dims[i].type == DIM_OUTER);
        int pi = 0;
#define YKH_ADD_PART(PART) add_part<PART>(s, &parts[pi++], ndd);
        YKH_FOR_EACH_PART(YKH_ADD_PART)
#undef YKH_ADD_PART
        // 2-D solutions with scratch stages: each run of scratch stages + the stage it feeds also as ONE kernel with the scratch vars in
        // the LDS (ykh_fused.hpp); prepare_solution() times a step both ways
        if (ndd == 2) {
#define YKH_ADD_FUSED(LIST, LEVEL, FIRST, LAST) add_fused_group<GenTraits, LIST, LEVEL>(s, FIRST, LAST);
            YKH_FOR_EACH_FUSE_GROUP(YKH_ADD_FUSED)
#undef YKH_ADD_FUSED
        }
#ifdef YKH_HAS_NOPK_TU
        ykh_add_np_variants(s, ndd);
#endif
        return s;
    }();
    return impl;
}
#endif   // YKH_NOPK_TU
}  // namespace ykh
