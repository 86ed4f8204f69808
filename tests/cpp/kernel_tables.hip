// Host-side probe of the compile-time tables of the generic kernel families (tests/test_kernel_tables_cpu.py): compiled per solution with
// -DYKH_GEN_HEADER / -DYKH_GEN_NS like csrc/stencil_generic.hip, host pass only; prints one JSON object per part.
#include YKH_GEN_HEADER
#include "ykh_stencil_tu.hpp"
#include <cstdio>
#include <string>
using namespace ykh;
using namespace YKH_GEN_NS;

template <class P, int K>
void clusters(const char* sep) {
    if constexpr (P::n_writes > 1) {
        printf("%s\"k%d\": {\"legal\": %d", sep, K, (int)clusters_legal<P, K>());
        if constexpr (P::n_writes >= K) {
            typedef SubPart<P, cluster_mask<P, K>(0)> S0;
            printf(", \"c0_reads\": %d, \"c0_writes\": %d, \"c0_march_ok\": %d, \"c0_lds_v2_128x8\": %zu", S0::n_reads, S0::n_writes, (int)march_eligible<S0>(),
                   MarchCfg<S0, 2, 64, 8>::lds_bytes);
        }
        printf("}");
    }
}
template <class P>
void part(const char* name, bool first) {
    constexpr int VZ = 16 / (int)sizeof(typename P::real_t);
    typedef BoxCfg<P, VZ, 32, 16, 1> B16;
    typedef BoxCfg<P, VZ, 32, 8, 1, 80> B8h;
    typedef BoxCfg<P, VZ, 32, 8, 1> B8;
    printf("%s{\"part\": \"%s\", \"groups\": %d, \"reads\": %d, \"writes\": %d, \"mixed\": %d, \"box_eligible\": %d, \"march_eligible\": %d,\n"
           "  \"box_128x16\": {\"lds\": %zu, \"ring_reads\": %d, \"xover\": %d}, \"box_128x8_80k\": {\"lds\": %zu, \"ring_reads\": %d}, \"box_128x8\": {\"lds\": %zu, \"ring_reads\": %d}",
           first ? "" : ",\n", name, P::n_groups, P::n_reads, P::n_writes, count_mixed<P>(), (int)box_eligible<P>(), (int)march_eligible<P>(),
           B16::lds_bytes, B16::ring_reads(), B16::XOVER, B8h::lds_bytes, B8h::ring_reads(), B8::lds_bytes, B8::ring_reads());
    // round 6, second half: the plane-ring rule, the centre-only operands the late refill (_lo) holds once, the operands loaded once per block
    int xinv = 0;
    for (int g = 0; g < P::n_groups; g++) xinv += march_x_invariant<P>(g) ? 1 : 0;
    printf(", \"box_wanted\": %d, \"once\": %d, \"x_invariant\": %d", (int)box_wanted<P>(), march_once_count<P>(), xinv);
    clusters<P, 2>(", ");
    clusters<P, 4>(", ");
    // the part lifted to one x plane (ykh_lift2d.hpp: what a 2-D solution's parts are given to the 3-D kernel families as)
    printf(", \"lift2d_shape\": %d", (int)lift2d_shape<P>());
    if constexpr (lift2d_shape<P>()) {
        typedef Lift2D<P> L;
        typedef BoxCfg<L, VZ, 32, 16, 1> LB;
        int max_dy = 0, max_dz = 0, nonzero_dx = 0;
        for (int i = 0; i < L::n_reads; i++) {
            nonzero_dx += L::reads[i].dx != 0;
            if (L::reads[i].dy > max_dy) max_dy = L::reads[i].dy;
            if (L::reads[i].dz > max_dz) max_dz = L::reads[i].dz;
        }
        printf(", \"lifted\": {\"reads\": %d, \"nonzero_dx\": %d, \"max_dy\": %d, \"max_dz\": %d, \"mixed\": %d, \"box_eligible\": %d, \"box_128x16\": {\"lds\": %zu, \"ring_reads\": %d, \"xover\": %d}}",
               L::n_reads, nonzero_dx, max_dy, max_dz, count_mixed<L>(), (int)box_eligible<L>(), LB::lds_bytes, LB::ring_reads(), LB::XOVER);
    }
    printf("}");
}
struct GenTraits {
    typedef YKH_GEN_NS::real_t real_t;
    static constexpr const VarMeta* vars = YKH_GEN_NS::vars;
    static constexpr const DimMeta* dims = YKH_GEN_NS::dims;
    static constexpr int n_vars = YKH_GEN_NS::soln.n_vars;
};
template <class LIST, const int* LEVEL>
void fuse_group(int first, int last, bool first_group) {
    typedef FusedCfg<GenTraits, LIST, LEVEL, 16, 64> C16;
    typedef FusedCfg<GenTraits, LIST, LEVEL, 8, 64> C8;
    int nsv = 0, max_level = 0;
    for (int v = 0; v < GenTraits::n_vars && v < FUSED_MAX_VARS; v++) nsv += C16::tab.first[v] >= 0;
    for (int i = 0; i < LIST::N; i++) if (LEVEL[i] > max_level) max_level = LEVEL[i];
    // the plan's invariant: two scratch vars share a slot only if one's last use lies at an EARLIER level than the other's first write
    int conflicts = 0, unassigned = 0;
    for (int v = 0; v < GenTraits::n_vars && v < FUSED_MAX_VARS; v++) {
        if (C16::tab.first[v] < 0) continue;
        if (C16::tab.slot[v] < 0 || C16::tab.slot[v] >= C16::tab.n_slots) unassigned++;
        for (int w = v + 1; w < GenTraits::n_vars && w < FUSED_MAX_VARS; w++) {
            if (C16::tab.first[w] < 0 || C16::tab.slot[w] != C16::tab.slot[v]) continue;
            const bool disjoint = C16::tab.last[v] < C16::tab.first[w] || C16::tab.last[w] < C16::tab.first[v];
            if (!disjoint) conflicts++;
        }
    }
    printf("%s{\"slot_conflicts\": %d, \"unassigned\": %d, ", first_group ? "" : ",\n", conflicts, unassigned);
    first_group = true;
    printf("%s\"fuse_group\": [%d, %d], \"parts\": %d, \"levels\": %d, \"scratch_vars\": %d, \"slots\": %d, \"parts_ok\": %d, \"halo\": [%d, %d, %d, %d], "
           "\"lds_16x64\": %zu, \"ok_16x64\": %d, \"lds_8x64\": %zu, \"ok_8x64\": %d}",
           "", first, last, LIST::N, max_level + 1, nsv, C16::tab.n_slots, (int)C16::tab.ok, C16::tab.hl0, C16::tab.hr0, C16::tab.hl1, C16::tab.hr1,
           C16::lds_bytes, (int)C16::ok, C8::lds_bytes, (int)C8::ok);
}
int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "fused") {
        printf("[");
        bool firstg = true;
#define YKH_PROBE_FG(LIST, LEVEL, FIRST, LAST) fuse_group<LIST, LEVEL>(FIRST, LAST, firstg); firstg = false;
        YKH_FOR_EACH_FUSE_GROUP(YKH_PROBE_FG)
        printf("]\n");
        return 0;
    }
    printf("[");
    bool first = true;
#define YKH_PROBE(PART) part<PART>(#PART, first); first = false;
    YKH_FOR_EACH_PART(YKH_PROBE)
    printf("]\n");
    return 0;
}
