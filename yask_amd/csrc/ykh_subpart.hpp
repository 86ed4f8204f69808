// ykh_subpart.hpp -- a part evaluated as several CLUSTERS of its equations, each a kernel of its own.
//
// The reference bundles every equation of a stage that may be evaluated together into one part (src/compiler/lib/Eqs.cpp
// make_parts; `-[no-]bundle`), and its CPU kernel evaluates the whole bundle per vector with the operands in cache.  On the GPU a
// bundle like fsg's stress update -- 24 equations, 435 reads over 57 arrays -- fits no marching kernel (LDS slabs of a dozen
// velocity arrays, register queues, 21 coefficient operands: hipcc runs out of registers) and runs on the point kernel at 330
// VGPRs and 2.45 TB/s, re-fetching the x-neighbour planes of 57 arrays that no L2 holds (profiles/r5_generic).  `-no-bundle`
// gives 36 one-equation parts that read 408 array sweeps instead of 81.  In between: K clusters of equations that share most
// of their inputs (fsg: the six stress components of one sub-grid share their nine velocity derivatives), each small enough for
// the marching kernel, launched one after the other.  The equations of a part do not depend on each other, so any order is the
// part's result -- provided no cluster reads a var another cluster of the part writes (in-place step slots): checked at compile
// time, `clusters_legal`.
//
// No second code generator: SubPart<P, WM> presents part P restricted to the equations in the bit mask WM (bit k = the equation
// that writes P::writes[k]) -- its reads[] are the reads those equations use (P::read_wmask, emitted by the compiler target,
// yask_amd/compiler/YaskHip.cpp), and eval() runs P::eval() through an accessor whose wr<G>() drops the other equations'
// results and whose rd<>() returns a constant for reads outside the table: hipcc's dead-code elimination removes everything
// that only fed the dropped writes.
#pragma once
#include "ykh_meta.hpp"

namespace ykh {

template <class P, unsigned long long WM>
struct SubPartTab {
    ReadOff reads[P::n_reads > 0 ? P::n_reads : 1];
    int n_reads;
    int writes[P::n_writes > 0 ? P::n_writes : 1];
    int n_writes;
};
template <class P, unsigned long long WM>
constexpr SubPartTab<P, WM> make_subpart_tab() {
    SubPartTab<P, WM> t = {};
    for (int i = 0; i < P::n_reads; i++)
        if (P::read_wmask[i] & WM) t.reads[t.n_reads++] = P::reads[i];
    for (int k = 0; k < P::n_writes; k++)
        if ((WM >> k) & 1ull) t.writes[t.n_writes++] = P::writes[k];
    return t;
}

template <class A, class P, unsigned long long WM>
struct MaskAcc {
    typedef typename A::V V;
    A& a;
    static constexpr bool writes_group(int g) {
        for (int k = 0; k < P::n_writes; k++)
            if (P::writes[k] == g) return ((WM >> k) & 1ull) != 0;
        return false;
    }
    static constexpr bool has_read(int g, int dx, int dy, int dz) {
        for (int i = 0; i < P::n_reads; i++)
            if (P::reads[i].g == g && P::reads[i].dx == dx && P::reads[i].dy == dy && P::reads[i].dz == dz) return (P::read_wmask[i] & WM) != 0;
        return false;
    }
    template <int G, int DX, int DY, int DZ>
    __device__ __forceinline__ V rd() const {
        if constexpr (has_read(G, DX, DY, DZ)) return a.template rd<G, DX, DY, DZ>();
        else return V(typename P::real_t(0));        // (feeds dropped equations only: dead code)
    }
    template <int G>
    __device__ __forceinline__ void wr(V v) { if constexpr (writes_group(G)) a.template wr<G>(v); }
    __device__ __forceinline__ void pin(V&) const {}      // (a pinned temporary of a dropped equation would stay alive)
    template <class L, class R> __device__ __forceinline__ V sub(L l, R r) const { return a.sub(l, r); }
    template <class L, class R> __device__ __forceinline__ V div(L l, R r) const { return a.div(l, r); }
    template <int D> __device__ __forceinline__ V idx() const { return a.template idx<D>(); }
    __device__ __forceinline__ V step() const { return a.step(); }
};

template <class P, unsigned long long WM>
struct SubPart {
    typedef typename P::real_t real_t;
    static constexpr int n_groups = P::n_groups;
    static constexpr const AccessGroup (&groups)[P::n_groups] = P::groups;
    static constexpr const bool (&group_full)[P::n_groups] = P::group_full;
    static constexpr const unsigned char (&group_dims)[P::n_groups] = P::group_dims;
    static constexpr SubPartTab<P, WM> tab = make_subpart_tab<P, WM>();
    static constexpr int n_reads = tab.n_reads;
    static constexpr const ReadOff (&reads)[P::n_reads > 0 ? P::n_reads : 1] = tab.reads;
    static constexpr int n_writes = tab.n_writes;
    static constexpr const int (&writes)[P::n_writes > 0 ? P::n_writes : 1] = tab.writes;
    template <class A>
    __device__ __forceinline__ static void eval(A& a) {
        MaskAcc<A, P, WM> m{a};
        P::eval(m);
    }
    static constexpr bool has_lin = false;
    static constexpr bool has_domain_cond = P::has_domain_cond;
    template <class A>
    __device__ __forceinline__ static bool cond(const A& a) { return P::cond(a); }
    static constexpr bool has_step_cond = P::has_step_cond;
    static bool step_cond(long long t) { return P::step_cond(t); }
    static constexpr bool has_step_cond_dev = P::has_step_cond_dev;
    template <class A>
    __device__ __forceinline__ static bool step_cond_dev(const A& a) { return P::step_cond_dev(a); }
};

// K clusters of consecutive equations (in the order of P::writes[], which is the order of the stencil's own equations:
// fsg lists the six stress components of a sub-grid one after the other).
template <class P, int K>
constexpr unsigned long long cluster_mask(int c) {
    const int n = P::n_writes, per = (n + K - 1) / K;
    unsigned long long m = 0;
    for (int k = c * per; k < (c + 1) * per && k < n; k++) m |= 1ull << k;
    return m;
}
// (same var, same constant misc indices, same outer-dim offset: the same array but for the step slot -- fsg2 keeps its 12 velocity
//  and 24 stress components in two vars with a misc dim)
constexpr bool same_storage(const AccessGroup& a, const AccessGroup& b) {
    if (a.var != b.var || a.nmisc != b.nmisc || a.dw != b.dw) return false;
    for (int i = 0; i < a.nmisc; i++)
        if (a.misc[i] != b.misc[i]) return false;
    return true;
}
// No cluster may read a var that another cluster of the part writes: a var written with one step slot (in place) would hand the
// later cluster the new value.  (Reads of the var an equation itself updates are the common case and stay inside its cluster.)
template <class P, int K>
constexpr bool clusters_legal() {
    if (P::n_writes < K || P::n_writes > 64) return false;
    if (!P::has_read_wmask) return false;        // (two equations write one group, or > 64 equations: the compiler target says so)
    for (int k = 0; k < P::n_writes; k++) {
        const int gw = P::writes[k];
        unsigned long long mine = 0;
        for (int c = 0; c < K; c++)
            if ((cluster_mask<P, K>(c) >> k) & 1ull) mine = cluster_mask<P, K>(c);
        for (int i = 0; i < P::n_reads; i++)
            if (same_storage(P::groups[P::reads[i].g], P::groups[gw]) && (P::read_wmask[i] & ~mine)) return false;
    }
    return true;
}

}  // namespace ykh
