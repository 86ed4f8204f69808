// ykh_fused.hpp -- a run of scratch stages and the stage they feed as ONE kernel: scratch vars live in the LDS.
//
// The reference evaluates scratch vars per micro-block: a thread computes the scratch values its block needs (the block grown by the
// scratch halos) into a small per-thread array that stays in cache, then the equations that read them
// (StencilBundleBase::calc_micro_block -> scratch bundles first, src/kernel/lib/stencil_calc.cpp:40-289; halos accumulated along the
// chain by the compiler, src/compiler/lib/Eqs.cpp "scratch" + Var::update_halo).  The GPU runtime so far gave every scratch var a
// whole device array and every scratch part a sweep of the grid: swe2d = 61 scratch parts + 4 = 65 sweeps per step, wave2d 12 + 3,
// moving ~5 TB/s of scratch data through HBM (profiles/r6_generic).  Here the micro-block is a workgroup's tile:
//
//   * the compiler target lists, per consuming stage, the parts of the scratch stages that feed it and of the stage itself in
//     evaluation order with their LEVEL (gen/*.hpp: fuse_group_N, fuse_group_N_level) -- parts of a level are independent;
//   * a workgroup owns a TI x TJ tile of the (d0, d1) plane; every scratch var of the group has a (TI + halos) x (TJ + halos) array
//     in the LDS -- one SLOT, shared by vars whose live ranges (first write .. last read, in levels) do not overlap;
//   * level by level (one barrier between levels) the threads evaluate each part over the tile grown by the halo of the scratch var
//     it writes -- P::eval() through an accessor that maps scratch groups to LDS slots and everything else to global memory; the last
//     level's parts (the non-scratch ones) run over the tile itself and store to the output vars;
//   * sub-domain conditions are evaluated per point (P::cond), as the scalar point kernel does.
// Redundant work: the halo ring of a tile is computed by its neighbours too ((TI + h)(TJ + h) / (TI TJ), ~1.5x at h = 4) -- flops
// and LDS reads instead of 2 HBM sweeps per scratch var.
// 2-D solutions only (a 3-D tile with halos of 4 does not leave room for 40 scratch vars).  A decomposed rank runs the group over its
// whole box and exchanges afterwards (Solution::run).
#pragma once
#include "ykh_device.hpp"

namespace ykh {

struct FusedGeom {
    int i0, i1, j0, j1;      // the box of the consuming stage (rank box), local indices
    int ntj;                 // tiles along j
    long long t;             // the step being evaluated (the PartArgs tables are per step-slot PHASE: their own `t` is not it)
};

constexpr int FUSED_MAX_VARS = 128;

struct FusedTab {
    int first[FUSED_MAX_VARS], last[FUSED_MAX_VARS];      // level of the first write / last use of every scratch var; -1: not in this group
    int slot[FUSED_MAX_VARS];
    int n_slots;
    int hl0, hr0, hl1, hr1;                               // largest halos among the scratch vars of the group
    bool ok;                                              // every part can be fused
};

template <class TR, class P>
constexpr void fused_note(FusedTab& t, int level) {
    // (step conditions: a host-side one that is false at some step sends that step down the unfused path, Solution::fused_ok_at();
    //  a device-side one -- it reads var values -- is evaluated per point here, like the scalar point kernel does per launch)
    for (int i = 0; i < P::n_reads; i++)
        if (P::reads[i].dz != 0) t.ok = false;
    for (int g = 0; g < P::n_groups; g++) {
        const int v = P::groups[g].var;
        if (v >= FUSED_MAX_VARS) { t.ok = false; continue; }
        if (!TR::vars[v].is_scratch) continue;
        if (t.first[v] < 0) t.first[v] = level;
        if (level > t.last[v]) t.last[v] = level;
    }
}

template <class TR, class LIST, const int* LEVEL> struct FusedPlanTab;
template <class TR, const int* LEVEL, class... Ps>
struct FusedPlanTab<TR, PartList<Ps...>, LEVEL> {
    static constexpr FusedTab make() {
        FusedTab t = {};
        t.ok = TR::n_vars <= FUSED_MAX_VARS;
        for (int v = 0; v < FUSED_MAX_VARS; v++) { t.first[v] = -1; t.last[v] = -1; t.slot[v] = -1; }
        int k = 0;
        (fused_note<TR, Ps>(t, LEVEL[k++]), ...);
        // slots: vars in the order of their first level take the lowest slot whose occupant was last used at an EARLIER level (a
        // barrier separates levels, so the last reader is done before the first writer starts)
        int free_after[FUSED_MAX_VARS] = {};
        int n = 0;
        int max_level = 0;
        for (int i = 0; i < (int)sizeof...(Ps); i++) if (LEVEL[i] > max_level) max_level = LEVEL[i];
        for (int lv = 0; lv <= max_level; lv++)
            for (int v = 0; v < TR::n_vars && v < FUSED_MAX_VARS; v++) {
                if (t.first[v] != lv) continue;
                int s = -1;
                for (int c = 0; c < n; c++) if (free_after[c] < lv) { s = c; break; }
                if (s < 0) s = n++;
                t.slot[v] = s;
                free_after[s] = t.last[v];
                if (TR::vars[v].halo_l[0] > t.hl0) t.hl0 = TR::vars[v].halo_l[0];
                if (TR::vars[v].halo_r[0] > t.hr0) t.hr0 = TR::vars[v].halo_r[0];
                if (TR::vars[v].halo_l[1] > t.hl1) t.hl1 = TR::vars[v].halo_l[1];
                if (TR::vars[v].halo_r[1] > t.hr1) t.hr1 = TR::vars[v].halo_r[1];
            }
        t.n_slots = n;
        return t;
    }
    static constexpr FusedTab tab = make();
};

template <class TR, class LIST, const int* LEVEL, int TI, int TJ>
struct FusedCfg {
    typedef FusedPlanTab<TR, LIST, LEVEL> PT;
    static constexpr FusedTab tab = PT::tab;
    static constexpr int ROWS = TI + tab.hl0 + tab.hr0, PITCH = TJ + tab.hl1 + tab.hr1;
    static constexpr int SLOT_ELEMS = ROWS * PITCH;
    static constexpr size_t lds_bytes = (size_t)(tab.n_slots > 0 ? tab.n_slots : 1) * SLOT_ELEMS * sizeof(typename TR::real_t);
    static constexpr bool ok = tab.ok && tab.n_slots > 0 && lds_bytes <= 160 * 1024;
};

// a var with neither domain nor misc dims that no equation writes: one value per (step slot)
template <class TR>
constexpr bool fused_var_is_scalar(int v) {
    const VarMeta& vm = TR::vars[v];
    if (vm.is_written || vm.is_scratch) return false;
    for (int d = 0; d < vm.ndims; d++)
        if (TR::dims[vm.dims[d]].type != DIM_STEP) return false;
    return true;
}

// accessor of one point (i, j) of a part inside a fused group
template <class TR, class C, class P>
struct FusedAcc {
    typedef typename P::real_t T;
    typedef T V;
    const PartArgs& a;
    T* lds;
    int i, j;             // the point, local indices
    int ti0, tj0;         // first point of the tile
    long long t;          // the step
    template <int G>
    __device__ __forceinline__ T* slot_ptr(int di, int dj) const {
        constexpr int v = P::groups[G].var;
        constexpr int s = C::tab.slot[v];
        static_assert(s >= 0, "scratch var without a slot");
        return lds + s * C::SLOT_ELEMS + (i + di - ti0 + C::tab.hl0) * C::PITCH + (j + dj - tj0 + C::tab.hl1);
    }
    template <int G, int DI, int DJ, int DZ>
    __device__ __forceinline__ V rd() const {
        if constexpr (TR::vars[P::groups[G].var].is_scratch) return *slot_ptr<G>(DI, DJ);
        else if constexpr (fused_var_is_scalar<TR>(P::groups[G].var)) {
            // a var without domain dims (swe2d's dt(), g(), inv_dx() ... are read by most of its 65 parts): one value for the launch,
            // never written by it -- a scalar load through the constant address space, not 64 lanes fetching the same word (by the SQ
            // counters these were a third of the kernel's vector-memory instructions)
            typedef const T __attribute__((address_space(4))) CT;
            return *(CT*)(const T*)a.ptr[G];
        } else {
            // (32-bit offsets: Solution::fused_usable() admits the fused path only while every var has fewer than 2^31 elements)
            const T* p = (const T*)a.ptr[G];
            return p[(i + DI) * (int)a.gsx[G] + (j + DJ) * (int)a.gsy[G]];
        }
    }
    template <int G>
    __device__ __forceinline__ void wr(V v) const {
        if constexpr (TR::vars[P::groups[G].var].is_scratch) *slot_ptr<G>(0, 0) = v;
        else {
            T* p = (T*)a.ptr[G];
            p[i * (int)a.gsx[G] + j * (int)a.gsy[G]] = v;
        }
    }
    __device__ __forceinline__ void pin(V&) const {}
    template <class L, class R> __device__ __forceinline__ V sub(L l, R r) const { return V(l) - V(r); }
    template <class L, class R> __device__ __forceinline__ V div(L l, R r) const { return V(l) / V(r); }
    template <int D> __device__ __forceinline__ V idx() const { return V(D == 0 ? i + a.ofs_x : j + a.ofs_y); }
    __device__ __forceinline__ V step() const { return V(t); }
    template <int D> __device__ __forceinline__ long long sidx() const { return D == 0 ? i + a.ofs_x : j + a.ofs_y; }
    template <int D> __device__ __forceinline__ long long first_idx() const { return 0; }
    template <int D> __device__ __forceinline__ long long last_idx() const { return D == 0 ? a.glast_x : a.glast_y; }
    __device__ __forceinline__ long long sstep() const { return t; }
};

// halo of the region a part is evaluated over: the largest halo of the scratch vars it writes (0 for the consuming parts)
struct FusedGrow { int l0, r0, l1, r1; bool scratch; };
template <class TR, class P>
constexpr FusedGrow fused_grow() {
    FusedGrow g = {0, 0, 0, 0, false};
    for (int w = 0; w < P::n_writes; w++) {
        const VarMeta& vm = TR::vars[P::groups[P::writes[w]].var];
        if (!vm.is_scratch) continue;
        g.scratch = true;
        if (vm.halo_l[0] > g.l0) g.l0 = vm.halo_l[0];
        if (vm.halo_r[0] > g.r0) g.r0 = vm.halo_r[0];
        if (vm.halo_l[1] > g.l1) g.l1 = vm.halo_l[1];
        if (vm.halo_r[1] > g.r1) g.r1 = vm.halo_r[1];
    }
    return g;
}

// the words of a part's PartArgs the kernel needs before it can do anything: its box, the flag, the hole.  Fetched one part AHEAD
// (fused_part): each part otherwise starts with scalar loads and a wait for them.
struct FusedBoxWords { int x0, x1, y0, y1, nxc, hx0, hx1, hy0, hy1; };
__device__ __forceinline__ FusedBoxWords fused_box_words(const PartArgs& a) { return {a.x0, a.x1, a.y0, a.y1, a.nxc, a.ax0, a.ax1, a.ay0, a.ay1}; }

template <class TR, class C, const int* LEVEL, int TI, int TJ, int NT, int NPARTS, int K, class P>
__device__ __forceinline__ void fused_part(const PartArgs* __restrict__ args, const FusedGeom& g, typename TR::real_t* lds, int ti0, int tj0,
                                           FusedBoxWords& ahead) {
    // (Same-box A/B, job r6za: with these nine words loaded at the top of the part -- BEFORE the level's barrier instead of behind it --
    //  swe2d went 1.69 -> 1.55 ms and wave2d 0.343 -> 0.324: the loads travel while the workgroup waits at the barrier; fetched one
    //  part ahead, 1.53 / 0.318.)
    const FusedBoxWords bw = ahead;
    if constexpr (K + 1 < NPARTS) ahead = fused_box_words(args[K + 1]);
    if constexpr (K > 0) {
        if constexpr (LEVEL[K] != LEVEL[K - 1]) __syncthreads();      // the next level reads what this one wrote (and may re-use its slots)
    }
    constexpr FusedGrow gr = fused_grow<TR, P>();
    constexpr int RI = TI + gr.l0 + gr.r0, RJ = TJ + gr.l1 + gr.r1;
    const PartArgs& a = args[K];
    // The part's region of this tile, clipped to the box the host found for the part (Solution::ensure_fused_args): the box of the
    // consuming stage grown by the halos of the scratch vars the part writes, cut down to the bounding box of its sub-domain condition.
    // a.nxc != 0: the condition holds at every point of that box -- nothing to evaluate per point (most parts: 64-bit index
    // comparisons per point cost more than the equations of a boundary strip).  A tile the box does not reach skips the part.
    const int ri0 = ti0 - gr.l0, rj0 = tj0 - gr.l1;
    const int bi0 = bw.x0, bi1 = bw.x1, bj0 = bw.y0, bj1 = bw.y1;                 // (uniform: scalar loads)
    if (bi0 >= ri0 + RI || bi1 <= ri0 || bj0 >= rj0 + RJ || bj1 <= rj0) return;
    // a.nxc == 2: a ring -- the condition holds in the box except in the hole [ax0, ax1) x [ay0, ay1) (both verified solid by the host;
    // any other part gets an empty hole, so that one test serves all).
    // The per-point tests are ONE vector comparison: distances to the four edges of the box and of the hole through v_min3, not a
    // chain of compares -- every compare result is a lane mask in SGPRs and every && of two masks a scalar instruction; the first
    // version of this loop issued twice as many scalar as vector instructions (7 336 against 3 556 for swe2d's 65 parts) and was bound by
    // the CU's one scalar issue per cycle.
    const bool solid = bw.nxc != 0;
    const bool ring = bw.nxc == 2;
    const int hi0 = ring ? bw.hx0 : 0, hi1 = ring ? bw.hx1 : 0, hj0 = ring ? bw.hy0 : 0, hj1 = ring ? bw.hy1 : 0;
    if (ring && ri0 >= hi0 && ri0 + RI <= hi1 && rj0 >= hj0 && rj0 + RJ <= hj1) return;      // the tile's region lies in the hole
    // (unrolled: the 1-3 iterations of a thread interleave their LDS reads; swe2d 1.72 -> 1.69 ms, wave2d 0.360 -> 0.343, same-box A/B r6z)
#pragma unroll
    for (int idx = threadIdx.x; idx < RI * RJ; idx += NT) {
        const int i = ri0 + idx / RJ, j = rj0 + idx % RJ;
        const int in_box = min(min(i - bi0, bi1 - 1 - i), min(j - bj0, bj1 - 1 - j));         // >= 0: inside the part's box
        const int in_hole = min(min(i - hi0, hi1 - 1 - i), min(j - hj0, hj1 - 1 - j));        // >= 0: inside the hole (never, for an empty one)
        bool act = min(in_box, -1 - in_hole) >= 0;
        FusedAcc<TR, C, P> acc{a, lds, i, j, ti0, tj0, g.t};
        if constexpr (P::has_step_cond_dev) act = act && P::step_cond_dev(acc);               // IF_STEP on var values (uniform)
        if constexpr (P::has_domain_cond) {
            if (!solid) act = act && P::cond(acc);                                            // (neither solid nor a ring: rare)
        }
        if (act) P::eval(acc);
    }
}

template <class TR, class C, const int* LEVEL, int TI, int TJ, int NT, class... Ps, int... Ks>
__device__ __forceinline__ void fused_all(const PartArgs* __restrict__ args, const FusedGeom& g, typename TR::real_t* lds, int ti0, int tj0,
                                          PartList<Ps...>, std::integer_sequence<int, Ks...>) {
    FusedBoxWords ahead = fused_box_words(args[0]);
    (fused_part<TR, C, LEVEL, TI, TJ, NT, (int)sizeof...(Ps), Ks, Ps>(args, g, lds, ti0, tj0, ahead), ...);
}

template <class TR, class LIST, const int* LEVEL, int TI, int TJ, int NT>
__global__ void __launch_bounds__(NT) fused2d_kernel(const PartArgs* __restrict__ args, const FusedGeom g) {
    typedef FusedCfg<TR, LIST, LEVEL, TI, TJ> C;
    typedef typename TR::real_t T;
    extern __shared__ __attribute__((aligned(16))) unsigned char ykh_smem[];
    T* lds = reinterpret_cast<T*>(ykh_smem);
    const int ti0 = g.i0 + (int)(blockIdx.x / (unsigned)g.ntj) * TI, tj0 = g.j0 + (int)(blockIdx.x % (unsigned)g.ntj) * TJ;
    // (Points of a slot that no part defines -- outside every condition -- hold whatever the slot held before, as the reference's per-thread
    //  scratch arrays do (never initialised, stencil_calc.cpp:40-289) and as the device arrays of the part-by-part path do from step to
    //  step: a solution must not read them.  The first version zero-filled the slots per tile -- 36 LDS stores per thread and a barrier:
    //  1.94 -> 1.91 ms for swe2d, 0.372 -> 0.359 for wave2d without it, same-box A/B, job r6q.)
    //  Also measured and NOT kept (jobs r6o / r6p, same box, alternating): a uniform fast path that skips the per-point test on interior
    //  tiles and a fixed-trip-count form of the point loop -- 21 % fewer vector instructions by the SQ counters and 12 % SLOWER (2.16 ms):
    //  the extra uniform branches and scalar work cost more issue slots than the v_min3 tests they saved.
    fused_all<TR, C, LEVEL, TI, TJ, NT>(args, g, lds, ti0, tj0, LIST{}, std::make_integer_sequence<int, LIST::N>{});
}

}  // namespace ykh
