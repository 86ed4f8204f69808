// ykh_box.hpp -- marching kernel with an LDS RING OF PLANES, for parts that read a var at MANY mixed offsets
// (box and plane neighbourhoods: the reference's `cube` 5x5x5, `3plane` 3 x 7x7, `3axis_with_diags`, `tti`).
//
// The other kernels serve such reads badly: the point kernels (ykh_vecpt.hpp) issue one global load per distinct
// (dx, dy, z-vector) and run at the vector-L1 rate of the chip (cube: 75 16-byte loads per 4 points, 42 TB/s of loads for
// 8 B of compulsory traffic per point, profiles/r5_generic); the generic marching kernel (ykh_march.hpp) keeps x-neighbours in
// REGISTER queues and one LDS slab of the centre plane, which covers axis-aligned offsets only -- more than MAX_MIXED
// mixed-offset reads fall back to global loads where they are used.
// Here a workgroup owns a (y, z) tile and marches along x like the other marching kernels, but every group that is read at
// an offset keeps ALL the planes x+xlo .. x+xhi of its tile (+ the y / z halo of the reads) in an LDS ring of xhi-xlo+2
// slots: each plane of the tile comes over the fabric once (+ tile halos), every read is a 16-byte LDS load at
// [slot(x+dx)] + thread offset + a compile-time (dy, dz) displacement, and the LDS pipe delivers twice the bytes per clock of
// the L1.  One barrier per plane: the plane x+xhi+1 is requested at the top of iteration x into registers, stored into the
// ring's free slot after plane x has been evaluated, and becomes readable behind the barrier.  Operands read at the centre
// only are prefetched one plane ahead into a register, as in ykh_march.hpp.  A thread evaluates RY rows as ONE wide vector
// (BoxAcc): the LDS rows two of its rows share are loaded once.
// Replaces the reference's generated calc_vectors loop + block loops for such parts
// (src/compiler/lib/YaskKernel.cpp:591-719, src/kernel/lib/stencil_calc.cpp:40-289).
#pragma once
#include "ykh_device.hpp"
#include "ykh_starlin.hpp"   // vecn, zshiftn, ldv/stv, sbase
#include "ykh_march.hpp"     // MAX_MIXED

namespace ykh {

struct BoxShape {
    int xlo, xhi, ylo, yhi, zlo, zhi;   // ranges over ALL reads of the group
    bool any, offs, written;
};
template <class P>
constexpr BoxShape box_shape(int g) {
    BoxShape s = {0, 0, 0, 0, 0, 0, false, false, false};
    for (int i = 0; i < P::n_reads; i++) {
        if (P::reads[i].g != g) continue;
        s.any = true;
        const int dx = P::reads[i].dx, dy = P::reads[i].dy, dz = P::reads[i].dz;
        if (dx || dy || dz) s.offs = true;
        if (dx < s.xlo) s.xlo = dx;
        if (dx > s.xhi) s.xhi = dx;
        if (dy < s.ylo) s.ylo = dy;
        if (dy > s.yhi) s.yhi = dy;
        if (dz < s.zlo) s.zlo = dz;
        if (dz > s.zhi) s.zhi = dz;
    }
    for (int i = 0; i < P::n_writes; i++)
        if (P::writes[i] == g) s.written = true;
    return s;
}
// distinct mixed-offset reads of a part (more than MAX_MIXED: the marching kernel of ykh_march.hpp has no good answer)
template <class P>
constexpr int count_mixed() {
    int n = 0;
    for (int i = 0; i < P::n_reads; i++) {
        if ((P::reads[i].dx != 0) + (P::reads[i].dy != 0) + (P::reads[i].dz != 0) < 2) continue;
        bool seen = false;
        for (int k = 0; k < i; k++)
            if (P::reads[k].g == P::reads[i].g && P::reads[k].dx == P::reads[i].dx && P::reads[k].dy == P::reads[i].dy && P::reads[k].dz == P::reads[i].dz) seen = true;
        if (!seen) n++;
    }
    return n;
}
// Which parts get plane-ring shapes: more mixed-offset reads than the marching kernel prefetches (MAX_MIXED: cube, 3plane, tti ...), or
// a small part that is MOSTLY mixed reads -- the reference's test_3d / test_stages_3d / test_boundary_3d read the 8 corners of a box
// (-2..4, -6..5, -4..3) around the point: on the point kernel every corner plane crosses the fabric again 6 planes later (test_3d
// 0.353 ms at 512^3), with the 7 planes in an LDS ring 0.241 (job r6zl).
template <class P>
constexpr bool box_wanted() { return count_mixed<P>() > MAX_MIXED || (count_mixed<P>() >= 4 && 2 * count_mixed<P>() + 1 >= P::n_reads); }
// Written groups are vars over all domain dims (shared strides / pads).  Groups read at an offset may be vars over a SUBSET of the
// domain dims since round 6 (test_partial_3d: 1-D and 2-D coefficient tables read at offsets): they never get a ring -- a table
// that lacks a dim is small and stays in the L1 / L2 -- and are loaded where they are used through their own strides (kind 4).
template <class P>
constexpr bool box_eligible() {
    for (int g = 0; g < P::n_groups; g++) {
        const BoxShape s = box_shape<P>(g);
        if (s.written && !P::group_full[g]) return false;
    }
    return P::n_groups <= MAX_GROUPS;
}

// some group of the part is a var over a subset of the domain dims that is read at an offset (kind 4 of BoxTab)
template <class P>
constexpr bool box_has_tables() {
    for (int g = 0; g < P::n_groups; g++) {
        const BoxShape s = box_shape<P>(g);
        if (s.offs && !P::group_full[g]) return true;
    }
    return false;
}

struct BoxTab {
    int kind[MAX_GROUPS];               // 0: not read, 1: centre only (prefetch register), 2: ring of planes in the LDS, 3: global loads where used,
                                        // 4: a var over a subset of the domain dims read at offsets: global loads through its own strides
    int xlo[MAX_GROUPS], nx[MAX_GROUPS], nr[MAX_GROUPS];      // planes x+xlo .. x+xlo+nx-1 are live; nr = nx + 1 slots
    int yl[MAX_GROUPS], zlv[MAX_GROUPS], lp[MAX_GROUPS], lpv[MAX_GROUPS], lrows[MAX_GROUPS], plane[MAX_GROUPS];
    int roff[MAX_GROUPS + 1];           // element offset of the group's ring
    int nv[MAX_GROUPS], nvt[MAX_GROUPS], voff[MAX_GROUPS + 1];   // vectors of one plane of the slab; per thread; index of the first
    int soff[MAX_GROUPS + 1];           // index of the group's first slot base (one per live plane)
};

// LDS_KB: budget of the rings.  Groups are taken in the order of their read counts while their rings fit; a group that is read
// at an offset and does not fit is loaded from global memory where it is used (L1 / L2 served, as the point kernels do for
// every read) -- tti: u(t) and v(t) with 47 reads each take 78 KB at a 128 x 8 tile, its four 10-read coefficient arrays stay out.
template <class P, int VZ_, int TZL_, int TYL_, int RY_, int LDS_KB_ = 160>
struct BoxCfg {
    typedef typename P::real_t T;
    static constexpr int VZ = VZ_, TZL = TZL_, TYL = TYL_, RY = RY_, NT = TZL_ * TYL_, NG = P::n_groups;
    static constexpr int TZ = TZL * VZ, TY = TYL * RY;
    static constexpr BoxTab make() {
        BoxTab t = {};
        int nreads[MAX_GROUPS] = {};
        bool taken[MAX_GROUPS] = {};
        for (int i = 0; i < P::n_reads; i++) nreads[P::reads[i].g]++;
        for (int g = 0; g < NG; g++) {
            const BoxShape s = box_shape<P>(g);
            t.kind[g] = !s.any ? 0 : (s.offs ? (P::group_full[g] ? 3 : 4) : 1);
            if (t.kind[g] != 3) continue;
            t.xlo[g] = s.xlo; t.nx[g] = s.xhi - s.xlo + 1; t.nr[g] = t.nx[g] + 1;
            t.yl[g] = -s.ylo;
            t.zlv[g] = (-s.zlo + VZ - 1) / VZ;
            const int zhv = (s.zhi + VZ - 1) / VZ;
            t.lpv[g] = TZL + t.zlv[g] + zhv;
            t.lp[g] = t.lpv[g] * VZ;
            t.lrows[g] = TY - s.ylo + s.yhi;
            t.plane[g] = t.lrows[g] * t.lp[g];
            t.nv[g] = t.lrows[g] * t.lpv[g];
            t.nvt[g] = (t.nv[g] + NT - 1) / NT;
        }
        long long used = 0;
        for (int round = 0; round < NG; round++) {          // most-read group first
            int best = -1;
            for (int g = 0; g < NG; g++)
                if (t.kind[g] == 3 && !taken[g] && (best < 0 || nreads[g] > nreads[best])) best = g;
            if (best < 0) break;
            taken[best] = true;
            const long long bytes = (long long)t.nr[best] * t.plane[best] * (long long)sizeof(T);
            if (used + bytes <= (long long)LDS_KB_ * 1024) { t.kind[best] = 2; used += bytes; }
        }
        int ro = 0, vo = 0, so = 0;
        for (int g = 0; g < NG; g++) {
            t.roff[g] = ro; t.voff[g] = vo; t.soff[g] = so;
            if (t.kind[g] != 2) continue;
            ro += t.nr[g] * t.plane[g];
            vo += t.nvt[g];
            so += t.nx[g];
        }
        t.roff[NG] = ro; t.voff[NG] = vo; t.soff[NG] = so;
        return t;
    }
    static constexpr BoxTab tab = make();
    // the largest y / z reach of the groups that are loaded where they are used (kind 3): the thread's own row offset is clamped ONCE so
    // that every such read stays inside the allocation, and a read is then that offset + a uniform term (BoxAcc::rd)
    struct K3 { int yl, yh, zlv, zhv; };
    static constexpr K3 make_k3() {
        K3 k = {0, 0, 0, 1};
        for (int g = 0; g < NG; g++) {
            if (tab.kind[g] != 3) continue;
            const BoxShape s = box_shape<P>(g);
            if (-s.ylo > k.yl) k.yl = -s.ylo;
            if (s.yhi > k.yh) k.yh = s.yhi;
            if ((-s.zlo + VZ - 1) / VZ > k.zlv) k.zlv = (-s.zlo + VZ - 1) / VZ;
            if ((s.zhi + VZ - 1) / VZ + 1 > k.zhv) k.zhv = (s.zhi + VZ - 1) / VZ + 1;       // (+1: the second vector of a shifted pair)
        }
        return k;
    }
    static constexpr K3 k3 = make_k3();
    static constexpr int RING_TOT = tab.roff[NG], NVTOT = tab.voff[NG], NSTOT = tab.soff[NG];
    static constexpr size_t lds_bytes = sizeof(T) * (size_t)(RING_TOT > 0 ? RING_TOT : 1);
    // reads served from the LDS rings -- or from small tables that lack a dim (kind 4: L1-resident); what is left are the kind-3 reads
    static constexpr int ring_reads() { int n = 0; for (int i = 0; i < P::n_reads; i++) if (tab.kind[P::reads[i].g] == 2 || tab.kind[P::reads[i].g] == 4) n++; return n; }
    static constexpr int max_nx() { int m = 1; for (int g = 0; g < NG; g++) if (tab.kind[g] == 2 && tab.nx[g] > m) m = tab.nx[g]; return m; }
    static constexpr int XOVER = max_nx();      // a block fills its rings before its first plane
};

// The accessor evaluates the RY rows of a thread AT ONCE: its value type is a vector of RY * VZ elements (row 0's z-vector, row 1's,
// ...), so the generated expression is walked once and every read brings RY rows -- reads that differ by one in dy share RY - 1 of
// their LDS rows, and being loads of one basic block with no store between them hipcc merges them (cube, RY = 2: 90 LDS loads per
// 8 points instead of 150).  Two separate evaluations cannot share anything: their loads are hoisted wholesale and spill.
// (Honouring the generated code's pin() after every temporary -- strict program order, as ykh_march.hpp offers -- was tried for the
// parts that spill: with a memory clobber, or with a plain register barrier, hipcc repeats the LDS loads per temporary, tti 3.5x as
// many; what brought tti inside the register file instead is 8-byte lanes, stencil_generic.hip.)
template <class C, class P>
struct BoxAcc {
    typedef typename C::T T;
    static constexpr int VZ = C::VZ, RY = C::RY;
    typedef typename vecn<T, VZ>::type V1;             // one row
    typedef typename vecn<T, VZ * RY>::type V;         // the thread's RY rows
    const PartArgs& a;
    const T* ring;
    const int (&sl)[C::NSTOT > 0 ? C::NSTOT : 1];     // element offset of the slot that holds plane x+xlo+i of a group (uniform)
    const int (&tofs)[C::NG];                          // the thread's first row and z lane within a group's slab
    const V (&nx)[C::NG];                              // centre-only operands
    int x, y, z0;            // first row, first of the VZ points
    V (&out)[MAX_GROUPS];
    const unsigned (&o3)[C::RY];                       // byte offset of the thread's rows within a plane, clamped for the kind-3 reads
    template <class F>
    __device__ __forceinline__ static V rows(F f) {   // f(row) -> V1
        if constexpr (RY == 1) return f(std::integral_constant<int, 0>{});
        else {
            V r;
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const V1 v = f(jc);
                static_for<VZ>([&](auto ec) { constexpr int e = decltype(ec)::value; r[j * VZ + e] = v[e]; });
            });
            return r;
        }
    }
    template <class L, class R> __device__ __forceinline__ V sub(L l, R r) const { return V(l) - V(r); }
    template <class L, class R> __device__ __forceinline__ V div(L l, R r) const { return V(l) / V(r); }
    template <int G, int DX, int DY, int DZ>
    __device__ __forceinline__ V rd() const {
        constexpr int qq = (DZ >= 0) ? DZ / VZ : -((-DZ + VZ - 1) / VZ);      // floor(DZ / VZ)
        constexpr int e = DZ - qq * VZ;
        if constexpr (C::tab.kind[G] == 1) return nx[G];
        else if constexpr (C::tab.kind[G] == 3) {
            // a group whose ring did not fit: aligned global loads where the value is used.  The address is (uniform plane base, in
            // SGPRs) + (the thread's row offset, clamped once for all such reads: o3) + (a uniform term for this read's dy / dz) -- the
            // saddr form of global_load with one 32-bit add.  (Until round 6 every such read clamped y and z itself and multiplied by
            // the run-time row pitch in 64 bits: tti, 40 of these per point.)  Threads of a ragged tile's overhang compute on whatever
            // lies at the clamped rows and store nothing.
            const idx_t org = (idx_t)a.ay0 * a.sy + a.az0;
            // (no sbase() here: its opaque SGPR pair per call -- 40 of them in tti -- spilled the scalar file into vector registers)
            const T* pb = (const T*)a.ptr[G] + (org + (idx_t)clampi(x + DX, a.ax0, a.ax1 - 1) * a.sx);
            const int d = (DY * (int)a.sy + qq * VZ) * (int)sizeof(T);                               // uniform
            return rows([&](auto jc) -> V1 {
                constexpr int j = decltype(jc)::value;
                const unsigned o = o3[j] + (unsigned)d;
                if constexpr (e == 0) return ldv_b<V1>(pb, o);
                else return zshiftn<T, VZ, e>(ldv_b<V1>(pb, o), ldv_b<V1>(pb, o + (unsigned)(VZ * sizeof(T))));
            });
        } else if constexpr (C::tab.kind[G] == 4) {
            // a var over a subset of the domain dims (its strides are 0 in the dims it lacks), read at an offset: every var shares the
            // solution's pads in the dims it has (Var::compute_geometry), so the clamps of the shared layout hold for it too
            constexpr unsigned gd = GroupDims<P>::get(G);
            if constexpr (gd != 7) {
                // the table's dims at compile time (the compiler target's group_dims): no multiply for a dim it lacks, ONE value per
                // row where it has no z (79 live 16-byte operands were what spilled test_partial_3d on every shape of this kernel)
                const T* px = (const T*)a.ptr[G];
                if constexpr ((gd & 1) != 0) px += (idx_t)clampi(x + DX, a.ax0, a.ax1 - 1) * a.gsx[G];
                if constexpr ((gd & 4) == 0)
                    return rows([&](auto jc) -> V1 {
                        constexpr int j = decltype(jc)::value;
                        if constexpr ((gd & 2) != 0) return V1(px[(idx_t)clampi(y + j + DY, a.ay0, a.ay1 - 1) * a.gsy[G]]);
                        else return V1(px[0]);
                    });
                else {
                    const int zl = clampi(z0 + qq * VZ, a.az0, a.az1 - VZ), zh = clampi(z0 + (qq + 1) * VZ, a.az0, a.az1 - VZ);
                    return rows([&](auto jc) -> V1 {
                        constexpr int j = decltype(jc)::value;
                        const T* p = px;
                        if constexpr ((gd & 2) != 0) p += (idx_t)clampi(y + j + DY, a.ay0, a.ay1 - 1) * a.gsy[G];
                        if constexpr (e == 0) return ldv<V1>(p + zl);
                        else return zshiftn<T, VZ, e>(ldv<V1>(p + zl), ldv<V1>(p + zh));
                    });
                }
            }
            const T* px = (const T*)a.ptr[G] + (idx_t)clampi(x + DX, a.ax0, a.ax1 - 1) * a.gsx[G];
            if (a.gsz[G] == 0)          // no unit-stride dim: one value per row (uniform branch)
                return rows([&](auto jc) -> V1 {
                    constexpr int j = decltype(jc)::value;
                    return V1(px[(idx_t)clampi(y + j + DY, a.ay0, a.ay1 - 1) * a.gsy[G]]);
                });
            const int zl = clampi(z0 + qq * VZ, a.az0, a.az1 - VZ), zh = clampi(z0 + (qq + 1) * VZ, a.az0, a.az1 - VZ);
            return rows([&](auto jc) -> V1 {
                constexpr int j = decltype(jc)::value;
                const T* p = px + (idx_t)clampi(y + j + DY, a.ay0, a.ay1 - 1) * a.gsy[G];
                if constexpr (e == 0) return ldv<V1>(p + zl);
                else return zshiftn<T, VZ, e>(ldv<V1>(p + zl), ldv<V1>(p + zh));
            });
        } else {
            static_assert(C::tab.kind[G] == 2, "read of a group the table does not hold");
            constexpr int LP = C::tab.lp[G];
            const T* base = ring + sl[C::tab.soff[G] + DX - C::tab.xlo[G]] + tofs[G];
            return rows([&](auto jc) -> V1 {
                constexpr int j = decltype(jc)::value;
                const T* row = base + ((j + DY) * LP + qq * VZ);
                if constexpr (e == 0) return ldv<V1>(row);
                else return zshiftn<T, VZ, e>(ldv<V1>(row), ldv<V1>(row + VZ));
            });
        }
    }
    template <int G>
    __device__ __forceinline__ void wr(V v) { out[G] = v; }
    __device__ __forceinline__ void pin(V&) const {}
    template <int D>
    __device__ __forceinline__ V idx() const {
        return rows([&](auto jc) -> V1 {
            constexpr int j = decltype(jc)::value;
            if constexpr (D == 2) { V1 r; static_for<VZ>([&](auto ec) { constexpr int e2 = decltype(ec)::value; r[e2] = T(z0 + e2 + a.ofs_z); }); return r; }
            else return V1(T(D == 0 ? x + a.ofs_x : y + j + a.ofs_y));
        });
    }
    __device__ __forceinline__ V step() const { return V(T(a.t)); }
};

// FL & 1: non-temporal output stores and centre-only operand loads (one-touch streams);
// FL & 4: planes are requested TWO iterations before they are stored into the ring (two register sets; a workgroup's plane takes
// 1-2.5 us, about a loaded HBM round trip)
// DESC: the twin that takes its tile and x range from a block descriptor (planned launches of a decomposed rank, ykh_plan.cpp) and signals
// when done -- as in march_kernel; a block is self-contained (its prologue fills its own rings), so the planner may cut the box freely.
template <class P, int VZ, int TZL, int TYL, int RY, int MINW, int FL = 0, int LDS_KB = 160, bool DESC = false>
__global__ void __launch_bounds__(TZL* TYL, MINW) box_kernel(const PartArgs a) {
    typedef BoxCfg<P, VZ, TZL, TYL, RY, LDS_KB> C;
    typedef typename C::T T;
    typedef typename vecn<T, VZ>::type V;
    typedef typename vecn<T, VZ * RY>::type VW;
    constexpr int NG = C::NG, NT = C::NT;
    constexpr bool NTS = (FL & 1) != 0;
    constexpr int PD = (FL & 4) ? 2 : 1;
    static_assert(NG <= MAX_GROUPS, "too many access groups");

    extern __shared__ __attribute__((aligned(16))) unsigned char ykh_smem[];
    T* ring = reinterpret_cast<T*>(ykh_smem);

    const BlockBox bb = block_box<VZ, C::TZ, C::TY, DESC>(a);
    const int tid = threadIdx.x;
    const int lz = tid % TZL, ly = tid / TZL;
    const int zt0 = bb.zt0, yt0 = bb.yt0, xs = bb.xs, xe = bb.xe;
    if (xs >= xe) return;
    const int myz = zt0 + lz * VZ;
    const int myy0 = yt0 + ly * RY;

    // own-point offsets within a plane, clamped into the allocation (as in march_kernel)
    int yc[RY];
    static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; yc[j] = clampi(myy0 + j, a.ay0, a.ay1 - 1); });
    const int zc = clampi(myz, a.az0, a.az1 - VZ);
    auto xclamp = [&](int x) { return clampi(x, a.ax0, a.ax1 - 1); };
    const idx_t org = (idx_t)a.ay0 * a.sy + a.az0;
    auto xplane = [&](int x) -> idx_t { return org + (idx_t)xclamp(x) * a.sx; };     // uniform
    auto plane_off = [&](int y, int z) -> unsigned {
        return (unsigned)((y - a.ay0) * (int)a.sy + (z - a.az0)) * (unsigned)sizeof(T);
    };
    unsigned ooff[RY];
    static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; ooff[j] = plane_off(yc[j], zc); });
    // the same for the kind-3 reads: clamped so that row + dy and vector + dz stay inside the allocation for every such read (points of
    // the launch's box are never moved by it: their reads lie in the halos, which are allocated)
    unsigned o3[RY];
    static_for<RY>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        o3[j] = plane_off(clampi(myy0 + j, a.ay0 + C::k3.yl, a.ay1 - 1 - C::k3.yh), clampi(myz, a.az0 + C::k3.zlv * VZ, a.az1 - VZ - C::k3.zhv * VZ));
    });

    // which vectors of a group's slab this thread brings in: vector h = tid + k * NT of the (rows x vectors) slab
    unsigned vofs[C::NVTOT > 0 ? C::NVTOT : 1];       // byte offset within the plane
    int vlds[C::NVTOT > 0 ? C::NVTOT : 1];            // element offset within the slab, -1: none
    int tofs[NG];
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        tofs[g] = 0;
        if constexpr (C::tab.kind[g] == 2) {
            constexpr int YL = C::tab.yl[g], ZLV = C::tab.zlv[g], LP = C::tab.lp[g], LPV = C::tab.lpv[g];
            constexpr int NV = C::tab.nv[g], NVT = C::tab.nvt[g], VO = C::tab.voff[g];
            tofs[g] = (YL + ly * RY) * LP + (ZLV + lz) * VZ;
            static_for<NVT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                int h = tid + k * NT;
                const bool live = h < NV;
                if (!live) h = 0;
                const int row = h / LPV, zv = h % LPV;
                const int y = clampi(yt0 - YL + row, a.ay0, a.ay1 - 1);
                const int z = clampi(zt0 - ZLV * VZ + zv * VZ, a.az0, a.az1 - VZ);
                vofs[VO + k] = plane_off(y, z);
                vlds[VO + k] = live ? row * LP + zv * VZ : -1;
            });
        }
    });

    VW nxt[NG];
    V hreg[PD][C::NVTOT > 0 ? C::NVTOT : 1];
    auto ld_own = [&](auto gc, int j, int x) -> V {
        constexpr int g = decltype(gc)::value;
        if constexpr (P::group_full[g]) {
            if constexpr (NTS) return ldv_b_nt<V>(sbase((const T*)a.ptr[g] + xplane(x)), ooff[j]);
            else return ldv_b<V>(sbase((const T*)a.ptr[g] + xplane(x)), ooff[j]);
        } else {
            const T* p = (const T*)a.ptr[g] + (idx_t)xclamp(x) * a.gsx[g] + (idx_t)yc[j] * a.gsy[g];
            if (a.gsz[g] == 0) return V(p[0]);
            return ldv<V>(p + zc);
        }
    };
    // request plane xp of every ring group (xp counted from the group's newest live plane: x + xhi + d) into register set S
    auto fetch = [&](int x, auto sc) {
        constexpr int S = decltype(sc)::value;
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (C::tab.kind[g] == 2) {
                constexpr int NVT = C::tab.nvt[g], VO = C::tab.voff[g], XHI = C::tab.xlo[g] + C::tab.nx[g] - 1;
                auto p = sbase((const T*)a.ptr[g] + xplane(x + XHI));
                static_for<NVT>([&](auto kc) { constexpr int k = decltype(kc)::value; hreg[S][VO + k] = ldv_b<V>(p, vofs[VO + k]); });
            }
        });
    };
    auto fetch_once = [&](int x) {
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (C::tab.kind[g] == 1)
                static_for<RY>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const V v = ld_own(gc, j, x);
                    static_for<VZ>([&](auto ec) { constexpr int e = decltype(ec)::value; nxt[g][j * VZ + e] = v[e]; });
                });
        });
    };

    // prologue: planes xs+xlo .. xs+xhi of every ring group go to slots 0 .. nx-1
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (C::tab.kind[g] == 2) {
            constexpr int NVT = C::tab.nvt[g], VO = C::tab.voff[g], NX = C::tab.nx[g], XLO = C::tab.xlo[g], RO = C::tab.roff[g], PL = C::tab.plane[g];
            static_for<NX>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                auto p = sbase((const T*)a.ptr[g] + xplane(xs + XLO + i));
                V tmp[NVT];
                static_for<NVT>([&](auto kc) { constexpr int k = decltype(kc)::value; tmp[k] = ldv_b<V>(p, vofs[VO + k]); });
                static_for<NVT>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if (vlds[VO + k] >= 0) stv<V>(ring + RO + i * PL + vlds[VO + k], tmp[k]);
                });
            });
        }
    });
    fetch_once(xs);
    // (PD = 2: the plane that iteration xs stores was requested "one iteration earlier")
    if constexpr (PD == 2) fetch(xs + 1, std::integral_constant<int, 0>{});
    __syncthreads();

    int c[NG];       // (x - xs) % nr of each ring group (uniform)
    static_for<NG>([&](auto gc) { c[decltype(gc)::value] = 0; });
    // One centre plane; register set S holds (or receives, PD = 1) the plane this iteration stores, x + xhi + 1
    auto plane = [&](int x, auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (PD == 1) fetch(x + 1, sc);
        else fetch(x + 2, std::integral_constant<int, 1 - S>{});
        // slot bases of the live planes
        int sl[C::NSTOT > 0 ? C::NSTOT : 1];
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (C::tab.kind[g] == 2) {
                constexpr int NX = C::tab.nx[g], NR = C::tab.nr[g], SO = C::tab.soff[g], RO = C::tab.roff[g], PL = C::tab.plane[g];
                static_for<NX>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    int s = c[g] + i;
                    if (s >= NR) s -= NR;
                    sl[SO + i] = RO + s * PL;
                });
            }
        });
        VW out[MAX_GROUPS];
        VW cur[NG];
        static_for<NG>([&](auto gc) { constexpr int g = decltype(gc)::value; if constexpr (C::tab.kind[g] == 1) cur[g] = nxt[g]; });
        if (x + 1 < xe) fetch_once(x + 1);
        {
            BoxAcc<C, P> acc{a, ring, sl, tofs, cur, x, myy0, myz, out, o3};
            P::eval(acc);
        }
        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int myy = myy0 + j;
            if (myy < a.y1 && myz < a.z1 && myz + VZ > a.z0) {
                const idx_t xo = org + (idx_t)x * a.sx;
                const bool whole = myz >= a.z0 && myz + VZ <= a.z1;
                static_for<P::n_writes>([&](auto wc) {
                    constexpr int g = P::writes[decltype(wc)::value];
                    auto ob = sbase((T*)a.ptr[g] + xo);
                    V o;
                    static_for<VZ>([&](auto ec) { constexpr int e = decltype(ec)::value; o[e] = out[g][j * VZ + e]; });
                    if (whole) {
                        if constexpr (NTS) stv_b_nt<V>(ob, ooff[j], o); else stv_b<V>(ob, ooff[j], o);
                    } else
                        static_for<VZ>([&](auto ec) {
                            constexpr int e = decltype(ec)::value;
                            if (myz + e >= a.z0 && myz + e < a.z1) stv_b<T>(ob, ooff[j] + e * (unsigned)sizeof(T), o[e]);
                        });
                });
            }
        });
        // the new plane goes into the free slot of each ring: (c + nx) % nr
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (C::tab.kind[g] == 2) {
                constexpr int NVT = C::tab.nvt[g], VO = C::tab.voff[g], NX = C::tab.nx[g], NR = C::tab.nr[g], RO = C::tab.roff[g], PL = C::tab.plane[g];
                int s = c[g] + NX;
                if (s >= NR) s -= NR;
                T* dst = ring + RO + s * PL;
                static_for<NVT>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if (vlds[VO + k] >= 0) stv<V>(dst + vlds[VO + k], hreg[S][VO + k]);
                });
                c[g] = c[g] + 1 == NR ? 0 : c[g] + 1;
            }
        });
        __syncthreads();
    };
    // (a trip of PD planes may run one plane past xe-1: loads are clamped, its stores fall under the x < xe test below)
    for (int x = xs; x < xe; x += PD)
        static_for<PD>([&](auto sc) { if (x + decltype(sc)::value < xe) plane(x + decltype(sc)::value, sc); });
    if constexpr (DESC) block_done(a, bb.flags);
}

}  // namespace ykh
