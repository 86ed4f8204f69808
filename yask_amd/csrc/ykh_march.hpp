// ykh_march.hpp -- generic 2.5-D marching kernel for multi-var / multi-equation parts (ssg).
//
// Generalises star25d (ykh_device.hpp) from one star group to every access group of a part.  The
// generated part is analysed at compile time, per (var, step) group g:
//   * x-neighbours   -> a register queue of the planes x+xlo_g .. x+xhi_g at the thread's own point
//                       (depth 1 for centre-only operands: a one-plane prefetch register);
//   * y/z-neighbours -> an LDS slab of the centre plane of g with exactly g's halo (y-only and z-only
//                       groups get row/column halos only), double buffered, one barrier per plane;
//                       the slab interior is written from the queue's centre entry, the halo from
//                       per-thread prefetch registers;
//   * mixed offsets  -> (rare: ssg's `mu`) one (possibly unaligned) global vector load per read, requested with
//                       the rest of the plane's prefetch, i.e. PD planes before it is used.
// A workgroup owns a (y,z) tile and marches along x, so x re-use lives in registers and never depends
// on cache capacity: the point kernels re-fetch every x-neighbour plane over the fabric
// (ssg stage 1: ~124 B/point moved for 52 B/point algorithmic), this kernel moves each plane once
// (+ tile halos).  All loads are 16-byte vectors along z, a wave covers whole tile rows.
// Replaces the reference's generated calc_vectors loop + block loops for such parts
// (src/compiler/lib/YaskKernel.cpp:591-719, src/kernel/lib/stencil_calc.cpp:40-289).
#pragma once
#include "ykh_device.hpp"
#include "ykh_starlin.hpp"   // vecn, zshiftn, ldv/stv, nt variants

namespace ykh {

constexpr int MAX_MIXED = 8;      // more mixed-offset reads than this (e.g. the 5x5x5 `cube`) are loaded where they are used

struct GroupShape {
    int xlo, xhi, ylo, yhi, zlo, zhi;   // ranges over axis-aligned reads
    bool any, mixed, written;
};

template <class P>
constexpr GroupShape group_shape(int g) {
    GroupShape s = {0, 0, 0, 0, 0, 0, false, false, false};
    for (int i = 0; i < P::n_reads; i++) {
        if (P::reads[i].g != g) continue;
        s.any = true;
        int dx = P::reads[i].dx, dy = P::reads[i].dy, dz = P::reads[i].dz;
        if ((dx != 0) + (dy != 0) + (dz != 0) > 1) { s.mixed = true; continue; }
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

// A part can use the marching kernel if every group that is read at an offset (queue or slab) and every
// written group is a var over all domain dims (shared strides/pads); centre-only operands may be partial.
template <class P>
constexpr bool march_eligible() {
    for (int g = 0; g < P::n_groups; g++) {
        GroupShape s = group_shape<P>(g);
        bool offs = s.xlo || s.xhi || s.ylo || s.yhi || s.zlo || s.zhi || s.mixed;
        if ((offs || s.written) && !P::group_full[g]) return false;
    }
    return true;
}

// Per-group layout tables, computed once per instantiation at compile time (always used through
// constexpr values: a call in a runtime context would put the analysis loops into the kernel).
struct MarchTab {
    int nq[MAX_GROUPS], qoff[MAX_GROUPS + 1], xlo[MAX_GROUPS];
    bool slab[MAX_GROUPS];
    int yl[MAX_GROUPS], yh[MAX_GROUPS], zlv[MAX_GROUPS], zhv[MAX_GROUPS], lp[MAX_GROUPS], lrows[MAX_GROUPS];
    int soff[MAX_GROUPS + 1];
    int nhy[MAX_GROUPS], nh[MAX_GROUPS], nht[MAX_GROUPS], hoff[MAX_GROUPS + 1];
    bool ring[MAX_GROUPS];              // halo ring (HR): the halo is loaded with the interior, xhi planes early
    int rdepth[MAX_GROUPS], roff[MAX_GROUPS + 1];
    int nmix, mix[MAX_MIXED][4];        // distinct mixed-offset reads (g, dx, dy, dz)
    bool in_mix[MAX_GROUPS];            // the group is read at a mixed offset
};

template <class P, int VZ_, int TZL_, int TYL_, int RY_ = 1, bool HR_ = false>
struct MarchCfg {
    typedef typename P::real_t T;
    static constexpr int VZ = VZ_, TZL = TZL_, TYL = TYL_, RY = RY_, NT = TZL_ * TYL_, NG = P::n_groups;
    static constexpr int TZ = TZL * VZ, TY = TYL * RY;      // RY rows per thread
    static constexpr MarchTab make() {
        MarchTab t = {};
        int qo = 0, so = 0, ho = 0, ro = 0;
        for (int g = 0; g < NG; g++) {
            GroupShape s = group_shape<P>(g);
            t.xlo[g] = s.xlo;
            t.nq[g] = s.any ? s.xhi - s.xlo + 1 : 0;       // queue: planes x+xlo .. x+xhi
            t.qoff[g] = qo; qo += t.nq[g];
            t.slab[g] = s.any && (s.ylo || s.yhi || s.zlo || s.zhi);
            t.yl[g] = -s.ylo; t.yh[g] = s.yhi;
            t.zlv[g] = (-s.zlo + VZ - 1) / VZ; t.zhv[g] = (s.zhi + VZ - 1) / VZ;
            t.lp[g] = TZ + (t.zlv[g] + t.zhv[g]) * VZ;
            t.lrows[g] = TY + t.yl[g] + t.yh[g];
            t.soff[g] = so; so += t.slab[g] ? t.lrows[g] * t.lp[g] : 0;
            t.nhy[g] = (t.yl[g] + t.yh[g]) * TZL;
            t.nh[g] = t.slab[g] ? t.nhy[g] + TY * (t.zlv[g] + t.zhv[g]) : 0;
            t.nht[g] = (t.nh[g] + NT - 1) / NT;
            t.hoff[g] = ho; ho += t.nht[g];
            t.ring[g] = HR_ && t.slab[g] && s.xhi > 0;
            t.rdepth[g] = s.xhi + 1;
            t.roff[g] = ro; ro += t.ring[g] ? t.rdepth[g] * t.nh[g] * VZ : 0;
        }
        t.qoff[NG] = qo; t.soff[NG] = so; t.hoff[NG] = ho; t.roff[NG] = ro;
        for (int i = 0; i < P::n_reads; i++)
            if ((P::reads[i].dx != 0) + (P::reads[i].dy != 0) + (P::reads[i].dz != 0) > 1) t.in_mix[P::reads[i].g] = true;
        for (int i = 0; i < P::n_reads; i++) {
            int g = P::reads[i].g, dx = P::reads[i].dx, dy = P::reads[i].dy, dz = P::reads[i].dz;
            if ((dx != 0) + (dy != 0) + (dz != 0) < 2) continue;
            bool seen = false;
            for (int k = 0; k < t.nmix; k++)
                if (t.mix[k][0] == g && t.mix[k][1] == dx && t.mix[k][2] == dy && t.mix[k][3] == dz) seen = true;
            if (seen) continue;
            if (t.nmix == MAX_MIXED) { t.nmix = 0; break; }      // too many to hold in registers: no prefetch
            t.mix[t.nmix][0] = g; t.mix[t.nmix][1] = dx; t.mix[t.nmix][2] = dy; t.mix[t.nmix][3] = dz;
            t.nmix++;
        }
        return t;
    }
    static constexpr int mix_index(int g, int dx, int dy, int dz) {
        for (int k = 0; k < tab.nmix; k++)
            if (tab.mix[k][0] == g && tab.mix[k][1] == dx && tab.mix[k][2] == dy && tab.mix[k][3] == dz) return k;
        return -1;
    }
    static constexpr MarchTab tab = make();
    static constexpr int NQTOT = tab.qoff[NG];
    static constexpr int SLAB_TOT = tab.soff[NG];           // elements of one buffer set
    static constexpr int NHTOT = tab.hoff[NG];
    static constexpr int NMIX = tab.nmix;
    static constexpr int RING_TOT = tab.roff[NG];           // elements of the halo rings
    static constexpr size_t lds_bytes = sizeof(T) * (2 * (SLAB_TOT > 0 ? SLAB_TOT : 1) + RING_TOT);
    // what a block pays before its first plane, in plane-iterations (the cost model of planned launches, ykh_plan.cpp): its
    // deepest queue is filled with own-point loads only -- about half the work of a plane per queue plane
    static constexpr int max_nq() { int m = 1; for (int g = 0; g < NG; g++) if (tab.nq[g] > m) m = tab.nq[g]; return m; }
    static constexpr int XOVER = (max_nq() + 1) / 2 + 1;
};

// centre-only operands of a part that live in no slab and no mixed read ("once" operands: read once, by one thread) -- what the
// late refill (_lo, FL & 128) holds once instead of twice
template <class P>
constexpr int march_once_count() {
    typedef MarchCfg<P, 4, 32, 16> C;
    int n = 0;
    for (int g = 0; g < C::NG; g++)
        if (C::tab.nq[g] == 1 && !C::tab.slab[g] && !C::tab.in_mix[g]) n++;
    return n;
}

// a var over a subset of the domain dims that lacks the marching dim (awp's delta_t, h, cr_y, cr_z; iso3dfd_sponge's y and z
// profiles): the same value at every plane -- loaded once per block into a register, never prefetched, queued or refilled.
// (Such a var is read at the centre only: march_eligible() wants full vars wherever there is an offset.)
template <class P>
constexpr bool march_x_invariant(int g) { return !P::group_full[g] && (GroupDims<P>::get(g) & 1) == 0; }

// PIN: honour the generated code's pin() after every temporary (strict program order: smallest live
// ranges, least instruction-level parallelism); otherwise only the equations are kept sequential.
// OPS: how the generated code's a - b and a / b are issued (fp32, VZ a multiple of 2):
//   1  a - b as fma(b, -1, a) with the -1 an opaque scalar: bit-identical (the product is exact), but ONE v_pk_fma_f32 per two
//      lanes where hipcc splits a v2f32 subtract into two v_sub_f32 -- ssg's staggered differences are 36 vector subtractions
//      per point and stage: 144 of the ~500-640 vector instructions per thread-plane;
//   2  a / b as a * rcp(b): v_rcp_f32 (1 ulp) + one packed multiply, <= 1.5 ulp of the quotient (2 / x: 1 ulp), IEEE special
//      cases kept (x / 0 = inf, x / inf = 0), against ~11 instructions per lane of the correctly rounded sequence (v_div_scale x2,
//      v_rcp, 4 fma, v_div_fmas, v_div_fixup) -- ssg has 3 + 5 divisions per point: a quarter to a third of its instructions.
//      Not bit-identical to the reference's vdivps; far inside the stated parity bound (2e-5), see DESIGN.md 3.3.
// PH: rotation phase of the register queues (trips of several planes rename the queue slots instead of moving them, see
// march_kernel): logical queue entry i of a group with n entries lives in slot (i + PH) % n.
// LO: centre-only operands that live in no slab and no mixed read ("once" operands: read once, by one thread) are read straight
// from their prefetch registers, which are refilled AFTER the plane has been evaluated (see march_kernel, FL & 128).
template <class C, class P, bool PIN = false, int OPS = 0, int PH = 0, bool LO = false>
struct MarchAcc {
    typedef typename C::T T;
    typedef typename vecn<T, C::VZ>::type V;
    static constexpr int VZ = C::VZ;
    static constexpr bool PK32 = sizeof(T) == 4 && (C::VZ % 2) == 0;
    const PartArgs& a;
    const V (&q)[C::NQTOT > 0 ? C::NQTOT : 1];     // queues of the row being evaluated
    const V (&mx)[C::NMIX > 0 ? C::NMIX : 1];      // mixed-offset reads of the row, prefetched
    const V (&nx)[C::NG];                          // prefetch registers of the row (LO: the "once" operands are read from here)
    const V (&inv)[C::NG];                         // operands without the marching dim (march_x_invariant): loaded once per block
    const T* sb;            // current slab buffer set
    int ly, lz;             // row (within the tile) and z lane of the point
    int x, y, z0;           // point (first of the VZ)
    V (&out)[MAX_GROUPS];
    T m1;                   // -1, opaque to the optimiser (OPS & 1)
    template <class L, class R>
    __device__ __forceinline__ V sub(L l, R r) const {
        if constexpr (PK32 && (OPS & 1)) return __builtin_elementwise_fma(V(r), V(m1), V(l));
        else return V(l) - V(r);
    }
    template <class L, class R>
    __device__ __forceinline__ V div(L l, R r) const {
        if constexpr (sizeof(T) == 4 && (OPS & 2)) {
            const V d = V(r);
            V rc;
            static_for<VZ>([&](auto ec) { constexpr int e = decltype(ec)::value; rc[e] = __builtin_amdgcn_rcpf(d[e]); });
            return V(l) * rc;
        } else return V(l) / V(r);
    }
    template <int G, int DX, int DY, int DZ>
    __device__ __forceinline__ V rd() const {
        constexpr int nz = (DX != 0) + (DY != 0) + (DZ != 0);
        if constexpr (nz > 1 && C::NMIX > 0) {
            constexpr int k = C::mix_index(G, DX, DY, DZ);
            static_assert(k >= 0, "mixed read missing from the table");
            return mx[k];
        } else if constexpr (nz > 1) {
            // many mixed offsets: aligned global loads at the point of use (L1/L2 served)
            const T* p = (const T*)a.ptr[G] + (idx_t)(x + DX) * a.gsx[G] + (idx_t)(y + DY) * a.gsy[G];
            constexpr int qq = (DZ >= 0) ? DZ / VZ : -((-DZ + VZ - 1) / VZ);
            constexpr int e = DZ - qq * VZ;
            const T* pz = p + z0 + qq * VZ;
            if constexpr (e == 0) return ldv<V>(pz);
            else return zshiftn<T, VZ, e>(ldv<V>(pz), ldv<V>(pz + VZ));
        } else if constexpr (DY == 0 && DZ == 0) {
            if constexpr (march_x_invariant<P>(G)) return inv[G];
            if constexpr (LO && C::tab.nq[G] == 1 && !C::tab.slab[G] && !C::tab.in_mix[G]) return nx[G];
            constexpr int qi = C::tab.qoff[G] + (DX - C::tab.xlo[G] + PH) % C::tab.nq[G];
            return q[qi];
        } else {
            constexpr int so = C::tab.soff[G], yl = C::tab.yl[G], lp = C::tab.lp[G], zlv = C::tab.zlv[G];
            const T* row = sb + so + (yl + ly + DY) * lp + (zlv + lz) * VZ;
            if constexpr (DZ == 0) return ldv<V>(row);
            else {
                constexpr int qq = (DZ >= 0) ? DZ / VZ : -((-DZ + VZ - 1) / VZ);
                constexpr int e = DZ - qq * VZ;
                if constexpr (e == 0) return ldv<V>(row + qq * VZ);
                else return zshiftn<T, VZ, e>(ldv<V>(row + qq * VZ), ldv<V>(row + (qq + 1) * VZ));
            }
        }
    }
    // The empty asm keeps the equations of a part sequential: without it hipcc hoists the LDS/global
    // reads of all equations to the top of eval() and the live values exceed the register file.
    template <int G>
    __device__ __forceinline__ void wr(V v) { out[G] = v; asm volatile("" : "+v"(out[G]) : : "memory"); }
    __device__ __forceinline__ void pin(V& v) const { if constexpr (PIN) asm volatile("" : "+v"(v) : : "memory"); }
    template <int D>
    __device__ __forceinline__ V idx() const {
        if constexpr (D == 2) { V r; static_for<VZ>([&](auto ec) { constexpr int e = decltype(ec)::value; r[e] = T(z0 + e + a.ofs_z); }); return r; }
        else return V(T(D == 0 ? x + a.ofs_x : y + a.ofs_y));
    }
    __device__ __forceinline__ V step() const { return V(T(a.t)); }
};

// PD: planes prefetched ahead (1..3; PD alternating register sets, PD times the bytes in flight).
// NTS: non-temporal loads of the centre-only operands (read once, by one thread) and non-temporal stores, so that
// these streams do not push the halo lines -- which a neighbouring tile is about to read -- out of L2.
// FL & 2 (HR, halo ring): a group with a queue reaching xhi planes ahead loads its own points at plane x+xhi but -- without
// HR -- the halo of its slab only at plane x, xhi planes after the neighbouring tiles streamed the same lines as their
// interior (ssg: 4 planes x ~4.7 MB of plane data per XCD against 4 MiB of L2: the halo lines come over the fabric a second
// time).  With HR the halo of plane x+xhi is requested together with the interior and parked in an LDS ring of xhi+1
// planes until the plane becomes the centre; every ring entry is written and read back by the same thread (no barrier).
// DESC: the twin that takes its tile and x range from a block descriptor (planned launches, ykh_plan.cpp) and signals when done.
template <class P, int VZ, int TZL, int TYL, int MINW, int RY = 1, bool PIN = false, int PD = 1, int FL = 0, bool DESC = false>
__global__ void __launch_bounds__(TZL* TYL, MINW) march_kernel(const PartArgs a) {
    constexpr int NTS = FL & 1;
    constexpr bool HR = (FL & 2) != 0;
    constexpr int OPS = (FL >> 2) & 3;        // FL & 4: packed subtractions, FL & 8: reciprocal-based divisions (MarchAcc)
    // FL & 16 / 32 / 64: trips of 2 / 4 / 8 planes in which the queue slots are RENAMED (plane p of a trip reads logical entry i
    // of an n-deep queue from slot (i + p) % n) and rotated by the trip length once per trip -- n moves per trip instead of
    // n - 1 per plane; with the trip a multiple of n none at all.  ssg: the moves were 91 of the ~320 vector instructions per
    // thread-plane left after _ps and _fd.
    // FL & 128 (_lo): the "once" operands (centre-only, no slab, no mixed read: ssg stage 2 has nine) are normally prefetched one
    // plane ahead like everything else and copied into their queue slot when the plane starts -- so each is held twice while the
    // plane is evaluated (36 VGPRs in ssg stage 2, which sits at the 256 limit).  With _lo the evaluation reads the prefetch register
    // itself and the register is refilled right after the plane's evaluation: one copy, at the price of a shorter prefetch
    // distance (from the end of plane x to the operand's use in plane x+1).
    constexpr bool LO = (FL & 128) != 0;
    // FL bits 9-11 ("_ls<K>", round 5): soft lock-step of the XCD's workgroups every K = 1 << (code - 1) planes (ykh_device.hpp xcd_lockstep)
    constexpr int LS_K = ((FL >> 9) & 7) == 0 ? 0 : (1 << (((FL >> 9) & 7) - 1));
    constexpr int KT = (FL & 64) ? 8 : ((FL & 32) ? 4 : ((FL & 16) ? 2 : 1));
    static_assert(KT == 1 || KT % PD == 0, "the trip must be a multiple of the prefetch depth");
    typedef MarchCfg<P, VZ, TZL, TYL, RY, HR> C;
    typedef typename C::T T;
    typedef typename vecn<T, VZ>::type V;
    constexpr int NG = C::NG, NT = C::NT;
    static_assert(NG <= MAX_GROUPS, "too many access groups");

    extern __shared__ __attribute__((aligned(16))) unsigned char ykh_smem[];
    T* slab = reinterpret_cast<T*>(ykh_smem);
    T* ring = slab + 2 * (C::SLAB_TOT > 0 ? C::SLAB_TOT : 1);

    const BlockBox bb = block_box<VZ, C::TZ, C::TY, DESC>(a);      // regular tiling of the box, or a planned launch's descriptor
    const int tid = threadIdx.x;
    const int lz = tid % TZL, ly = tid / TZL;
    const int zt0 = bb.zt0, yt0 = bb.yt0, xs = bb.xs, xe = bb.xe;
    if (xs >= xe) return;
    const int myz = zt0 + lz * VZ;
    const int myy0 = yt0 + ly * RY;       // first of this thread's RY rows

    // Own-point offsets within a plane, clamped into the allocation.  Vars over all domain dims share strides
    // and pads, so one 32-bit byte offset from the plane's first allocated element serves every such group and
    // the accesses take the scalar-base + 32-bit-offset form (no per-group strides in SGPRs, no 64-bit vector
    // adds); vars over a subset of the dims (centre-only operands) go through their own strides.
    int yc[RY];
    static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; yc[j] = clampi(myy0 + j, a.ay0, a.ay1 - 1); });
    const int zc = clampi(myz, a.az0, a.az1 - VZ);
    auto xclamp = [&](int x) { return clampi(x, a.ax0, a.ax1 - 1); };
    const idx_t org = (idx_t)a.ay0 * a.sy + a.az0;
    auto xplane = [&](int x) -> idx_t { return org + (idx_t)xclamp(x) * a.sx; };     // uniform
    auto plane_off = [&](int y, int z) -> unsigned {                                   // y, z inside the allocation
        return (unsigned)((y - a.ay0) * (int)a.sy + (z - a.az0)) * (unsigned)sizeof(T);
    };
    unsigned ooff[RY];
    static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; ooff[j] = plane_off(yc[j], zc); });

    // halo assignments: for group g, halo vector h = tid + k*NT
    unsigned hofs[C::NHTOT > 0 ? C::NHTOT : 1];       // byte offset within the plane
    int hlds[C::NHTOT > 0 ? C::NHTOT : 1];
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr bool slabg = C::tab.slab[g];
        if constexpr (slabg) {
            constexpr int YL = C::tab.yl[g], ZLV = C::tab.zlv[g], ZHV = C::tab.zhv[g], LP = C::tab.lp[g];
            constexpr int NHY = C::tab.nhy[g], NH = C::tab.nh[g], NHT = C::tab.nht[g], HO = C::tab.hoff[g], SO = C::tab.soff[g];
            constexpr int ZV2 = (ZLV + ZHV > 0) ? ZLV + ZHV : 1;
            static_for<NHT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                int h = tid + k * NT;
                int row, zv;
                if (h < NHY) {
                    int r = h / TZL;
                    row = r < YL ? r : r + C::TY;
                    zv = ZLV + h % TZL;
                } else {
                    int hh = h - NHY;
                    int r = hh / ZV2, c = hh % ZV2;
                    row = YL + r;
                    zv = c < ZLV ? c : c + TZL;
                }
                if (h >= NH) { row = 0; zv = 0; }
                int y = clampi(yt0 - YL + row, a.ay0, a.ay1 - 1);
                int z = clampi(zt0 - ZLV * VZ + zv * VZ, a.az0, a.az1 - VZ);
                hofs[HO + k] = plane_off(y, z);
                hlds[HO + k] = (h < NH) ? SO + row * LP + zv * VZ : -1;
            });
        }
    });
    // mixed-offset reads: the (y+dy, z+dz) part of each read's address, per row
    unsigned moff[RY][C::NMIX > 0 ? C::NMIX : 1];
    static_for<C::NMIX>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int DY = C::tab.mix[k][2], DZ = C::tab.mix[k][3];
        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            moff[j][k] = plane_off(clampi(myy0 + j + DY, a.ay0, a.ay1 - 1), clampi(myz + DZ, a.az0, a.az1 - VZ));
        });
    });

    T m1 = T(-1);
    if constexpr (OPS & 1) asm volatile("" : "+s"(m1));
    V q[RY][C::NQTOT > 0 ? C::NQTOT : 1];
    V nxt[PD][RY][NG];
    V hreg[PD][C::NHTOT > 0 ? C::NHTOT : 1];
    V mreg[PD][RY][C::NMIX > 0 ? C::NMIX : 1];

    // load the own-point vector of group g, row j, at plane x (vars without z: broadcast)
    auto ld_own = [&](auto gc, int j, int x) -> V {
        constexpr int g = decltype(gc)::value;
        if constexpr (P::group_full[g]) {
            constexpr bool once = C::tab.nq[g] == 1 && !C::tab.slab[g] && !C::tab.in_mix[g];
            if constexpr (NTS && once) return ldv_b_nt<V>(sbase((const T*)a.ptr[g] + xplane(x)), ooff[j]);
            else return ldv_b<V>(sbase((const T*)a.ptr[g] + xplane(x)), ooff[j]);
        }
        else {
            constexpr unsigned gd = GroupDims<P>::get(g);
            if constexpr (gd == 7) {         // (no compile-time table: the run-time strides say which dims the var has)
                const T* p = (const T*)a.ptr[g] + (idx_t)xclamp(x) * a.gsx[g] + (idx_t)yc[j] * a.gsy[g];
                if (a.gsz[g] == 0) return V(p[0]);
                return ldv<V>(p + zc);
            } else {
                // the dims the var has, known here: no stride multiplies for the others, ONE value where z is missing (a splat the
                // compiler keeps in one register), a uniform address -- a scalar load -- where y and z are
                const T* p = (const T*)a.ptr[g];
                if constexpr ((gd & 1) != 0) p += (idx_t)xclamp(x) * a.gsx[g];
                if constexpr ((gd & 2) != 0) p += (idx_t)yc[j] * a.gsy[g];
                if constexpr ((gd & 4) == 0) return V(p[0]);
                else return ldv<V>(p + zc);
            }
        }
    };
    auto prefetch = [&](int x, auto sc) {      // everything needed to advance to centre plane x, into register set S
        constexpr int S = decltype(sc)::value;
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int NQ = C::tab.nq[g], XHI = C::tab.xlo[g] + C::tab.nq[g] - 1;
            constexpr bool slabg = C::tab.slab[g];
            constexpr bool once = NQ == 1 && !slabg && !C::tab.in_mix[g];
            if constexpr (NQ > 0 && !(LO && once) && !march_x_invariant<P>(g))
                static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; nxt[S][j][g] = ld_own(gc, j, x + XHI); });
            if constexpr (slabg) {
                constexpr int NHT = C::tab.nht[g], HO = C::tab.hoff[g];
                constexpr int HX = C::tab.ring[g] ? XHI : 0;      // ring: the halo travels with the interior
                auto p = sbase((const T*)a.ptr[g] + xplane(x + HX));
                static_for<NHT>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    hreg[S][HO + k] = ldv_b<V>(p, hofs[HO + k]);
                });
            }
        });
    };
    // the mixed-offset reads of centre plane x (their registers are read by eval(), so plane x's set is refilled
    // only after plane x has been evaluated)
    auto prefetch_mixed = [&](int x, auto sc) {
        constexpr int S = decltype(sc)::value;
        static_for<C::NMIX>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            constexpr int g = C::tab.mix[k][0], DX = C::tab.mix[k][1];
            auto p = sbase((const T*)a.ptr[g] + xplane(x + DX));
            static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; mreg[S][j][k] = ldv_u<V>(p, moff[j][k]); });
        });
    };

    // the "once" operands of centre plane x (LO only; like the mixed reads, refilled after plane x has been evaluated)
    auto prefetch_once = [&](int x, auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (LO)
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr bool once = C::tab.nq[g] == 1 && !C::tab.slab[g] && !C::tab.in_mix[g];
                if constexpr (once && !march_x_invariant<P>(g))
                    static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; nxt[S][j][g] = ld_own(gc, j, x + C::tab.xlo[g]); });
            });
    };

    // operands without the marching dim: once per block
    V inv[RY][NG];
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (march_x_invariant<P>(g) && C::tab.nq[g] > 0)
            static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; inv[j][g] = ld_own(gc, j, xs); });
    });
    // prologue: queues hold planes xs+xlo .. xs+xhi-1; newest plane + halos of plane xs prefetched
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr int NQ = C::tab.nq[g], QO = C::tab.qoff[g], XLO = C::tab.xlo[g];
        static_for<(NQ > 0 ? NQ - 1 : 0)>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; q[j][QO + i] = ld_own(gc, j, xs + XLO + i); });
        });
    });
    // halo rings: planes xs .. xs+xhi-1 (the later ones arrive through prefetch())
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (C::tab.ring[g]) {
            constexpr int NHT = C::tab.nht[g], HO = C::tab.hoff[g], NH = C::tab.nh[g], RO = C::tab.roff[g], RD = C::tab.rdepth[g];
            static_for<RD - 1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                auto p = sbase((const T*)a.ptr[g] + xplane(xs + i));
                static_for<NHT>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if (hlds[HO + k] >= 0) stv<V>(ring + RO + (i * NH + tid + k * NT) * VZ, ldv_b<V>(p, hofs[HO + k]));
                });
            });
        }
    });
    static_for<PD>([&](auto sc) { prefetch(xs + decltype(sc)::value, sc); prefetch_mixed(xs + decltype(sc)::value, sc); prefetch_once(xs + decltype(sc)::value, sc); });

    // One centre plane; `sc` = register set holding its prefetched data (the plane's position in the trip).
    auto plane = [&](int x, auto sc, auto phc) {
        constexpr int S = decltype(sc)::value;
        constexpr int PH = decltype(phc)::value;
        T* sb = slab + (x & 1) * C::SLAB_TOT;
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int NQ = C::tab.nq[g], QO = C::tab.qoff[g], XLO = C::tab.xlo[g];
            constexpr bool slabg = C::tab.slab[g];
            if constexpr (NQ > 0 && !(LO && NQ == 1 && !slabg && !C::tab.in_mix[g]) && !march_x_invariant<P>(g))
                static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; q[j][QO + (NQ - 1 + PH) % NQ] = nxt[S][j][g]; });
            if constexpr (slabg) {
                constexpr int YL = C::tab.yl[g], ZLV = C::tab.zlv[g], LP = C::tab.lp[g], SO = C::tab.soff[g];
                constexpr int NHT = C::tab.nht[g], HO = C::tab.hoff[g];
                static_for<RY>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    stv<V>(sb + SO + (YL + ly * RY + j) * LP + (ZLV + lz) * VZ, q[j][QO + (-XLO + PH) % NQ]);
                });
                if constexpr (C::tab.ring[g]) {
                    // hreg holds the halo of plane x+xhi: park it, and fetch plane x's from where this thread parked it
                    constexpr int NH = C::tab.nh[g], RO = C::tab.roff[g], RD = C::tab.rdepth[g];
                    const int sr = (int)((unsigned)(x - xs) % RD), sw = (int)((unsigned)(x - xs + RD - 1) % RD);
                    static_for<NHT>([&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        if (hlds[HO + k] >= 0) {
                            T* rp = ring + RO + (tid + k * NT) * VZ;
                            V cur = ldv<V>(rp + sr * (NH * VZ));
                            stv<V>(rp + sw * (NH * VZ), hreg[S][HO + k]);
                            stv<V>(sb + hlds[HO + k], cur);
                        }
                    });
                } else
                static_for<NHT>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    if (hlds[HO + k] >= 0) stv<V>(sb + hlds[HO + k], hreg[S][HO + k]);
                });
            }
        });
        prefetch(x + PD, sc);            // loads are clamped into the allocation
        __syncthreads();

        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int myy = myy0 + j;
            V out[MAX_GROUPS];
            MarchAcc<C, P, PIN, OPS, PH, LO> acc{a, q[j], mreg[S][j], nxt[S][j], inv[j], sb, ly * RY + j, lz, x, myy, myz, out, m1};
            P::eval(acc);
            if (x < xe && myy < a.y1 && myz < a.z1 && myz + VZ > a.z0) {
                // (written groups are vars over all dims; the store predicate implies yc[j] == myy, zc == myz)
                const idx_t xo = org + (idx_t)x * a.sx;
                const bool whole = myz >= a.z0 && myz + VZ <= a.z1;
                static_for<P::n_writes>([&](auto wc) {
                    constexpr int g = P::writes[decltype(wc)::value];
                    auto ob = sbase((T*)a.ptr[g] + xo);
                    if (whole) {
                        if constexpr ((FL & 256) != 0 && sizeof(V) == 16) stv_b_wt<V>(ob, ooff[j], out[g]);      // _wt: write-through, ykh_starlin.hpp
                        else if constexpr (NTS) stv_b_nt<V>(ob, ooff[j], out[g]); else stv_b<V>(ob, ooff[j], out[g]);
                    }
                    else
                        static_for<VZ>([&](auto ec) {
                            constexpr int e = decltype(ec)::value;
                            if (myz + e >= a.z0 && myz + e < a.z1) stv_b<T>(ob, ooff[j] + e * (unsigned)sizeof(T), out[g][e]);
                        });
                });
            }
        });
        prefetch_mixed(x + PD, sc);
        prefetch_once(x + PD, sc);
        // rotate the queues (per plane; trips of KT > 1 planes rename instead and rotate once, below)
        if constexpr (KT == 1)
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int NQ = C::tab.nq[g], QO = C::tab.qoff[g];
                static_for<(NQ > 0 ? NQ - 1 : 0)>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; q[j][QO + i] = q[j][QO + i + 1]; });
                });
            });
    };
    typedef std::integral_constant<int, 0> I0;
    [[maybe_unused]] bool ls_dead = false;
    if constexpr (KT == 1) {
        // PD planes per trip; a trip may run past xe-1 (stores are predicated, loads clamped)
        for (int x = xs; x < xe; x += PD) {
            if constexpr (LS_K > 0 && !DESC) xcd_lockstep<LS_K>(a.sig, x - xs, PD, ls_dead);
            static_for<PD>([&](auto sc) { plane(x + decltype(sc)::value, sc, I0{}); });
        }
    } else {
        for (int x = xs; x < xe; x += KT) {
            if constexpr (LS_K > 0 && !DESC) xcd_lockstep<LS_K>(a.sig, x - xs, KT, ls_dead);
            static_for<KT>([&](auto pc) {
                constexpr int p = decltype(pc)::value;
                plane(x + p, std::integral_constant<int, p % PD>{}, pc);
            });
            // the queues go back to phase 0: slot i takes what slot (i + KT) % n holds (nothing to do where n divides KT)
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int NQ = C::tab.nq[g], QO = C::tab.qoff[g];
                if constexpr (NQ > 1 && (KT % NQ) != 0)
                    static_for<RY>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        V tq[NQ];
                        static_for<NQ>([&](auto ic) { constexpr int i = decltype(ic)::value; tq[i] = q[j][QO + i]; });
                        static_for<NQ>([&](auto ic) { constexpr int i = decltype(ic)::value; q[j][QO + i] = tq[(i + KT) % NQ]; });
                    });
            });
        }
    }
    if constexpr (DESC) block_done(a, bb.flags);
}

}  // namespace ykh
