// ykh_starlin.hpp -- second-generation 2.5-D kernel for parts in *linear star form*.
//
// A part is in linear star form when its single equation can be written as
//     out = F( centre-only operands , L ),   L = sum_i c_i * S(x+dx_i, y+dy_i, z+dz_i)
// with S one (var, step) access group ("star group"), every offset axis-aligned and every c_i a
// compile-time constant.  iso3dfd (src/stencils/Iso3dfdStencil.cpp:63-137: L = the 49-point
// Laplacian of p(t), F = 2p(t) - p(t-1) + v*L) and the AxisStencil family (`3axis`,
// src/stencils/SimpleStencils.cpp:61-103) have it; the `cdna4_hip` compiler target detects the form
// and emits `lin[]` + `eval_lin()` next to the general `eval()`.
//
// Why a second kernel shape.  star25d (ykh_device.hpp) gathers all x-neighbours from a register queue
// of XL+XH+1 planes, so a plane's (y,z) halo is fetched XH planes *after* the neighbouring tiles
// streamed the same lines as their interior; at r=8 that reuse distance (8 planes x 3 streams x
// ~0.5 MB per XCD) exceeds the 4 MiB L2 and every halo line is fetched twice over the fabric
// (rocprof: FETCH 1.39x the algorithmic read bytes, profiles/r01a_star25d_v1).  Here the x direction
// is split into a *gathered past* and a *scattered future*:
//     when plane X arrives (interior + halos, all loaded at the same time by all tiles)
//       acc[X]    = sum_{dx<=0} c(dx)*S(X+dx) [register queue of the past XL planes]
//                 + sum_{dy,dz} c*S(X, y+dy, z+dz)   [LDS slab of plane X]
//       acc[X-k] += c(+k) * S(X)   for k = 1..XH      [partial sums held in registers]
//       out[X-XH] = F(operands(X-XH), acc[X-XH])      [complete now]
// The register budget is the same as star25d's (XL+1 queue planes + XH+1 partial sums vs XL+XH+1
// planes) but each plane of S is touched once, at one time, by every tile that needs it -> halo lines
// hit in L2 while the neighbour's interior load is still resident.  Centre-only operands and the
// output are streamed with non-temporal hints so that they do not evict S from L2.
// The summation order differs from the reference's expression order (rounding-level differences only;
// parity tolerance stated in tests/).
#pragma once
#include "ykh_device.hpp"

namespace ykh {

template <class P>
constexpr double lin_coef(int dx, int dy, int dz) {
    double s = 0;
    for (int i = 0; i < P::n_lin; i++)
        if (P::lin[i].dx == dx && P::lin[i].dy == dy && P::lin[i].dz == dz) s += P::lin[i].c;
    return s;
}

template <class P>
constexpr StarRange lin_range() {
    StarRange r = {0, 0, 0, 0, 0, 0, false, P::n_lin > 0, false};
    for (int i = 0; i < P::n_lin; i++) {
        int dx = P::lin[i].dx, dy = P::lin[i].dy, dz = P::lin[i].dz;
        int nz = (dx != 0) + (dy != 0) + (dz != 0);
        if (nz == 0) r.center = true;
        if (nz > 1) { r.mixed = true; continue; }
        if (dx < r.xlo) r.xlo = dx;
        if (dx > r.xhi) r.xhi = dx;
        if (dy < r.ylo) r.ylo = dy;
        if (dy > r.yhi) r.yhi = dy;
        if (dz < r.zlo) r.zlo = dz;
        if (dz > r.zhi) r.zhi = dz;
    }
    return r;
}

// Distinct coefficient values of the linear part.  The kernel keeps them as OPAQUE scalars (one SGPR, or pair, each): with
// compile-time literals hipcc canonicalises `s += x * -C` into `s = fma(-x, C, s)` so that C and -C share a register, and
// then fails to fold half of those negations into the packed-FMA modifiers -- the iso3dfd loop carried 144 `v_xor 0x80000000`
// per 672 vector instructions (r=8: every other coefficient is negative).
struct LinCoefTab { int n; double v[64]; };
template <class P>
constexpr LinCoefTab lin_coef_tab() {
    LinCoefTab t = {0, {}};
    for (int i = 0; i < P::n_lin; i++) {
        const double c = lin_coef<P>(P::lin[i].dx, P::lin[i].dy, P::lin[i].dz);
        bool seen = false;
        for (int k = 0; k < t.n; k++) if (t.v[k] == c) seen = true;
        if (!seen && t.n < 64) t.v[t.n++] = c;
    }
    return t;
}
template <class P>
constexpr int lin_coef_index(double c) {
    constexpr LinCoefTab t = lin_coef_tab<P>();
    for (int k = 0; k < t.n; k++) if (t.v[k] == c) return k;
    return -1;
}
template <typename T> __device__ __forceinline__ void opaque_scalar(T& v) { asm volatile("" : "+s"(v)); }
// coefficient of offset (DX, DY, DZ) out of the kernel's opaque scalars
template <class P, int DX, int DY, int DZ, typename T, int N>
__device__ __forceinline__ T lin_coef_of(const T (&cs)[N]) {
    constexpr double c = lin_coef<P>(DX, DY, DZ);
    if constexpr (c == 0.0) return T(0);
    else return cs[lin_coef_index<P>(c)];
}

constexpr int ce_gcd(int a, int b) { return b == 0 ? a : ce_gcd(b, a % b); }
// physical slot of logical queue entry i at rotation phase ph (queue of N entries)
template <int N>
constexpr int rot(int ph, int i) { return ((ph + i) % N + N) % N; }

// z-vector of N elements (N*sizeof(T) <= 16 bytes)
template <typename T, int N> struct vecn { typedef T type __attribute__((ext_vector_type(N))); };

// elements S..S+N-1 of the concatenation (lo, hi)
template <typename T, int N, int S>
__device__ __forceinline__ typename vecn<T, N>::type zshiftn(typename vecn<T, N>::type lo, typename vecn<T, N>::type hi) {
    static_assert(N == 2 || N == 4, "z-vector of 2 or 4 elements");
    if constexpr (S == 0) return lo;
    else if constexpr (N == 4) {
        if constexpr (S == 1) return __builtin_shufflevector(lo, hi, 1, 2, 3, 4);
        else if constexpr (S == 2) return __builtin_shufflevector(lo, hi, 2, 3, 4, 5);
        else return __builtin_shufflevector(lo, hi, 3, 4, 5, 6);
    } else return __builtin_shufflevector(lo, hi, 1, 2);
}

// VZ_ elements along z per thread (4 -> 16-byte accesses for fp32; 2 halves the per-thread state so
// that twice as many waves fit the register file); TZL_ lanes along z, TYL_ thread rows, RY_ rows
// per thread; CH_ = LDS reads in flight per batch.
template <class P, int VZ_, int TZL_, int TYL_, int RY_, int ROT_, int CH_>
struct StarLinCfg {
    typedef typename P::real_t T;
    static constexpr int VZ = VZ_;
    typedef typename vecn<T, VZ_>::type V;
    static constexpr int TZL = TZL_, TYL = TYL_, RY = RY_, ROT = ROT_, CH = CH_;
    static constexpr int NT = TZL * TYL;
    static constexpr int SG = P::lin_group;
    static constexpr StarRange R = lin_range<P>();
    static constexpr int XL = -R.xlo, XH = R.xhi, YL = -R.ylo, YH = R.yhi, ZL = -R.zlo, ZH = R.zhi;
    static constexpr int NP = (XL > XH ? XL : XH) + 1;   // queue of S planes X-NP+1 .. X
    static constexpr int NA = XH + 1;                    // partial sums of outputs X-XH .. X
    static constexpr int PERIOD = NP / ce_gcd(NP, NA) * NA;
    // LDS slab ring: with the unrolled loop the slab of a plane is a compile-time function of the
    // rotation phase (no per-plane address arithmetic): 2 slabs if the unroll count is even, 3 if it
    // is a multiple of 3, otherwise the loop is unrolled 2*PERIOD times.
    static constexpr int NS = (ROT_ == ROT_UNROLL && PERIOD % 2 != 0 && PERIOD % 3 == 0) ? 3 : 2;
    static constexpr int UNR = (PERIOD % NS == 0) ? PERIOD : 2 * PERIOD;
    static constexpr int ZLV = (ZL + VZ - 1) / VZ, ZHV = (ZH + VZ - 1) / VZ;
    static constexpr int TZ = TZL * VZ, TY = TYL * RY;
    static constexpr int LP = TZ + (ZLV + ZHV) * VZ;
    static constexpr int LROWS = TY + YL + YH;
    static constexpr int NHY = (YL + YH) * TZL;
    static constexpr int NHZ = TY * (ZLV + ZHV);
    static constexpr int NH = NHY + NHZ;
    static constexpr int NHT = (NH + NT - 1) / NT;
    static constexpr int NW = ZLV + 1 + ZHV;
    static constexpr size_t lds_bytes = sizeof(T) * NS * LROWS * LP;
    // y neighbours with a non-zero coefficient, in offset order
    static constexpr int count_y() { int n = 0; for (int dy = -YL; dy <= YH; dy++) if (dy != 0 && lin_coef<P>(0, dy, 0) != 0.0) n++; return n; }
    static constexpr int NY = count_y();
    // rows of the y window shared by the RY rows of a thread: offsets (relative to the thread's first row)
    // -YL .. RY-1+YH without the thread's own rows, kept if some row has a non-zero coefficient for it
    static constexpr bool yw_used(int w) {
        if (w >= 0 && w < RY) return false;
        for (int j = 0; j < RY; j++) { int dy = w - j; if (dy >= -YL && dy <= YH && dy != 0 && lin_coef<P>(0, dy, 0) != 0.0) return true; }
        return false;
    }
    static constexpr int count_yw() { int n = 0; for (int w = -YL; w <= RY - 1 + YH; w++) if (yw_used(w)) n++; return n; }
    static constexpr int NYW = count_yw();
    static constexpr int yw_off(int i) { int n = 0; for (int w = -YL; w <= RY - 1 + YH; w++) if (yw_used(w)) { if (n == i) return w; n++; } return 0; }
    static constexpr int y_off(int i) { int n = 0; for (int dy = -YL; dy <= YH; dy++) if (dy != 0 && lin_coef<P>(0, dy, 0) != 0.0) { if (n == i) return dy; n++; } return 0; }
};

// accessor handed to P::eval_lin(): centre reads only
template <class C>
struct LinAcc {
    typedef typename C::T T;
    typedef typename C::V V;
    const V& scen;                    // star group at the output point
    const V (&cen)[MAX_GROUPS];       // other groups at the output point
    V (&out)[MAX_GROUPS];
    template <int G, int DX, int DY, int DZ>
    __device__ __forceinline__ V rd() const {
        static_assert(DX == 0 && DY == 0 && DZ == 0, "eval_lin: only centre reads are allowed outside the linear form");
        if constexpr (G == C::SG) return scen;
        else return cen[G];
    }
    template <int G>
    __device__ __forceinline__ void wr(V v) { out[G] = v; }
    __device__ __forceinline__ void pin(V&) const {}
    // a - b and a / b of the generated code (the compiler target routes them through the accessor, YaskHip.cpp)
    template <class L, class R> __device__ __forceinline__ V sub(L l, R r) const { return V(l) - V(r); }
    template <class L, class R> __device__ __forceinline__ V div(L l, R r) const { return V(l) / V(r); }
};

// Keep `v` (and, through the memory clobber, later memory reads) at this point of the program order.
template <class V> __device__ __forceinline__ void pin_reg(V& v) { asm volatile("" : "+v"(v) : : "memory"); }
template <class V, typename T> __device__ __forceinline__ V ldv(const T* p) { return *reinterpret_cast<const V*>(p); }
// A uniform pointer made opaque in an SGPR pair: stops hipcc from re-associating `base + plane + lane_offset` into
// a loop-invariant 64-bit *vector* pointer `base + lane_offset` per access group (2 VGPRs each, plus a 64-bit
// vector add per access) -- the accesses below then keep the scalar-base form.
// (the pointer leaves as a global-address-space pointer: the asm would otherwise hide that from hipcc and the
// accesses would become flat_load/flat_store)
#define YKH_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ YKH_GLOBAL T* sbase(T* p) {
    // readfirstlane is free when hipcc already holds the (uniform) value in SGPRs and rescues the cases where it
    // chose to compute the plane offset with vector instructions ("illegal VGPR to SGPR copy" otherwise)
    unsigned long long v = (unsigned long long)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    YKH_GLOBAL T* g = (YKH_GLOBAL T*)(((unsigned long long)hi << 32) | lo);
    asm volatile("" : "+s"(g));
    return g;
}
// uniform base + 32-bit unsigned byte offset (selects the saddr form of global_load/store)
template <class V, typename T> __device__ __forceinline__ V ldv_b(const T* base, unsigned byte_off) {
    return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_off); }
// (the lane offset is made opaque at the point of use as well, so that its zero-extension stays in the basic
// block of the access: instruction selection only forms `global_load v, v_off, s[base]` when it sees both)
__device__ __forceinline__ unsigned voff(unsigned o) { asm volatile("" : "+v"(o)); return o; }
template <class V, typename T> __device__ __forceinline__ V ldv_b(YKH_GLOBAL const T* base, unsigned byte_off) {
    byte_off = voff(byte_off);
    return *(YKH_GLOBAL const V*)((YKH_GLOBAL const char*)base + byte_off); }
// same, for an address that is only element-aligned (global memory takes multi-dword loads at dword alignment)
template <class V, typename T> __device__ __forceinline__ V ldv_u(YKH_GLOBAL const T* base, unsigned byte_off) {
    typedef V __attribute__((aligned(sizeof(T)))) VU;
    byte_off = voff(byte_off);
    return *(YKH_GLOBAL const VU*)((YKH_GLOBAL const char*)base + byte_off); }
template <class V, typename T> __device__ __forceinline__ void stv_b(T* base, unsigned byte_off, V v) {
    *reinterpret_cast<V*>(reinterpret_cast<char*>(base) + byte_off) = v; }
template <class V, typename T> __device__ __forceinline__ void stv_b(YKH_GLOBAL T* base, unsigned byte_off, V v) {
    byte_off = voff(byte_off);
    *(YKH_GLOBAL V*)((YKH_GLOBAL char*)base + byte_off) = v; }
template <class V, typename T> __device__ __forceinline__ void stv_b_nt(YKH_GLOBAL T* base, unsigned byte_off, V v) {
    byte_off = voff(byte_off);
    __builtin_nontemporal_store(v, (YKH_GLOBAL V*)((YKH_GLOBAL char*)base + byte_off)); }
// WRITE-THROUGH store of a 16-byte vector (`buffer_store_dwordx4 ... sc1`).  MI355X_MICROARCH.md, "stores of each flavour": plain, sc0
// and nt stores KEEP the written line in the XCD's L2, sc1 stores DROP it -- an output stream that nobody reads again then stops
// competing with the halo lines of the arriving planes for the 4 MiB (3axis fp64: 1 MB of output per plane and XCD next to 1.15 MB
// of re-used input lines; 5 % of its reads are halo lines that fell out of the L2, profiles/r4_3axis_fetch).  The descriptor is
// built from the uniform plane base (two SGPR moves + two constants); the compiler tracks the store like any other.
template <class V, typename T> __device__ __forceinline__ void stv_b_wt(YKH_GLOBAL T* base, unsigned byte_off, V v) {
    static_assert(sizeof(V) == 16, "write-through stores are 16-byte vectors");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    byte_off = voff(byte_off);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00027000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, byte_off, 0, /*aux: sc1*/ 16);
}
template <class V, typename T> __device__ __forceinline__ void stv_b_wt(T* base, unsigned byte_off, V v) {
    stv_b_nt<V>(base, byte_off, v);         // (generic-pointer callers: no descriptor, keep the non-temporal store)
}
template <class V, typename T> __device__ __forceinline__ V ldv_b_nt(YKH_GLOBAL const T* base, unsigned byte_off) {
    byte_off = voff(byte_off);
    return __builtin_nontemporal_load((YKH_GLOBAL const V*)((YKH_GLOBAL const char*)base + byte_off)); }
template <class V, typename T> __device__ __forceinline__ void stv_b_nt(T* base, unsigned byte_off, V v) {
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(reinterpret_cast<char*>(base) + byte_off)); }
template <class V, typename T> __device__ __forceinline__ V ldv_b_nt(const T* base, unsigned byte_off) {
    return __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_off)); }
template <class V, typename T> __device__ __forceinline__ void stv(T* p, V v) { *reinterpret_cast<V*>(p) = v; }
template <class V, typename T> __device__ __forceinline__ V ldv_nt(const T* p) { return __builtin_nontemporal_load(reinterpret_cast<const V*>(p)); }
template <class V, typename T> __device__ __forceinline__ void stv_nt(T* p, V v) { __builtin_nontemporal_store(v, reinterpret_cast<V*>(p)); }

// NTH : bit 0 = non-temporal loads of centre-only operands and non-temporal stores;
//       bits 1-2 = when the halo vectors of the next plane are requested: 0 with its interior (before the
//       barrier), 1 after the first row of the current plane has been computed, 2 at the end of the
//       iteration.  Later = the neighbour tiles' interior loads of the same lines have reached L2 first.
//       bit 3 = prefetch the star-group planes TWO planes ahead (two alternating register sets): with one
//       workgroup per CU the loop is latency-bound (one plane of loads in flight per iteration), depth 2
//       doubles the bytes in flight for 16 more VGPRs.
//       bit 4 = likewise load the centre-only operands (two thirds of the read bytes of iso3dfd) two output
//       planes ahead.
// MINW: minimum resident waves per SIMD the register allocation must allow (k blocks of T threads per
//       CU <=> k*T/256), MI355X_MICROARCH.md "Register files".
// ABL (profiling only): 1 = no halo loads, 2 = no centre-operand loads, 4 = no stores.
// DESC: the twin that takes its tile and x range from a block descriptor (planned launches, ykh_plan.cpp) and signals when done.
template <class P, int VZ, int TZL, int TYL, int RY, int ROT, int NTH, int MINW, int CH, int ABL = 0, bool DESC = false>
__global__ void __launch_bounds__(TZL* TYL, MINW) starlin_kernel(const PartArgs a) {
    typedef StarLinCfg<P, VZ, TZL, TYL, RY, ROT, CH> C;
    typedef typename C::T T;
    typedef typename C::V V;
    constexpr int NP = C::NP, NA = C::NA, XL = C::XL, XH = C::XH, YL = C::YL;
    constexpr int ZLV = C::ZLV, ZHV = C::ZHV, LP = C::LP, NT = C::NT, NHT = C::NHT, NY = C::NY;
    constexpr int SG = C::SG, NG = P::n_groups;
    constexpr bool NT_STREAMS = (NTH & 1) != 0;
    constexpr int HALO_LATE = (NTH >> 1) & 3;
    constexpr int PD = ((NTH >> 5) & 1) ? 3 : (((NTH >> 3) & 1) ? 2 : 1);      // prefetch depth in planes (bit 5: three)
    constexpr int CD = ((NTH >> 4) & 1) ? 2 : 1;      // prefetch depth of the centre-only operands
    constexpr int TRIP = PD * CD / ce_gcd(PD, CD);    // planes per loop trip (the register sets rotate)
    constexpr bool TAILOPT = ((NTH >> 6) & 1) != 0;   // bit 6: cheap tail planes (see plane())
    constexpr bool WT_OUT = ((NTH >> 7) & 1) != 0 && sizeof(V) == 16;     // bit 7 ("_wt"): write-through (sc1) output stores, see stv_b_wt
    // bits 8-10 ("_ls<K>", round 5): soft LOCK-STEP of the workgroups that share an XCD -- every K planes wave 0 counts the workgroup in at
    // a counter of its XCD (an L2 atomic) and waits, for a bounded number of polls, until all of the XCD's workgroups of the launch have
    // arrived; the plane's own barrier holds the other waves.  Tiles that march within K planes of each other find the halo lines their
    // neighbours streamed as interior still in the XCD's 4 MiB L2, and all of them stream the same DRAM pages at the same time: 3axis fp64
    // 1024^3 fetches 4.4-5.1 % less and runs 4.1 % faster at K = 64, K <= 8 costs more than it saves (profiles/r5_3axis_lockstep).
    // a.sig = the 8 counters (32 words apart), zeroed by the host before the launch; null = no lock-step (Solution::launch_part_variant
    // decides: the hand-shake needs every block resident, equal x-chunks, block i on XCD i % 8).
    constexpr int LS_K = ((NTH >> 8) & 7) == 0 ? 0 : (1 << (((NTH >> 8) & 7) - 1));          // 1 -> 1, 2 -> 2, ... 7 -> 64
    static_assert(CD == 1 || XH > 0, "operand prefetch depth 2 needs a future x range");
    static_assert(!C::R.mixed, "linear form has a mixed-offset term");
    static_assert(NG <= MAX_GROUPS, "too many access groups");
    static_assert(VZ * sizeof(T) <= 16, "z-vector wider than 16 bytes");

    extern __shared__ __attribute__((aligned(16))) unsigned char ykh_smem[];
    T* slab = reinterpret_cast<T*>(ykh_smem);

    // the workgroup's tile and x range: the regular XCD-aware tiling of the box, or a planned launch's descriptor
    const BlockBox bb = block_box<VZ, C::TZ, C::TY, DESC>(a);
    const int tid = threadIdx.x;
    const int lz = tid % TZL, ly = tid / TZL;
    const int zt0 = bb.zt0, yt0 = bb.yt0, xs = bb.xs, xe = bb.xe;
    if (xs >= xe) return;
    const int xlast = xe + XH;       // planes xs .. xlast-1 arrive

    const int myz = zt0 + lz * VZ;
    const bool tile_inside = zt0 >= a.z0 && zt0 + C::TZ <= a.z1 && yt0 + C::TY <= a.y1;      // uniform
    const int zc = clampi(myz, a.az0, a.az1 - VZ);
    const T* __restrict__ sp = (const T*)a.ptr[SG];

    // Plane-relative element offsets, measured from the first allocated element of a plane so that they
    // are non-negative 32-bit numbers: global accesses then use the scalar-base + 32-bit-offset
    // addressing form (one VGPR per address instead of two, no 64-bit vector adds).
    const idx_t org = (idx_t)a.ay0 * a.sy + a.az0;          // offset of the plane's first allocated element
    constexpr LinCoefTab LCT = lin_coef_tab<P>();
    T cs[LCT.n > 0 ? LCT.n : 1];                             // the distinct coefficients, opaque to the optimiser (see LinCoefTab)
    static_for<LCT.n>([&](auto ic) { constexpr int i = decltype(ic)::value; cs[i] = T(LCT.v[i]); opaque_scalar(cs[i]); });
    unsigned roff[RY];
    static_for<RY>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        int y = clampi(yt0 + ly * RY + j, a.ay0, a.ay1 - 1);
        roff[j] = (unsigned)((y - a.ay0) * (int)a.sy + (zc - a.az0)) * (unsigned)sizeof(T);   // bytes
    });
    unsigned hoff[NHT > 0 ? NHT : 1];
    int hlds[NHT > 0 ? NHT : 1];
    static_for<NHT>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int ZV2 = (ZLV + ZHV > 0) ? ZLV + ZHV : 1;
        int h = tid + k * NT;
        int row, zv;
        if (h < C::NHY) {
            int r = h / TZL;
            row = r < YL ? r : r + C::TY;
            zv = ZLV + h % TZL;
        } else {
            int hh = h - C::NHY;
            int r = hh / ZV2, c = hh % ZV2;
            row = YL + r;
            zv = c < ZLV ? c : c + TZL;
        }
        if (h >= C::NH) { row = 0; zv = 0; }
        int y = clampi(yt0 - YL + row, a.ay0, a.ay1 - 1);
        int z = clampi(zt0 - ZLV * VZ + zv * VZ, a.az0, a.az1 - VZ);
        hoff[k] = (unsigned)((y - a.ay0) * (int)a.sy + (z - a.az0)) * (unsigned)sizeof(T);   // bytes
        hlds[k] = (h < C::NH) ? row * LP + zv * VZ : -1;
    });

    auto xplane = [&](int x) -> idx_t { return (idx_t)clampi(x, a.ax0, a.ax1 - 1) * a.sx + org; };

    V pq[NP][RY];
    V acc[NA][RY];
    V nxt[PD][RY];
    V hreg[PD][NHT > 0 ? NHT : 1];
    V cen[CD][NG][RY];     // centre-only operands of the next CD output planes (refilled after use)

    auto load_centres = [&](idx_t pc, auto cs) {
        constexpr int CS = decltype(cs)::value;
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g != SG && analyze_group<P>(g).any) {
                if constexpr (P::group_full[g]) {
                    auto gp = sbase((const T*)a.ptr[g] + pc);
                    static_for<RY>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        if constexpr (ABL & 2) cen[CS][g][j] = V(1);
                        else if constexpr (NT_STREAMS) cen[CS][g][j] = ldv_b_nt<V>(gp, roff[j]);
                        else cen[CS][g][j] = ldv_b<V>(gp, roff[j]);
                    });
                } else {
                    // var over a subset of the domain dims (e.g. the 1-D sponge coefficients of iso3dfd_sponge):
                    // own strides (0 for a missing dim); without the unit-stride dim the value is broadcast
                    // (operands without the marching dim -- iso3dfd_sponge's y and z profiles -- were loaded once, before the loop:
                    //  round 5; they used to be re-loaded every plane, two rows of 64-bit address arithmetic and a load each)
                    constexpr unsigned gd = GroupDims<P>::get(g);
                    if constexpr (gd != 7) {
                        // the var's dims are known here (the compiler target's group_dims, round 6): operands without x live in
                        // cinv[] (below) and are not touched per plane; no stride multiply for a missing dim, one value per row
                        // where z is missing
                        if constexpr ((gd & 1) != 0) {
                            const int xg = (int)((pc - org) / a.sx);      // plane index (uniform)
                            static_for<RY>([&](auto jc) {
                                constexpr int j = decltype(jc)::value;
                                const T* gp = (const T*)a.ptr[g] + (idx_t)xg * a.gsx[g];
                                if constexpr ((gd & 2) != 0) gp += (idx_t)clampi(yt0 + ly * RY + j, a.ay0, a.ay1 - 1) * a.gsy[g];
                                if constexpr ((gd & 4) == 0) cen[CS][g][j] = V(gp[0]);
                                else cen[CS][g][j] = ldv<V>(gp + zc);
                            });
                        }
                    } else
                    if (a.gsx[g] != 0) {
                    const int xg = (int)((pc - org) / a.sx);      // plane index (uniform)
                    static_for<RY>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const int y = clampi(yt0 + ly * RY + j, a.ay0, a.ay1 - 1);
                        const T* gp = (const T*)a.ptr[g] + (idx_t)xg * a.gsx[g] + (idx_t)y * a.gsy[g];
                        if (a.gsz[g] == 0) cen[CS][g][j] = V(gp[0]);
                        else cen[CS][g][j] = ldv<V>(gp + zc);
                    });
                    }
                }
            }
        });
    };
    // partial-dim operands that do not depend on x, known as such at compile time: ONE copy for every plane and operand set (the
    // run-time form below keeps CD copies of each: 48 VGPRs for iso3dfd_sponge's three profiles in the headline's shape)
    V cinv[NG][RY];
    static_for<NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g != SG && analyze_group<P>(g).any && !P::group_full[g]) {
            constexpr unsigned gd = GroupDims<P>::get(g);
            if constexpr (gd != 7 && (gd & 1) == 0)
                static_for<RY>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const T* gp = (const T*)a.ptr[g];
                    if constexpr ((gd & 2) != 0) gp += (idx_t)clampi(yt0 + ly * RY + j, a.ay0, a.ay1 - 1) * a.gsy[g];
                    if constexpr ((gd & 4) == 0) cinv[g][j] = V(gp[0]);
                    else cinv[g][j] = ldv<V>(gp + zc);
                });
        }
    });
    // partial-dim operands that do not depend on x (run-time form): every operand set, once
    static_for<CD>([&](auto cs) {
        constexpr int CS = decltype(cs)::value;
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g != SG && analyze_group<P>(g).any && !P::group_full[g] && GroupDims<P>::get(g) == 7) {
                if (a.gsx[g] == 0)
                    static_for<RY>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const int y = clampi(yt0 + ly * RY + j, a.ay0, a.ay1 - 1);
                        const T* gp = (const T*)a.ptr[g] + (idx_t)y * a.gsy[g];
                        if (a.gsz[g] == 0) cen[CS][g][j] = V(gp[0]);
                        else cen[CS][g][j] = ldv<V>(gp + zc);
                    });
            }
        });
    });
    auto load_interior = [&](int x, auto sc) {
        constexpr int S = decltype(sc)::value;
        auto pp = sbase(sp + xplane(x));
        static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; nxt[S][j] = ldv_b<V>(pp, roff[j]); });
    };
    auto load_halo = [&](int x, auto sc) {
        constexpr int S = decltype(sc)::value;
        auto pp = sbase(sp + xplane(x));
        static_for<NHT>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (ABL & 1) hreg[S][k] = V(0); else hreg[S][k] = ldv_b<V>(pp, hoff[k]);
        });
    };

    // ---- prologue: the past planes xs-NP+1 .. xs-1 (tile interior only), zeroed partial sums,
    // and the first arriving plane xs with its halos.
    static_for<NP - 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        auto pp = sbase(sp + xplane(xs - (NP - 1) + i));
        static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; pq[i][j] = ldv_b<V>(pp, roff[j]); });
    });
    static_for<NA>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; acc[i][j] = V(0); });
    });
    static_for<PD>([&](auto sc) { load_interior(xs + decltype(sc)::value, sc); load_halo(xs + decltype(sc)::value, sc); });
    if constexpr (XH == 0) load_centres(xplane(xs), std::integral_constant<int, 0>{});

    // One arriving plane. PH renames the queue slots when the loop is unrolled PERIOD times.
    // `set_tag`: which prefetch register set holds the arriving plane (PD == 2: the plane's parity).
    // `trip_tag`: position of the plane within a loop trip; selects the alternating register sets.
    auto plane = [&](int xin, auto ph_tag, auto trip_tag, auto tail_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr bool ALLOW_TAIL = decltype(tail_tag)::value;      // only the epilogue of a _tl shape compiles the tail path
        constexpr int S = decltype(trip_tag)::value % PD;
        constexpr int CS = decltype(trip_tag)::value % CD;
        typedef std::integral_constant<int, S> set_t;
        typedef std::integral_constant<int, CS> cset_t;
        const set_t set_tag{};
        constexpr int qn = rot<NP>(PH, NP - 1), an = rot<NA>(PH, NA - 1);
        constexpr int qo = rot<NP>(PH, NP - 1 - XH), ao = rot<NA>(PH, 0);
        T* sb;
        if constexpr (ALLOW_TAIL) sb = slab + ((xin - xs) % C::NS) * (C::LROWS * LP);        // (epilogue: continues the main loop's ring)
        else if constexpr (ROT == ROT_UNROLL) sb = slab + (PH % C::NS) * (C::LROWS * LP);
        else if constexpr (ROT == ROT_TRIP || ROT == ROT_TRIP2) sb = slab + (decltype(trip_tag)::value & 1) * (C::LROWS * LP);   // (even trips)
        else sb = slab + ((xin - xs) & 1) * (C::LROWS * LP);
        const int xo = xin - XH;
        // TAILOPT (NTH bit 6, "_tl"): the last XH planes of a block (xin >= xe) arrive only to complete the outputs that are still
        // waiting for their +x neighbours: they need the plane's own points -- no halo, no slab, no barrier, no y/z sums (the
        // partial sum of output plane xin itself belongs to the next block).  A block then pays ~1 plane-equivalent for its tail
        // instead of XH: the prologue overhead that makes short x-chunks expensive (512^3: 4 chunks of 128 + 8; planned launches
        // of a decomposed rank: 8 chunks of 64 + 8).  Uniform branch; the arithmetic of every stored point is unchanged.
        const bool tail = ALLOW_TAIL && xin >= xe;
        V c[RY], sum[RY];
        if (!tail) {
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                pq[qn][j] = nxt[S][j];
                stv<V>(sb + (YL + ly * RY + j) * LP + (ZLV + lz) * VZ, nxt[S][j]);
            });
            static_for<NHT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr ((k + 1) * NT <= C::NH) stv<V>(sb + hlds[k], hreg[S][k]);      // every thread has one
                else if (hlds[k] >= 0) stv<V>(sb + hlds[k], hreg[S][k]);
            });
            // prefetch the next arriving plane (registers of nxt/hreg are free again)
            load_interior(xin + PD, set_tag);
            if constexpr (HALO_LATE == 0) { if (!ALLOW_TAIL || xin + PD < xe) load_halo(xin + PD, set_tag); }      // (tail planes use no halo)
            __syncthreads();

            const T* colp = sb + (ZLV + lz) * VZ;
            const T* row0 = colp + (YL + ly * RY) * LP;      // the thread's first row in the slab
            // ---- new partial sums for output plane xin, all RY rows of the thread together.
            // Centre and past x come from registers ...
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const T c000 = lin_coef_of<P, 0, 0, 0>(cs);
                c[j] = pq[qn][j];
                sum[j] = c[j] * c000;
                static_for<XL>([&](auto kc) {
                    constexpr int k = decltype(kc)::value + 1;
                    const T ck = lin_coef_of<P, -k, 0, 0>(cs);
                    constexpr int qi = rot<NP>(PH, NP - 1 - k);
                    sum[j] = fmacc(pq[qi][j], ck, sum[j]);
                });
            });
            // ... the thread's own rows are y-neighbours of each other (registers) ...
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                static_for<RY>([&](auto j2c) {
                    constexpr int dy = decltype(j2c)::value - j;
                    if constexpr (dy != 0 && dy >= -YL && dy <= C::YH) {
                        if constexpr (lin_coef<P>(0, dy, 0) != 0.0) {
                            const T ck = lin_coef_of<P, 0, dy, 0>(cs);
                            sum[j] = fmacc(c[decltype(j2c)::value], ck, sum[j]);
                        }
                    }
                });
            });
            // ... the other y-neighbours come from ONE window of slab rows shared by the RY rows (YL+YH reads
            // instead of RY*(YL+YH)); every value read feeds the sums of all rows it is a neighbour of, which
            // also gives RY independent FMA chains.  Reads go in batches of CH, double-buffered: while batch b is
            // summed, batch b+1 is in flight.  The empty asm pins the partial sums (and, through its memory
            // clobber, the later reads) in program order; unconstrained, hipcc issues all reads first and needs
            // VZ*(NYW+NW) more VGPRs.
            {
                constexpr int NYW = C::NYW;
                constexpr int NB = (NYW + CH - 1) / CH;
                V t[NB > 0 ? NB : 1][CH];
                auto issue = [&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    static_for<CH>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, r = b * CH + i;
                        if constexpr (r < NYW) {
                            constexpr int w = C::yw_off(r);
                            t[b][i] = ldv<V>(row0 + w * LP);
                        }
                    });
                };
                if constexpr (NB > 0) issue(std::integral_constant<int, 0>{});
                static_for<NB>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    if constexpr (b + 1 < NB) issue(std::integral_constant<int, b + 1>{});
                    static_for<CH>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, r = b * CH + i;
                        if constexpr (r < NYW) {
                            constexpr int w = C::yw_off(r);
                            static_for<RY>([&](auto jc) {
                                constexpr int j = decltype(jc)::value;
                                constexpr int dy = w - j;
                                if constexpr (dy >= -YL && dy <= C::YH) {
                                    if constexpr (lin_coef<P>(0, dy, 0) != 0.0) {
                                        const T ck = lin_coef_of<P, 0, dy, 0>(cs);
                                        sum[j] = fmacc(t[b][i], ck, sum[j]);
                                    }
                                }
                            });
                        }
                    });
                    static_for<RY>([&](auto jc) { pin_reg(sum[decltype(jc)::value]); });
                });
            }
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                // ... z neighbours from a window of the row itself
                constexpr int NZW = (C::ZL + C::ZH > 0) ? C::NW - 1 : 0;       // window reads (own vector comes from c)
                if constexpr (NZW > 0) {
                    const T* rowc = row0 + j * LP;
                    V zw[C::NW];
                    zw[ZLV] = c[j];
                    static_for<C::NW>([&](auto wc) {
                        constexpr int w = decltype(wc)::value;
                        if constexpr (w != ZLV) zw[w] = ldv<V>(rowc + (w - ZLV) * VZ);
                    });
                    static_for<C::ZL + C::ZH + 1>([&](auto dc) {
                        constexpr int dz = decltype(dc)::value - C::ZL;
                        if constexpr (dz != 0 && lin_coef<P>(0, 0, dz) != 0.0) {
                            constexpr int e = ZLV * VZ + dz;
                            const T ck = lin_coef_of<P, 0, 0, dz>(cs);
                            sum[j] = fmacc(zshiftn<T, VZ, e % VZ>(zw[e / VZ], zw[(e / VZ + 1) < C::NW ? (e / VZ + 1) : e / VZ]), ck, sum[j]);
                        }
                    });
                    pin_reg(sum[j]);
                }
            });
        } else {
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                pq[qn][j] = nxt[S][j];
                c[j] = pq[qn][j];
                sum[j] = V(0);
            });
            if (xin + PD < xlast) load_interior(xin + PD, set_tag);
        }
        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const V s = sum[j];
            acc[an][j] = s;
            // ---- this plane's contribution to the outputs still waiting for their future
            static_for<XH>([&](auto kc) {
                constexpr int k = decltype(kc)::value + 1;
                const T ck = lin_coef_of<P, k, 0, 0>(cs);
                constexpr int ai = rot<NA>(PH, NA - 1 - k);
                acc[ai][j] = fmacc(c[j], ck, acc[ai][j]);
            });
            // ---- output plane xo = xin - XH is complete
            V cj[MAX_GROUPS], out[MAX_GROUPS];
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g != SG && analyze_group<P>(g).any) {
                    if constexpr (!P::group_full[g] && GroupDims<P>::get(g) != 7 && (GroupDims<P>::get(g) & 1) == 0) cj[g] = cinv[g][j];
                    else cj[g] = cen[CS][g][j];
                }
            });
            LinAcc<C> la{pq[qo][j], cj, out};
            P::eval_lin(la, acc[ao][j]);
            const int y = yt0 + ly * RY + j;
            if (tile_inside && !(ABL & 4)) {
                // the whole tile lies in the box (all tiles but those on the high y / z edges): no per-lane predicates,
                // i.e. no exec-mask regions (each costs three scalar instructions and a branch) around the stores
                if (xo >= xs && xo < xe)
                    static_for<P::n_writes>([&](auto wc) {
                        constexpr int g = P::writes[decltype(wc)::value];
                        auto ob = sbase((T*)a.ptr[g] + ((idx_t)xo * a.sx + org));          // uniform
                        if constexpr (WT_OUT) stv_b_wt<V>(ob, roff[j], out[g]);
                        else if constexpr (NT_STREAMS) stv_b_nt<V>(ob, roff[j], out[g]); else stv_b<V>(ob, roff[j], out[g]);
                    });
            } else if (xo >= xs && xo < xe && y < a.y1 && myz < a.z1 && myz + VZ > a.z0) {
                // points inside the box are never clamped, so roff[j] is also the store offset
                static_for<P::n_writes>([&](auto wc) {
                    constexpr int g = P::writes[decltype(wc)::value];
                    auto ob = sbase((T*)a.ptr[g] + ((idx_t)xo * a.sx + org));          // uniform
                    if constexpr (ABL & 4) { if (out[g][0] == T(123.456)) ob[0] = out[g][0]; }
                    else if (myz >= a.z0 && myz + VZ <= a.z1) {
                        if constexpr (WT_OUT) stv_b_wt<V>(ob, roff[j], out[g]);
                        else if constexpr (NT_STREAMS) stv_b_nt<V>(ob, roff[j], out[g]); else stv_b<V>(ob, roff[j], out[g]);
                    } else {
                        static_for<VZ>([&](auto ec) {
                            constexpr int e = decltype(ec)::value;
                            if (myz + e >= a.z0 && myz + e < a.z1) stv_b<T>(ob, roff[j] + e * (unsigned)sizeof(T), out[g][e]);
                        });
                    }
                });
            }
            if constexpr (HALO_LATE == 1 && j == 0) { if (!tail) load_halo(xin + PD, set_tag); }
        });
        if constexpr (HALO_LATE == 2) { if (!tail) load_halo(xin + PD, set_tag); }
        // operands of the next output plane: a whole plane of work hides their latency
        if (xo + CD >= xs && xo + CD < xe) load_centres((idx_t)(xo + CD) * a.sx + org, cset_t{});
    };

    auto rotate = [&]() {
        static_for<NP - 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; pq[i][j] = pq[i + 1][j]; });
        });
        static_for<NA - 1>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; acc[i][j] = acc[i + 1][j]; });
        });
    };
    // ROT_TRIP: the K planes of a loop trip rename the queue slots (as the unrolled loop does), and the queues are rotated by
    // K slots once per trip: NP + NA register moves per K planes instead of (NP - 1 + NA - 1) per plane, with no more code
    // than the K-plane trip has anyway (iso3dfd r=8, K = 2: the moves were 21 % of the vector instructions of the loop).
    auto rotate_by = [&](auto kc) {
        constexpr int K = decltype(kc)::value;
        V tq[NP][RY], ta[NA][RY];
        static_for<NP>([&](auto ic) { constexpr int i = decltype(ic)::value; static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; tq[i][j] = pq[i][j]; }); });
        static_for<NA>([&](auto ic) { constexpr int i = decltype(ic)::value; static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; ta[i][j] = acc[i][j]; }); });
        static_for<NP>([&](auto ic) { constexpr int i = decltype(ic)::value; static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; pq[i][j] = tq[rot<NP>(K, i)][j]; }); });
        static_for<NA>([&](auto ic) { constexpr int i = decltype(ic)::value; static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; acc[i][j] = ta[rot<NA>(K, i)][j]; }); });
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::false_type NoTail;
    // Soft lock-step of the XCD's workgroups (LS_K > 0, see above; ykh_device.hpp xcd_lockstep): at the top of every loop trip.
    [[maybe_unused]] bool ls_dead = false;
    [[maybe_unused]] auto xcd_sync = [&](int x, int step) {
        if constexpr (LS_K > 0 && !DESC) xcd_lockstep<LS_K>(a.sig, x - xs, step, ls_dead);
    };
    if constexpr (TAILOPT) {
        // _tl shapes: the main loop runs whole trips of full planes only (its code is exactly the plain shape's); what is left of
        // the block -- fewer than a trip of full planes and the XH tail planes -- goes plane by plane with moved queues, and the
        // tail planes take the cheap path (plane(): no halo, no slab, no barrier, no y/z sums).
        int x = xs;
        if constexpr (ROT == ROT_TRIP || ROT == ROT_TRIP2) {
            constexpr int K = (ROT == ROT_TRIP2 ? 2 : 1) * (TRIP % 2 == 0 ? TRIP : 2 * TRIP);
            for (; x + K <= xe; x += K) {
                xcd_sync(x, K);
                static_for<K>([&](auto sc) { plane(x + decltype(sc)::value, sc, sc, NoTail{}); });
                rotate_by(std::integral_constant<int, K>{});
            }
        } else if constexpr (ROT == ROT_UNROLL) {
            static_assert(C::UNR % TRIP == 0, "the unroll count must be a multiple of the prefetch trip");
            for (; x + C::UNR <= xe; x += C::UNR) {
                xcd_sync(x, C::UNR);
                static_for<C::UNR>([&](auto phc) { plane(x + decltype(phc)::value, phc, phc, NoTail{}); });
            }
        }
        // (the main loop ran whole trips, so the queues are in canonical order, the prefetch sets and the slab ring continue
        //  from position 0; ROT_MOVE shapes run everything here)
        for (; x < xlast; x += TRIP)
            static_for<TRIP>([&](auto sc) { plane(x + decltype(sc)::value, I0{}, sc, std::true_type{}); rotate(); });
    } else if constexpr (ROT == ROT_MOVE) {
        // PD planes per trip (the prefetch sets alternate); a trip may run past xlast-1: loads are clamped
        // to the allocation and stores are predicated on xo < xe.
        for (int x = xs; x < xlast; x += TRIP) {
            xcd_sync(x, TRIP);
            static_for<TRIP>([&](auto sc) { plane(x + decltype(sc)::value, I0{}, sc, NoTail{}); rotate(); });
        }
    } else if constexpr (ROT == ROT_TRIP || ROT == ROT_TRIP2) {
        constexpr int K = (ROT == ROT_TRIP2 ? 2 : 1) * (TRIP % 2 == 0 ? TRIP : 2 * TRIP);     // even: the two slabs alternate
        for (int x = xs; x < xlast; x += K) {
            xcd_sync(x, K);
            static_for<K>([&](auto sc) { plane(x + decltype(sc)::value, sc, sc, NoTail{}); });
            rotate_by(std::integral_constant<int, K>{});
        }
    } else {
        // UNR planes per trip (queue rotation by renaming); the last trip may run past xlast-1.
        static_assert(C::UNR % TRIP == 0, "the unroll count must be a multiple of the prefetch trip");
        for (int x = xs; x < xlast; x += C::UNR) {
            xcd_sync(x, C::UNR);
            static_for<C::UNR>([&](auto phc) { plane(x + decltype(phc)::value, phc, phc, NoTail{}); });
        }
    }
    if constexpr (DESC) block_done(a, bb.flags);
}

}  // namespace ykh
