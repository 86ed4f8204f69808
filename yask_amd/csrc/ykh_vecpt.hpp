// ykh_vecpt.hpp -- generic vector-per-thread kernel for ANY renderable stencil part (multi-equation,
// multi-var, staggered offsets: ssg's two stages run on it).
//
// Replaces the generated `calc_vectors` loop of the reference for parts that are not single-group stars
// (emitter src/compiler/lib/YaskKernel.cpp:591-719).  Where the naive kernel issues one 4-byte load per
// read per point, here a thread owns VZ consecutive z points and RX consecutive x planes:
//   * reads with dz == 0 are aligned 16-byte loads; reads with dz != 0 are assembled from the two
//     aligned vectors that cover them (register shuffle), so the z-neighbours of one row cost
//     (halo/VZ)+2 loads instead of 2*halo -- hipcc CSEs the repeated aligned loads because all loads
//     of a point precede its stores;
//   * the RX x-planes of a thread share most of their x-neighbours (same addresses -> CSE);
//   * no LDS, no barriers: y/x re-use is left to L1/L2, helped by an XCD-aware block order in which each
//     XCD sweeps its own y-strip plane by plane, so the x-neighbours (previous planes of the same strip)
//     are still in that XCD's L2.
// Every var with all domain dims shares strides and pads (Var::compute_geometry), which is what makes the
// over-reads of aligned vectors at row ends land inside the allocation (pads are multiples of 16 bytes).
#pragma once
#include "ykh_device.hpp"
#include "ykh_starlin.hpp"   // vecn, zshiftn, ldv/stv

namespace ykh {

template <class P, int VZ>
struct VecPtAcc {
    typedef typename P::real_t T;
    typedef typename vecn<T, VZ>::type V;
    const PartArgs& a;
    int x, y, z0;      // z0: first of the VZ points, multiple of VZ
    idx_t c;           // x*sx + y*sy + z0: offset of the point in every var over all domain dims (shared strides)
    template <int G, int DX, int DY, int DZ>
    __device__ __forceinline__ V rd() const {
        constexpr int q = (DZ >= 0) ? DZ / VZ : -((-DZ + VZ - 1) / VZ);   // floor(DZ / VZ)
        constexpr int e = DZ - q * VZ;                                    // 0 .. VZ-1
        const T* pz;
        if constexpr (P::group_full[G])      // uniform base + uniform offset, + c (no per-group strides in SGPRs)
            pz = ((const T*)a.ptr[G] + ((idx_t)DX * a.sx + (idx_t)DY * a.sy + q * VZ)) + c;
        else {
            // (the point's offset once per group, the read's offset as a uniform term: see NaiveAcc::rd)
            constexpr unsigned gd = GroupDims<P>::get(G);
            if constexpr (gd == 7) {
                const idx_t base = (idx_t)x * a.gsx[G] + (idx_t)y * a.gsy[G];
                const T* p = (const T*)a.ptr[G] + base + ((idx_t)DX * a.gsx[G] + (idx_t)DY * a.gsy[G]);
                if (a.gsz[G] == 0) return V(p[0]);                 // var without the unit-stride dim: broadcast
                pz = p + z0 + q * VZ;
            } else {
                // (the var's dims at compile time, see NaiveAcc::rd: no multiply for a missing dim, no branch on the z stride)
                const T* p = (const T*)a.ptr[G];
                if constexpr ((gd & 1) != 0) p += (idx_t)(x + DX) * a.gsx[G];
                if constexpr ((gd & 2) != 0) p += (idx_t)(y + DY) * a.gsy[G];
                if constexpr ((gd & 4) == 0) return V(p[0]);
                pz = p + z0 + q * VZ;
            }
        }
        if constexpr (e == 0) return ldv<V>(pz);
        else return zshiftn<T, VZ, e>(ldv<V>(pz), ldv<V>(pz + VZ));
    }
    template <int G>
    __device__ __forceinline__ void wr(V v) const {
        T* p = P::group_full[G] ? (T*)a.ptr[G] + c : (T*)a.ptr[G] + (idx_t)x * a.gsx[G] + (idx_t)y * a.gsy[G] + z0;
        if (z0 >= a.z0 && z0 + VZ <= a.z1) stv<V>(p, v);
        else
            static_for<VZ>([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                if (z0 + e >= a.z0 && z0 + e < a.z1) p[e] = v[e];
            });
    }
    __device__ __forceinline__ void pin(V&) const {}
    // a - b and a / b of the generated code (the compiler target routes them through the accessor, YaskHip.cpp)
    template <class L, class R> __device__ __forceinline__ V sub(L l, R r) const { return V(l) - V(r); }
    template <class L, class R> __device__ __forceinline__ V div(L l, R r) const { return V(l) / V(r); }
    template <int D>
    __device__ __forceinline__ V idx() const {
        if constexpr (D == 2) { V r; static_for<VZ>([&](auto ec) { constexpr int e = decltype(ec)::value; r[e] = T(z0 + e + a.ofs_z); }); return r; }
        else return V(T(D == 0 ? x + a.ofs_x : y + a.ofs_y));
    }
    __device__ __forceinline__ V step() const { return V(T(a.t)); }
};

// Block = TZL lanes along z (x VZ points) by TYL rows; a block handles RX consecutive x planes.
// Grid (1-D): tiles ordered so that XCD k (= blockIdx % 8) owns y-tiles [k*nty/8, (k+1)*nty/8) and walks
// them x-plane by x-plane (z fastest, then y within the strip, then x).
template <class P, int VZ, int TZL, int TYL, int RX>
__global__ void __launch_bounds__(TZL* TYL) vecpt_kernel(const PartArgs a) {
    const int ntz = a.ntz, nty = a.nty, nxb = a.nxc;       // tiles in z, y; x blocks
    int bid = blockIdx.x;
    int tz_i, ty_i, xb_i;
    if ((nty & 7) == 0) {
        const int xcd = bid & 7, k = bid >> 3, spy = nty >> 3;      // y-tiles per strip
        tz_i = k % ntz;
        ty_i = xcd * spy + (k / ntz) % spy;
        xb_i = k / (ntz * spy);
    } else {
        tz_i = bid % ntz;
        ty_i = (bid / ntz) % nty;
        xb_i = bid / (ntz * nty);
    }
    const int lz = threadIdx.x % TZL, ly = threadIdx.x / TZL;
    const int z0 = (a.z0 & ~(VZ - 1)) + (tz_i * TZL + lz) * VZ;
    const int y = a.y0 + ty_i * TYL + ly;
    if (y >= a.y1 || z0 >= a.z1) return;
    const int xs = a.x0 + xb_i * RX;
    static_for<RX>([&](auto rc) {
        const int x = xs + decltype(rc)::value;
        if (x < a.x1) {
            VecPtAcc<P, VZ> acc{a, x, y, z0, (idx_t)x * a.sx + (idx_t)y * a.sy + z0};
            P::eval(acc);
        }
    });
}

}  // namespace ykh
