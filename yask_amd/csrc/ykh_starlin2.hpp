// ykh_starlin2.hpp -- TWO time steps per pass, fused on chip, for parts of the form
//     S(t+1) = F( S(t) at the point , L ),   L = sum_i c_i * S(t)(x+dx_i, y+dy_i, z+dz_i)
// where S is the ONLY var the part reads (the AxisStencil family `3axis`: src/stencils/SimpleStencils.cpp:61-103 of the
// reference) -- the GPU counterpart of the reference's temporal blocking (`-bt`, src/kernel/lib/context.cpp:747-819)
// for the radii where a level of on-chip storage can hold two steps of state (iso3dfd r=8 and ssg
// cannot).
//
// A workgroup owns an OUTER (y,z) tile and marches along x exactly like starlin_kernel.  Level 1 computes
// B = S(t+1) on the whole outer tile; B never goes to memory (unless STORE_B): the thread keeps its own B values in a
// second register queue and publishes the plane in a second LDS slab, from which level 2 takes the y/z neighbours to
// compute C = S(t+2) on the INNER tile (outer minus a ring of the stencil radius).  Tiles overlap by the ring, i.e. the
// ring of B is computed redundantly by neighbouring workgroups instead of being exchanged -- no inter-workgroup
// traffic, no ordering between workgroups.  Per point and two steps the pass moves 8 B x (outer+halo)/inner of reads and
// 8 B of writes instead of 2 x (8 B x (tile+halo)/tile + 8 B).
//
// Exactness.  (i) C is written OUT OF PLACE (into a scratch slot, ptr[2]); writing S(t+2) over S(t) in place would race
// with neighbouring workgroups that still read S(t) there.  Solution::run_fused() alternates between the slot and the
// scratch and restores the layout at the end.  (ii) Where the outer tile sticks out of the rank's domain, S(t+1) is NOT
// what level 1 computes there but what the (t+1) step slot holds in its pads -- a single rank never updates its pads
// (SURVEY.md appendix C), and step t+2 of a plain run reads exactly those values.  Such points are replaced by loads
// from ptr[1] (the t+1 slot).  (iii) Per-point arithmetic of both levels is that of starlin_kernel (same summation order).
//
// Pipeline of one iteration (arriving plane X of S(t)); one barrier per plane, both slabs double-buffered:
//     slab1[X]   <- plane X of S(t) (interior from the prefetch registers, halos)           level-1 input
//     slab2[X]   <- plane X-XH-1 of B, completed in the previous iteration                   level-2 input
//     barrier
//     level 1: partial sums of plane X, contributions to the XH planes waiting   -> B[X-XH] complete
//     level 2: the same with (slab2, queue of B planes)                          -> C[X-2*XH-1] complete -> store
#pragma once
#include "ykh_starlin.hpp"

namespace ykh {

// like pin_reg() but without the memory clobber: fixes the value's place in program order only
template <class V> __device__ __forceinline__ void keep_reg(V& v) { asm volatile("" : "+v"(v)); }

template <class P>
constexpr bool fused2_eligible() {
    if constexpr (!P::has_lin) return false;
    else {
        if (P::n_groups != 2 || P::n_writes != 1 || P::lin_group != 0 || P::writes[0] != 1) return false;
        if (P::groups[0].var != P::groups[1].var || !P::groups[0].has_step || !P::groups[1].has_step) return false;
        if (P::groups[0].dt != 0 || (P::groups[1].dt != 1 && P::groups[1].dt != -1)) return false;
        if (!P::group_full[0] || !P::group_full[1]) return false;
        if (P::has_domain_cond || P::has_step_cond || P::has_step_cond_dev) return false;
        return !lin_range<P>().mixed;
    }
}

template <class P, int VZ_, int TZL_, int TYL_, int RY_, int CH_>
struct StarLin2Cfg : StarLinCfg<P, VZ_, TZL_, TYL_, RY_, ROT_MOVE, CH_> {
    typedef StarLinCfg<P, VZ_, TZL_, TYL_, RY_, ROT_MOVE, CH_> B;
    static constexpr int ZRL = B::ZLV * B::VZ, ZRH = B::ZHV * B::VZ;            // ring in z (whole vectors)
    static constexpr int TZI = B::TZ - ZRL - ZRH, TYI = B::TY - B::YL - B::YH;   // inner tile = tile stride
    static constexpr int SLAB = B::LROWS * B::LP;                                // elements of one slab
    static constexpr size_t lds_bytes = sizeof(typename B::T) * 4 * SLAB;        // two slabs, double-buffered
    static_assert(TZI > 0 && TYI > 0, "tile smaller than twice the stencil radius");
};

// a.ptr[0] = S(t) (read), a.ptr[1] = the t+1 step slot (pads read; domain written if STORE_B), a.ptr[2] = where S(t+2) goes.
// a.dom_*1: the rank's domain [0, dom) in local indices (the launch box x0..z1 may be a part of it).
template <class P, int VZ, int TZL, int TYL, int RY, int NTH, int MINW, int CH, bool STORE_B>
__global__ void __launch_bounds__(TZL* TYL, MINW) starlin2_kernel(const PartArgs a) {
    typedef StarLin2Cfg<P, VZ, TZL, TYL, RY, CH> C;
    typedef typename C::T T;
    typedef typename C::V V;
    constexpr int NP = C::NP, NA = C::NA, XL = C::XL, XH = C::XH, YL = C::YL, YH = C::YH;
    constexpr int ZLV = C::ZLV, LP = C::LP, NT = C::NT, NHT = C::NHT;
    constexpr bool NT_STREAMS = (NTH & 1) != 0;
    static_assert(fused2_eligible<P>(), "part is not of the single-var linear-star form");
    static_assert(XL == XH, "fused two-step kernel expects a symmetric x range");

    extern __shared__ __attribute__((aligned(16))) unsigned char ykh_smem[];
    T* slab1 = reinterpret_cast<T*>(ykh_smem);
    T* slab2 = slab1 + 2 * C::SLAB;

    const int ntiles = a.ntz * a.nty * a.nxc;
    int bid = blockIdx.x;
    if ((ntiles & 7) == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);       // XCD-aware tile order (see starlin_kernel)
    const int tz_i = bid % a.ntz;
    const int ty_i = (bid / a.ntz) % a.nty;
    const int xc_i = bid / (a.ntz * a.nty);

    const int tid = threadIdx.x;
    const int lz = tid % TZL, ly = tid / TZL;
    const int zb = a.z0 & ~(VZ - 1);
    const int zi0 = zb + tz_i * C::TZI, yi0 = a.y0 + ty_i * C::TYI;           // inner tile origin
    const int zt0 = zi0 - C::ZRL, yt0 = yi0 - YL;                             // outer tile origin
    const int xs = a.x0 + xc_i * a.xchunk;
    const int xe = (xs + a.xchunk < a.x1) ? xs + a.xchunk : a.x1;
    if (xs >= xe) return;

    const int myz = zt0 + lz * VZ;
    const int zc = clampi(myz, a.az0, a.az1 - VZ);
    const T* __restrict__ sp = (const T*)a.ptr[0];
    const T* __restrict__ bp = (const T*)a.ptr[1];
    const idx_t org = (idx_t)a.ay0 * a.sy + a.az0;
    unsigned roff[RY];
    bool in_y[RY];          // row belongs to the inner tile and to the launch box: its C (and B) values are this thread's to store
    bool out_y[RY];         // row lies outside the rank's domain: B comes from the pads of the t+1 slot
    static_for<RY>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int yr = yt0 + ly * RY + j;
        const int y = clampi(yr, a.ay0, a.ay1 - 1);
        roff[j] = (unsigned)((y - a.ay0) * (int)a.sy + (zc - a.az0)) * (unsigned)sizeof(T);
        in_y[j] = yr >= yi0 && yr < yi0 + C::TYI && yr >= a.y0 && yr < a.y1;
        out_y[j] = yr < 0 || yr >= a.dom_y1;
    });
    const bool in_zv = myz >= zi0 && myz + VZ <= zi0 + C::TZI;                 // (the ring is a whole number of vectors)
    bool out_z[VZ];
    bool any_out_z = false;
    static_for<VZ>([&](auto ec) { constexpr int e = decltype(ec)::value; out_z[e] = myz + e < 0 || myz + e >= a.dom_z1; any_out_z |= out_z[e]; });

    unsigned hoff[NHT > 0 ? NHT : 1];
    int hlds[NHT > 0 ? NHT : 1];
    static_for<NHT>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int ZV2 = (ZLV + C::ZHV > 0) ? ZLV + C::ZHV : 1;
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
        hoff[k] = (unsigned)((y - a.ay0) * (int)a.sy + (z - a.az0)) * (unsigned)sizeof(T);
        hlds[k] = (h < C::NH) ? row * LP + zv * VZ : -1;
    });
    auto xplane = [&](int x) -> idx_t { return (idx_t)clampi(x, a.ax0, a.ax1 - 1) * a.sx + org; };

    V pq1[NP][RY], acc1[NA][RY], pq2[NP][RY], acc2[NA][RY];
    V nxt[RY], hreg[NHT > 0 ? NHT : 1], bprev[RY];
    static_for<NP>([&](auto ic) { static_for<RY>([&](auto jc) { pq1[decltype(ic)::value][decltype(jc)::value] = V(0); pq2[decltype(ic)::value][decltype(jc)::value] = V(0); }); });
    static_for<NA>([&](auto ic) { static_for<RY>([&](auto jc) { acc1[decltype(ic)::value][decltype(jc)::value] = V(0); acc2[decltype(ic)::value][decltype(jc)::value] = V(0); }); });
    static_for<RY>([&](auto jc) { bprev[decltype(jc)::value] = V(0); });

    auto load_plane = [&](int x) {
        auto pp = sbase(sp + xplane(x));
        static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; nxt[j] = ldv_b<V>(pp, roff[j]); });
        static_for<NHT>([&](auto kc) { constexpr int k = decltype(kc)::value; hreg[k] = ldv_b<V>(pp, hoff[k]); });
    };

    // Both levels of one iteration, interleaved statement by statement: level 2 works on the B plane of the PREVIOUS
    // iteration, so the two are independent and give every wave two dependency chains (at 2 waves per SIMD a single
    // chain of LDS read -> FMA leaves the SIMD idle most of the time: 2.3 us per iteration serial).
    // Per level L: the arriving plane's own values are in pqL[NP-1][.], its y/z neighbourhood in slab sbL; the plane is
    // added to the partial sums and the completed output plane (XH planes back) is returned in outL.
    auto levels = [&](const T* sbA, V(&pqA)[NP][RY], V(&accA)[NA][RY], V(&outA)[RY],
                      const T* sbB, V(&pqB)[NP][RY], V(&accB)[NA][RY], V(&outB)[RY]) {
        const T* row0A = sbA + (ZLV + lz) * VZ + (YL + ly * RY) * LP;
        const T* row0B = sbB + (ZLV + lz) * VZ + (YL + ly * RY) * LP;
        V cA[RY], sumA[RY], cB[RY], sumB[RY];
        auto head = [&](V(&pq)[NP][RY], V(&c)[RY], V(&sum)[RY]) {
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                constexpr T c000 = T(lin_coef<P>(0, 0, 0));
                c[j] = pq[NP - 1][j];
                sum[j] = c[j] * c000;
                static_for<XL>([&](auto kc) {
                    constexpr int k = decltype(kc)::value + 1;
                    constexpr T ck = T(lin_coef<P>(-k, 0, 0));
                    sum[j] = fmacc(pq[NP - 1 - k][j], ck, sum[j]);
                });
            });
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                static_for<RY>([&](auto j2c) {
                    constexpr int dy = decltype(j2c)::value - j;
                    if constexpr (dy != 0 && dy >= -YL && dy <= YH) {
                        if constexpr (lin_coef<P>(0, dy, 0) != 0.0) {
                            constexpr T ck = T(lin_coef<P>(0, dy, 0));
                            sum[j] = fmacc(c[decltype(j2c)::value], ck, sum[j]);
                        }
                    }
                });
            });
        };
        head(pqA, cA, sumA);
        head(pqB, cB, sumB);
        {
            // y window: rows shared by the RY rows of a thread, batches of CH reads per level, both levels in flight
            constexpr int NYW = C::NYW;
            constexpr int NB = (NYW + CH - 1) / CH;
            V tA[NB > 0 ? NB : 1][CH], tB[NB > 0 ? NB : 1][CH];
            auto issue = [&](auto bc, const T* row0, V(&t)[NB > 0 ? NB : 1][CH]) {
                constexpr int b = decltype(bc)::value;
                static_for<CH>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, r = b * CH + i;
                    if constexpr (r < NYW) { constexpr int w = C::yw_off(r); t[b][i] = ldv<V>(row0 + w * LP); }
                });
            };
            auto use = [&](auto bc, V(&t)[NB > 0 ? NB : 1][CH], V(&sum)[RY]) {
                constexpr int b = decltype(bc)::value;
                static_for<CH>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, r = b * CH + i;
                    if constexpr (r < NYW) {
                        constexpr int w = C::yw_off(r);
                        static_for<RY>([&](auto jc) {
                            constexpr int j = decltype(jc)::value;
                            constexpr int dy = w - j;
                            if constexpr (dy >= -YL && dy <= YH) {
                                if constexpr (lin_coef<P>(0, dy, 0) != 0.0) {
                                    constexpr T ck = T(lin_coef<P>(0, dy, 0));
                                    sum[j] = fmacc(t[b][i], ck, sum[j]);
                                }
                            }
                        });
                    }
                });
            };
            if constexpr (NB > 0) { issue(std::integral_constant<int, 0>{}, row0A, tA); issue(std::integral_constant<int, 0>{}, row0B, tB); }
            static_for<NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                if constexpr (b + 1 < NB) { issue(std::integral_constant<int, b + 1>{}, row0A, tA); issue(std::integral_constant<int, b + 1>{}, row0B, tB); }
                use(bc, tA, sumA);
                use(bc, tB, sumB);
                static_for<RY>([&](auto jc) { pin_reg(sumA[decltype(jc)::value]); });
                static_for<RY>([&](auto jc) { keep_reg(sumB[decltype(jc)::value]); });
            });
        }
        static_for<RY>([&](auto jc) {
            // z neighbours from a window of the row itself, both levels
            constexpr int j = decltype(jc)::value;
            constexpr int NZW = (C::ZL + C::ZH > 0) ? C::NW - 1 : 0;
            if constexpr (NZW > 0) {
                V zwA[C::NW], zwB[C::NW];
                zwA[ZLV] = cA[j];
                zwB[ZLV] = cB[j];
                static_for<C::NW>([&](auto wc) {
                    constexpr int w = decltype(wc)::value;
                    if constexpr (w != ZLV) { zwA[w] = ldv<V>(row0A + j * LP + (w - ZLV) * VZ); zwB[w] = ldv<V>(row0B + j * LP + (w - ZLV) * VZ); }
                });
                static_for<C::ZL + C::ZH + 1>([&](auto dc) {
                    constexpr int dz = decltype(dc)::value - C::ZL;
                    if constexpr (dz != 0 && lin_coef<P>(0, 0, dz) != 0.0) {
                        constexpr int e = ZLV * VZ + dz;
                        constexpr T ck = T(lin_coef<P>(0, 0, dz));
                        constexpr int hi = (e / VZ + 1) < C::NW ? (e / VZ + 1) : e / VZ;
                        sumA[j] = fmacc(zshiftn<T, VZ, e % VZ>(zwA[e / VZ], zwA[hi]), ck, sumA[j]);
                        sumB[j] = fmacc(zshiftn<T, VZ, e % VZ>(zwB[e / VZ], zwB[hi]), ck, sumB[j]);
                    }
                });
                pin_reg(sumA[j]);
                keep_reg(sumB[j]);
            }
        });
        auto tail = [&](V(&pq)[NP][RY], V(&acc)[NA][RY], V(&c)[RY], V(&sum)[RY], V(&outv)[RY]) {
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                acc[NA - 1][j] = sum[j];
                static_for<XH>([&](auto kc) {
                    constexpr int k = decltype(kc)::value + 1;
                    constexpr T ck = T(lin_coef<P>(k, 0, 0));
                    acc[NA - 1 - k][j] = fmacc(c[j], ck, acc[NA - 1 - k][j]);
                });
                V cj[MAX_GROUPS], out[MAX_GROUPS];
                LinAcc<C> la{pq[NP - 1 - XH][j], cj, out};
                P::eval_lin(la, acc[0][j]);
                outv[j] = out[P::writes[0]];
            });
            // rotate the queues (ROT_MOVE)
            static_for<NP - 1>([&](auto ic) { constexpr int i = decltype(ic)::value; static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; pq[i][j] = pq[i + 1][j]; }); });
            static_for<NA - 1>([&](auto ic) { constexpr int i = decltype(ic)::value; static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; acc[i][j] = acc[i + 1][j]; }); });
        };
        tail(pqA, accA, cA, sumA, outA);
        tail(pqB, accB, cB, sumB, outB);
    };

    auto store_plane = [&](T* dst, int x, const V(&val)[RY]) {
        auto ob = sbase(dst + ((idx_t)x * a.sx + org));
        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (in_y[j] && in_zv && myz < a.z1 && myz + VZ > a.z0) {
                // points inside the box are never clamped, so roff[j] is also the store offset
                if (myz >= a.z0 && myz + VZ <= a.z1) {
                    if constexpr (NT_STREAMS) stv_b_nt<V>(ob, roff[j], val[j]); else stv_b<V>(ob, roff[j], val[j]);
                } else {
                    static_for<VZ>([&](auto ec) {
                        constexpr int e = decltype(ec)::value;
                        if (myz + e >= a.z0 && myz + e < a.z1) stv_b<T>(ob, roff[j] + e * (unsigned)sizeof(T), val[j][e]);
                    });
                }
            }
        });
    };

    const int xa0 = xs - 2 * XL, xend = xe + 2 * XH;      // arriving planes of S(t): the pipelines of both levels fill up first
    load_plane(xa0);
    for (int X = xa0; X <= xend; X++) {
        const int par = (X - xa0) & 1;
        T* sb1 = slab1 + par * C::SLAB;
        T* sb2 = slab2 + par * C::SLAB;
        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int lo = (YL + ly * RY + j) * LP + (ZLV + lz) * VZ;
            pq1[NP - 1][j] = nxt[j];
            stv<V>(sb1 + lo, nxt[j]);
            pq2[NP - 1][j] = bprev[j];
            stv<V>(sb2 + lo, bprev[j]);
        });
        static_for<NHT>([&](auto kc) { constexpr int k = decltype(kc)::value; if (hlds[k] >= 0) stv<V>(sb1 + hlds[k], hreg[k]); });
        load_plane(X + 1);
        // B plane of this iteration: values outside the rank's domain come from the t+1 slot's pads
        const int xb = X - XH;
        const bool out_x = xb < 0 || xb >= a.dom_x1;
        V bpad[RY];
        {
            auto bb = sbase(bp + xplane(xb));
            static_for<RY>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (out_x || out_y[j] || any_out_z) bpad[j] = ldv_b<V>(bb, roff[j]); else bpad[j] = V(0);
            });
        }
        __syncthreads();

        V bnew[RY], cnew[RY];
        levels(sb1, pq1, acc1, bnew, sb2, pq2, acc2, cnew);
        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (out_x || out_y[j]) bnew[j] = bpad[j];
            else if (any_out_z) static_for<VZ>([&](auto ec) { constexpr int e = decltype(ec)::value; if (out_z[e]) bnew[j][e] = bpad[j][e]; });
        });
        const int xc = X - 2 * XH - 1;
        if (xc >= xs && xc < xe) store_plane((T*)a.ptr[2], xc, cnew);
        if constexpr (STORE_B) { if (xb >= xs && xb < xe) store_plane((T*)a.ptr[1], xb, bnew); }
        static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; bprev[j] = bnew[j]; });
    }
}

}  // namespace ykh
