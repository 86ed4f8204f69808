// ykh_stencil_tu.hpp -- helpers used by the per-stencil translation unit (stencil_<name>.hip) to
// instantiate kernel variants for the parts emitted by the `cdna4_hip` compiler target and to
// register them with the runtime (ykh_solution_impl()).
#pragma once
#include <string>

#include "ykh_device.hpp"
#include "ykh_starlin.hpp"
#include "ykh_starlin2.hpp"
#include "ykh_vecpt.hpp"
#include "ykh_march.hpp"
#include "ykh_box.hpp"
#include "ykh_subpart.hpp"
#include "ykh_lift2d.hpp"
#include "ykh_fused.hpp"
#include "ykh_runtime.hpp"

namespace ykh {

template <class P>
void launch_naive(const PartArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((naive_kernel<P>), grid, dim3(256), 0, s, a);
}

template <class P, int TZL, int TYL, int RY, int ROT, int ABL = 0>
void launch_star(const PartArgs& a, dim3 grid, hipStream_t s) {
    typedef Star25dCfg<P, TZL, TYL, RY, ROT> C;
    static bool attr_set = false;
    if (!attr_set) {
        // allow more than the default 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&star25d_kernel<P, TZL, TYL, RY, ROT, ABL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((star25d_kernel<P, TZL, TYL, RY, ROT, ABL>), grid, dim3(C::NT), C::lds_bytes, s, a);
}

template <class P>
KernelVariant naive_variant() {
    KernelVariant kv{"naive", false, 64, 4, 0, 256, &launch_naive<P>};
    kv.func = reinterpret_cast<const void*>(&naive_kernel<P>);
    return kv;
}

template <class P>
void launch_cond_bb(const PartArgs& a, dim3 grid, int* dev_out, hipStream_t s) {
    hipLaunchKernelGGL((cond_bb_kernel<P>), grid, dim3(256), 0, s, a, dev_out);
}

template <class P>
void launch_cond_profile(const PartArgs& a, dim3 grid, unsigned* dev_hist, hipStream_t s) {
    hipLaunchKernelGGL((cond_profile_kernel<P>), grid, dim3(256), 0, s, a, dev_hist);
}

// ABL != 0 variants compute WRONG results on purpose (profiling ablations); their names start with
// "abl" and neither the default selection nor the auto-tuner ever picks them.
template <class P, int TZL, int TYL, int RY, int ROT, int ABL = 0>
KernelVariant star_variant() {
    typedef Star25dCfg<P, TZL, TYL, RY, ROT> C;
    static const std::string name = std::string(ABL ? "abl" + std::to_string(ABL) + "_" : "") + "star25d_z" +
                                    std::to_string(C::TZ) + "_y" + std::to_string(C::TY) + "_r" +
                                    std::to_string(RY) + (ROT == ROT_UNROLL ? "_u" : "_m");
    KernelVariant kv{name.c_str(), true, C::TZ, C::TY, C::lds_bytes, C::NT, &launch_star<P, TZL, TYL, RY, ROT, ABL>};
    kv.func = reinterpret_cast<const void*>(&star25d_kernel<P, TZL, TYL, RY, ROT, ABL>);
    return kv;
}

template <class P, int VZ, int TZL, int TYL, int RY, int ROT, int NTH, int MINW, int CH, int ABL = 0, bool DESC = false>
void launch_starlin(const PartArgs& a, dim3 grid, hipStream_t s) {
    typedef StarLinCfg<P, VZ, TZL, TYL, RY, ROT, CH> C;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&starlin_kernel<P, VZ, TZL, TYL, RY, ROT, NTH, MINW, CH, ABL, DESC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((starlin_kernel<P, VZ, TZL, TYL, RY, ROT, NTH, MINW, CH, ABL, DESC>), grid, dim3(C::NT), C::lds_bytes, s, a);
}

template <class P, int VZ, int TZL, int TYL, int RX>
void launch_vecpt(const PartArgs& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((vecpt_kernel<P, VZ, TZL, TYL, RX>), grid, dim3(TZL * TYL), 0, s, a);
}
// Generic vector-per-thread kernel (ykh_vecpt.hpp). Name: vecpt_v<VZ>_z<tile z>_y<tile y>_x<planes per thread>
template <class P, int VZ, int TZL, int TYL, int RX>
KernelVariant vecpt_variant() {
    static const std::string name = "vecpt_v" + std::to_string(VZ) + "_z" + std::to_string(TZL * VZ) + "_y" +
                                    std::to_string(TYL) + "_x" + std::to_string(RX);
    KernelVariant kv{name.c_str(), true, TZL * VZ, TYL, 0, TZL * TYL, &launch_vecpt<P, VZ, TZL, TYL, RX>};
    kv.vz = VZ;
    kv.rx = RX;
    kv.func = reinterpret_cast<const void*>(&vecpt_kernel<P, VZ, TZL, TYL, RX>);
    return kv;
}

template <class P, int VZ, int TZL, int TYL, int MINW, int RY = 1, bool PIN = false, int PD = 1, int NT = 0, bool DESC = false>
void launch_march(const PartArgs& a, dim3 grid, hipStream_t s) {
    typedef MarchCfg<P, VZ, TZL, TYL, RY, (NT & 2) != 0> C;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&march_kernel<P, VZ, TZL, TYL, MINW, RY, PIN, PD, NT, DESC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((march_kernel<P, VZ, TZL, TYL, MINW, RY, PIN, PD, NT, DESC>), grid, dim3(C::NT), C::lds_bytes, s, a);
}
// Generic marching kernel (ykh_march.hpp). Name: march_v<VZ>_z<tile z>_y<tile y>[_r<rows>][_pin][_pd<planes ahead>][_nt][_hr]_w<min waves/SIMD>
// (NT: flag bits, 1 = non-temporal one-touch streams, 2 = halo rings, 4 = packed subtractions (_ps), 8 = reciprocal divisions (_fd), 16 / 32 / 64 = queue renaming in trips of 2 / 4 / 8 planes, 128 = late refill of the once operands (_lo), 256 = write-through output stores (_wt, ykh_starlin.hpp stv_b_wt))
template <class P, int VZ, int TZL, int TYL, int MINW, int RY = 1, bool PIN = false, int PD = 1, int NT = 0>
KernelVariant march_variant() {
    typedef MarchCfg<P, VZ, TZL, TYL, RY, (NT & 2) != 0> C;
    static_assert(C::lds_bytes <= 160 * 1024, "march tile does not fit the 160 KiB LDS");
    static const std::string name = "march_v" + std::to_string(VZ) + "_z" + std::to_string(C::TZ) + "_y" +
                                    std::to_string(C::TY) + (RY > 1 ? "_r" + std::to_string(RY) : "") + (PIN ? "_pin" : "") + (PD > 1 ? "_pd" + std::to_string(PD) : "") + ((NT & 1) ? "_nt" : "") + ((NT & 2) ? "_hr" : "") + ((NT & 4) ? "_ps" : "") + ((NT & 8) ? "_fd" : "") + ((NT & 64) ? "_t8" : ((NT & 32) ? "_t4" : ((NT & 16) ? "_t2" : ""))) + ((NT & 128) ? "_lo" : "") + ((NT & 256) ? "_wt" : "") +
                                    (((NT >> 9) & 7) ? "_ls" + std::to_string(1 << (((NT >> 9) & 7) - 1)) : "") + "_w" + std::to_string(MINW);
    KernelVariant kv{name.c_str(), true, C::TZ, C::TY, C::lds_bytes, C::NT, &launch_march<P, VZ, TZL, TYL, MINW, RY, PIN, PD, NT>};
    kv.vz = VZ;
    kv.lockstep = ((NT >> 9) & 7) != 0;
    kv.func = reinterpret_cast<const void*>(&march_kernel<P, VZ, TZL, TYL, MINW, RY, PIN, PD, NT>);
    kv.xover = C::XOVER;                  // a block's prologue: the deepest x queue it fills before its first plane
    return kv;
}
// The same shape plus its descriptor-reading twin (planned launches of a decomposed rank): a second instantiation of the kernel,
// so only the shapes a decomposed run may use are registered this way.
template <class P, int VZ, int TZL, int TYL, int MINW, int RY = 1, bool PIN = false, int PD = 1, int NT = 0>
KernelVariant march_variant_planned() {
    KernelVariant kv = march_variant<P, VZ, TZL, TYL, MINW, RY, PIN, PD, NT>();
    kv.launch_desc = &launch_march<P, VZ, TZL, TYL, MINW, RY, PIN, PD, NT, true>;
    kv.func_desc = reinterpret_cast<const void*>(&march_kernel<P, VZ, TZL, TYL, MINW, RY, PIN, PD, NT, true>);
    return kv;
}

template <class P, int VZ, int TZL, int TYL, int RY, int MINW, int FL = 0, int LDS_KB = 160, bool DESC = false>
void launch_box(const PartArgs& a, dim3 grid, hipStream_t s) {
    typedef BoxCfg<P, VZ, TZL, TYL, RY, LDS_KB> C;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&box_kernel<P, VZ, TZL, TYL, RY, MINW, FL, LDS_KB, DESC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((box_kernel<P, VZ, TZL, TYL, RY, MINW, FL, LDS_KB, DESC>), grid, dim3(C::NT), C::lds_bytes, s, a);
}
// Plane-ring marching kernel (ykh_box.hpp). Name: box_v<VZ>_z<tile z>_y<tile y>_r<rows per thread>[_nt][_p2][_l<LDS budget, KiB>]_w<min waves/SIMD>
template <class P, int VZ, int TZL, int TYL, int RY, int MINW, int FL = 0, int LDS_KB = 160>
KernelVariant box_variant() {
    typedef BoxCfg<P, VZ, TZL, TYL, RY, LDS_KB> C;
    static_assert(C::lds_bytes <= 160 * 1024, "box tile does not fit the 160 KiB LDS");
    static const std::string name = "box_v" + std::to_string(VZ) + "_z" + std::to_string(C::TZ) + "_y" + std::to_string(C::TY) + "_r" +
                                    std::to_string(RY) + ((FL & 1) ? "_nt" : "") + ((FL & 4) ? "_p2" : "") + (LDS_KB != 160 ? "_l" + std::to_string(LDS_KB) : "") +
                                    "_w" + std::to_string(MINW);
    KernelVariant kv{name.c_str(), true, C::TZ, C::TY, C::lds_bytes, C::NT, &launch_box<P, VZ, TZL, TYL, RY, MINW, FL, LDS_KB>};
    kv.vz = VZ;
    kv.func = reinterpret_cast<const void*>(&box_kernel<P, VZ, TZL, TYL, RY, MINW, FL, LDS_KB>);
    kv.xover = C::XOVER;
    return kv;
}
// ... plus its descriptor-reading twin (planned launches of a decomposed rank; see march_variant_planned)
template <class P, int VZ, int TZL, int TYL, int RY, int MINW, int FL = 0, int LDS_KB = 160>
KernelVariant box_variant_planned() {
    KernelVariant kv = box_variant<P, VZ, TZL, TYL, RY, MINW, FL, LDS_KB>();
    kv.launch_desc = &launch_box<P, VZ, TZL, TYL, RY, MINW, FL, LDS_KB, true>;
    kv.func_desc = reinterpret_cast<const void*>(&box_kernel<P, VZ, TZL, TYL, RY, MINW, FL, LDS_KB, true>);
    return kv;
}

// ---- a part as K clusters of its equations, one launch each (ykh_subpart.hpp).  The runtime sees ONE variant whose launch function
// issues the K kernels back to back on the stream (same tile shape, same grid, same PartArgs: a cluster's kernel touches only the
// groups its equations use).  Name: c<K>_<name of the shape the clusters run on>.
template <class P, int K, int VZ, int TZL, int TYL, int RX, int... C>
void launch_vecpt_clusters_(const PartArgs& a, dim3 grid, hipStream_t s, std::integer_sequence<int, C...>) {
    (launch_vecpt<SubPart<P, cluster_mask<P, K>(C)>, VZ, TZL, TYL, RX>(a, grid, s), ...);
}
template <class P, int K, int VZ, int TZL, int TYL, int RX>
void launch_vecpt_clusters(const PartArgs& a, dim3 grid, hipStream_t s) {
    launch_vecpt_clusters_<P, K, VZ, TZL, TYL, RX>(a, grid, s, std::make_integer_sequence<int, K>{});
}
template <class P, int K, int VZ, int TZL, int TYL, int RX>
KernelVariant vecpt_cluster_variant() {
    typedef SubPart<P, cluster_mask<P, K>(0)> S0;
    KernelVariant kv = vecpt_variant<S0, VZ, TZL, TYL, RX>();
    static const std::string name = "c" + std::to_string(K) + "_" + kv.name;
    kv.name = name.c_str();
    kv.launch = &launch_vecpt_clusters<P, K, VZ, TZL, TYL, RX>;
    static_assert(K - 1 <= 7, "KernelVariant::more_funcs");
    static_for<K - 1>([&](auto cc) {
        constexpr int c = decltype(cc)::value + 1;
        kv.more_funcs[kv.n_more_funcs++] = vecpt_variant<SubPart<P, cluster_mask<P, K>(c)>, VZ, TZL, TYL, RX>().func;
    });
    return kv;
}
template <class P, int K, int VZ, int TZL, int TYL, int MINW, int NT, int... C>
void launch_march_clusters_(const PartArgs& a, dim3 grid, hipStream_t s, std::integer_sequence<int, C...>) {
    (launch_march<SubPart<P, cluster_mask<P, K>(C)>, VZ, TZL, TYL, MINW, 1, false, 1, NT>(a, grid, s), ...);
}
template <class P, int K, int VZ, int TZL, int TYL, int MINW, int NT>
void launch_march_clusters(const PartArgs& a, dim3 grid, hipStream_t s) {
    launch_march_clusters_<P, K, VZ, TZL, TYL, MINW, NT>(a, grid, s, std::make_integer_sequence<int, K>{});
}
// every cluster of the part fits the marching kernel at this tile (eligible, slabs within the LDS)
template <class P, int K, int VZ, int TZL, int TYL, int NT, int... C>
constexpr bool march_clusters_fit_(std::integer_sequence<int, C...>) {
    return ((march_eligible<SubPart<P, cluster_mask<P, K>(C)>>() &&
             MarchCfg<SubPart<P, cluster_mask<P, K>(C)>, VZ, TZL, TYL, 1, (NT & 2) != 0>::lds_bytes <= 160 * 1024) && ...);
}
template <class P, int K, int VZ, int TZL, int TYL, int NT>
constexpr bool march_clusters_fit() { return march_clusters_fit_<P, K, VZ, TZL, TYL, NT>(std::make_integer_sequence<int, K>{}); }
template <class P, int K, int VZ, int TZL, int TYL, int NT, int... C>
constexpr size_t march_clusters_lds_(std::integer_sequence<int, C...>) {
    size_t m = 0;
    ((m = MarchCfg<SubPart<P, cluster_mask<P, K>(C)>, VZ, TZL, TYL, 1, (NT & 2) != 0>::lds_bytes > m ? MarchCfg<SubPart<P, cluster_mask<P, K>(C)>, VZ, TZL, TYL, 1, (NT & 2) != 0>::lds_bytes : m), ...);
    return m;
}
template <class P, int K, int VZ, int TZL, int TYL, int MINW, int NT>
KernelVariant march_cluster_variant() {
    typedef SubPart<P, cluster_mask<P, K>(0)> S0;
    KernelVariant kv = march_variant<S0, VZ, TZL, TYL, MINW, 1, false, 1, NT>();
    static const std::string name = "c" + std::to_string(K) + "_" + kv.name;
    kv.name = name.c_str();
    kv.launch = &launch_march_clusters<P, K, VZ, TZL, TYL, MINW, NT>;
    kv.lds_bytes = march_clusters_lds_<P, K, VZ, TZL, TYL, NT>(std::make_integer_sequence<int, K>{});
    static_assert(K - 1 <= 7, "KernelVariant::more_funcs");
    static_for<K - 1>([&](auto cc) {
        constexpr int c = decltype(cc)::value + 1;
        kv.more_funcs[kv.n_more_funcs++] = march_variant<SubPart<P, cluster_mask<P, K>(c)>, VZ, TZL, TYL, MINW, 1, false, 1, NT>().func;
    });
    return kv;
}

// Linear-star-form kernel (ykh_starlin.hpp); only for parts with P::has_lin.
// Name: starlin_v<VZ>_z<tile z>_y<tile y>_r<rows/thread>_{u|m}[_nt]_w<min waves/SIMD>_c<LDS batch>
template <class P, int VZ, int TZL, int TYL, int RY, int ROT, int NTH, int MINW, int CH, int ABL = 0>
KernelVariant starlin_variant() {
    typedef StarLinCfg<P, VZ, TZL, TYL, RY, ROT, CH> C;
    static const std::string name = std::string(ABL ? "abl" + std::to_string(ABL) + "_" : "") + "starlin_v" + std::to_string(VZ) +
                                    "_z" + std::to_string(C::TZ) + "_y" + std::to_string(C::TY) + "_r" +
                                    std::to_string(RY) + (ROT == ROT_UNROLL ? "_u" : (ROT == ROT_TRIP ? "_t" : (ROT == ROT_TRIP2 ? "_t2" : "_m"))) + ((NTH & 1) ? "_nt" : "") +
                                    (((NTH >> 1) & 3) ? "_hl" + std::to_string((NTH >> 1) & 3) : "") + ((NTH & 32) ? "_pd3" : ((NTH & 8) ? "_pd2" : "")) + ((NTH & 16) ? "_cd2" : "") + ((NTH & 64) ? "_tl" : "") + ((NTH & 128) ? "_wt" : "") +
                                    (((NTH >> 8) & 7) ? "_ls" + std::to_string(1 << (((NTH >> 8) & 7) - 1)) : "") + "_w" +
                                    std::to_string(MINW) + "_c" + std::to_string(CH);
    KernelVariant kv{name.c_str(), true, C::TZ, C::TY, C::lds_bytes, C::NT, &launch_starlin<P, VZ, TZL, TYL, RY, ROT, NTH, MINW, CH, ABL>};
    kv.lockstep = ((NTH >> 8) & 7) != 0;
    kv.vz = VZ;
    kv.func = reinterpret_cast<const void*>(&starlin_kernel<P, VZ, TZL, TYL, RY, ROT, NTH, MINW, CH, ABL>);
    kv.xover = C::XH + 1;                 // a block runs XH plane-iterations before its first output plane (+ the queue loads)
    return kv;
}
template <class P, int VZ, int TZL, int TYL, int RY, int ROT, int NTH, int MINW, int CH>
KernelVariant starlin_variant_planned() {        // the shape plus its descriptor-reading twin (see march_variant_planned)
    KernelVariant kv = starlin_variant<P, VZ, TZL, TYL, RY, ROT, NTH, MINW, CH, 0>();
    kv.launch_desc = &launch_starlin<P, VZ, TZL, TYL, RY, ROT, NTH, MINW, CH, 0, true>;
    kv.func_desc = reinterpret_cast<const void*>(&starlin_kernel<P, VZ, TZL, TYL, RY, ROT, NTH, MINW, CH, 0, true>);
    return kv;
}


// Two-steps-per-pass kernel (ykh_starlin2.hpp); only for parts with fused2_eligible<P>().
// Name: starlin2_v<VZ>_z<outer tile z>_y<outer tile y>_r<rows/thread>[_nt]_w<min waves/SIMD>_c<LDS batch>
template <class P, int VZ, int TZL, int TYL, int RY, int NTH, int MINW, int CH>
void launch_starlin2(const PartArgs& a, dim3 grid, hipStream_t s, bool store_b) {
    typedef StarLin2Cfg<P, VZ, TZL, TYL, RY, CH> C;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&starlin2_kernel<P, VZ, TZL, TYL, RY, NTH, MINW, CH, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&starlin2_kernel<P, VZ, TZL, TYL, RY, NTH, MINW, CH, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
        attr_set = true;
    }
    if (store_b) hipLaunchKernelGGL((starlin2_kernel<P, VZ, TZL, TYL, RY, NTH, MINW, CH, true>), grid, dim3(C::NT), C::lds_bytes, s, a);
    else hipLaunchKernelGGL((starlin2_kernel<P, VZ, TZL, TYL, RY, NTH, MINW, CH, false>), grid, dim3(C::NT), C::lds_bytes, s, a);
}
template <class P, int VZ, int TZL, int TYL, int RY, int NTH, int MINW, int CH>
Fused2Variant fused2_variant() {
    typedef StarLin2Cfg<P, VZ, TZL, TYL, RY, CH> C;
    static_assert(C::lds_bytes <= 160 * 1024, "fused tile does not fit the 160 KiB LDS");
    static const std::string name = "starlin2_v" + std::to_string(VZ) + "_z" + std::to_string(C::TZ) + "_y" + std::to_string(C::TY) +
                                    "_r" + std::to_string(RY) + ((NTH & 1) ? "_nt" : "") + "_w" + std::to_string(MINW) + "_c" + std::to_string(CH);
    Fused2Variant f;
    f.name = name.c_str();
    f.tz = C::TZ; f.ty = C::TY; f.tzi = C::TZI; f.tyi = C::TYI; f.vz = VZ; f.xr = C::XH;
    f.lds_bytes = C::lds_bytes; f.threads = C::NT;
    f.launch = &launch_starlin2<P, VZ, TZL, TYL, RY, NTH, MINW, CH>;
    f.func = reinterpret_cast<const void*>(&starlin2_kernel<P, VZ, TZL, TYL, RY, NTH, MINW, CH, false>);
    return f;
}

// ---- fused scratch groups (ykh_fused.hpp)
template <class TR, class LIST, const int* LEVEL, int TI, int TJ, int NT>
void launch_fused2d(const PartArgs* dev_args, const FusedGeom& g, unsigned grid, hipStream_t s) {
    typedef FusedCfg<TR, LIST, LEVEL, TI, TJ> C;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fused2d_kernel<TR, LIST, LEVEL, TI, TJ, NT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((fused2d_kernel<TR, LIST, LEVEL, TI, TJ, NT>), dim3(grid), dim3(NT), C::lds_bytes, s, dev_args, g);
}
// registers the group at every tile shape of the candidate list whose slots fit the LDS (largest first: the static default).  Which one is
// fastest depends on the group: a large tile recomputes less of its halo ring (swe2d, halos of 5 + 4: 1.46x the points at 32 x 64, 2.0x
// at 32 x 32), a small one lets TWO workgroups share a CU's LDS so that one computes while the other waits at a level's barrier
// (swe2d: 143 KB of slots at 32 x 64, 81 KB at 32 x 32: 1.91 -> 1.72 ms; wave2d, 6 slots, is the other way round: 0.36 vs 0.48 ms;
// same-box A/B, job r6w).  prepare_solution() times them (Solution::tune_variants).
template <class TR, class LIST, const int* LEVEL, int TI, int TJ, int NT>
void add_fused_cfg(SolnImpl& s, int first_stage, int last_stage) {
    typedef FusedCfg<TR, LIST, LEVEL, TI, TJ> C;
    if constexpr (C::ok) {
        FusedGroupImpl fg;
        fg.first_stage = first_stage; fg.last_stage = last_stage; fg.n_parts = LIST::N;
        fg.lds_bytes = C::lds_bytes; fg.threads = NT; fg.ti = TI; fg.tj = TJ; fg.n_slots = C::tab.n_slots;
        for (int v = 0; v < TR::n_vars && v < FUSED_MAX_VARS; v++) fg.n_scratch_vars += C::tab.first[v] >= 0;
        fg.launch = &launch_fused2d<TR, LIST, LEVEL, TI, TJ, NT>;
        fg.func = reinterpret_cast<const void*>(&fused2d_kernel<TR, LIST, LEVEL, TI, TJ, NT>);
        s.fused.push_back(fg);
    }
}
template <class TR, class LIST, const int* LEVEL>
void add_fused_group(SolnImpl& s, int first_stage, int last_stage) {
#ifdef YKH_FUSED_TI
    // (experiment builds: ONE tile shape named on the compiler's command line, e.g. -DYKH_FUSED_TI=24 -DYKH_FUSED_TJ=40)
    add_fused_cfg<TR, LIST, LEVEL, YKH_FUSED_TI, YKH_FUSED_TJ, 1024>(s, first_stage, last_stage);
    return;
#endif
    add_fused_cfg<TR, LIST, LEVEL, 32, 64, 1024>(s, first_stage, last_stage);
    add_fused_cfg<TR, LIST, LEVEL, 32, 32, 1024>(s, first_stage, last_stage);
    add_fused_cfg<TR, LIST, LEVEL, 16, 64, 1024>(s, first_stage, last_stage);
    add_fused_cfg<TR, LIST, LEVEL, 8, 64, 512>(s, first_stage, last_stage);
}

}  // namespace ykh
