// ykh_util_kernels.hip -- small data-movement kernels around the stencil sweep.
//
// GPU counterparts of the reference's device regions D3 (halo pack/unpack and slice copies,
// YkVarBase::_copy_vecs_in_slice, src/kernel/lib/yk_var.hpp:1900-2040) and D4 (var init,
// src/kernel/lib/generic_var.cpp:137-200), plus reductions (yk_var::reduce_elements_in_slice,
// src/kernel/lib/yk_var_apis.cpp) and the ref-vs-opt compare (YkVarBase::compare,
// src/kernel/lib/yk_var.cpp:401-477; tolerance rule src/kernel/lib/realv.hpp:974-994).
// All are HBM-bound copies: threads run along z (unit stride) so accesses coalesce.
#include <cstdint>

#include "ykh_runtime.hpp"

namespace ykh {

template <typename T>
__device__ __forceinline__ T* box_elem(const BoxCopyArgs& a, idx_t i, idx_t j, idx_t k) {
    return (T*)a.var_base + (a.lo[0] + i) * a.sx + (a.lo[1] + j) * a.sy + (a.lo[2] + k) * a.sz;
}

// grid: x over z (blocks of 256), y over box y, z over box x (grid-strided)
#define BOX_LOOP_BEGIN                                                         \
    const idx_t k = (idx_t)blockIdx.x * blockDim.x + threadIdx.x;              \
    if (k >= a.n[2]) return;                                                   \
    for (idx_t i = blockIdx.z; i < a.n[0]; i += gridDim.z)                     \
        for (idx_t j = blockIdx.y; j < a.n[1]; j += gridDim.y) {
#define BOX_LOOP_END }

template <typename TV, typename TB>
__global__ void box_gather_k(const BoxCopyArgs a) {
    BOX_LOOP_BEGIN
    ((TB*)a.buf)[i * a.bs[0] + j * a.bs[1] + k * a.bs[2]] = (TB)*box_elem<TV>(a, i, j, k);
    BOX_LOOP_END
}
template <typename TV, typename TB>
__global__ void box_scatter_k(const BoxCopyArgs a) {
    BOX_LOOP_BEGIN
    *box_elem<TV>(a, i, j, k) = (TV)((const TB*)a.buf)[i * a.bs[0] + j * a.bs[1] + k * a.bs[2]];
    BOX_LOOP_END
}
template <typename TV>
__global__ void box_fill_k(const BoxCopyArgs a, double v) {
    BOX_LOOP_BEGIN
    *box_elem<TV>(a, i, j, k) = (TV)v;
    BOX_LOOP_END
}
template <typename TV>
__global__ void box_add_k(const BoxCopyArgs a, double v) {
    BOX_LOOP_BEGIN
    TV* p = box_elem<TV>(a, i, j, k);
    *p = (TV)(*p + (TV)v);
    BOX_LOOP_END
}

// logical-index hash; MUST stay identical to oracle/stencil_oracle.c:yo_hash_unit_i
__device__ __forceinline__ double hash_unit(idx_t vid, idx_t slot, idx_t x, idx_t y, idx_t z) {
    unsigned u = (unsigned)x * 0x9E3779B1u ^ (unsigned)y * 0x85EBCA77u ^ (unsigned)z * 0xC2B2AE3Du ^
                 (unsigned)slot * 0x27D4EB2Fu ^ (unsigned)vid * 0x165667B1u;
    u ^= u >> 15; u *= 0x2C1B3C6Du; u ^= u >> 12; u *= 0x297A2D39u; u ^= u >> 15;
    return (double)(int)u * (1.0 / 2147483648.0);
}
template <typename TV>
__global__ void box_hash_k(const BoxCopyArgs a, double offset, double scale, idx_t vid, idx_t slot,
                           idx_t gx, idx_t gy, idx_t gz) {
    BOX_LOOP_BEGIN
    // gx,gy,gz = global index of box element (0,0,0); unused dims contribute 0
    idx_t X = a.sx ? gx + i : 0, Y = a.sy ? gy + j : 0, Z = a.sz ? gz + k : 0;
    *box_elem<TV>(a, i, j, k) = (TV)(offset + scale * hash_unit(vid, slot, X, Y, Z));
    BOX_LOOP_END
}

static dim3 box_grid(const BoxCopyArgs& a) {
    unsigned gx = (unsigned)((a.n[2] + 255) / 256);
    unsigned gy = (unsigned)(a.n[1] < 1024 ? a.n[1] : 1024);
    unsigned gz = (unsigned)(a.n[0] < 256 ? a.n[0] : 256);
    return dim3(gx ? gx : 1, gy ? gy : 1, gz ? gz : 1);
}
static bool box_empty(const BoxCopyArgs& a) { return a.n[0] <= 0 || a.n[1] <= 0 || a.n[2] <= 0; }

#define DISPATCH2(kern, ...)                                                                        \
    do {                                                                                            \
        if (box_empty(a)) return;                                                                   \
        dim3 g = box_grid(a);                                                                       \
        if (a.var_elem_bytes == 4 && a.buf_elem_bytes == 4) kern<float, float><<<g, 256, 0, s>>>(__VA_ARGS__);        \
        else if (a.var_elem_bytes == 4 && a.buf_elem_bytes == 8) kern<float, double><<<g, 256, 0, s>>>(__VA_ARGS__);  \
        else if (a.var_elem_bytes == 8 && a.buf_elem_bytes == 4) kern<double, float><<<g, 256, 0, s>>>(__VA_ARGS__);  \
        else kern<double, double><<<g, 256, 0, s>>>(__VA_ARGS__);                                   \
        YKH_HIP(hipGetLastError());                                                                 \
    } while (0)
#define DISPATCH1(kern, ...)                                                                        \
    do {                                                                                            \
        if (box_empty(a)) return;                                                                   \
        dim3 g = box_grid(a);                                                                       \
        if (a.var_elem_bytes == 4) kern<float><<<g, 256, 0, s>>>(__VA_ARGS__);                      \
        else kern<double><<<g, 256, 0, s>>>(__VA_ARGS__);                                           \
        YKH_HIP(hipGetLastError());                                                                 \
    } while (0)

void launch_box_gather(const BoxCopyArgs& a, hipStream_t s) { DISPATCH2(box_gather_k, a); }
void launch_box_scatter(const BoxCopyArgs& a, hipStream_t s) { DISPATCH2(box_scatter_k, a); }
void launch_box_fill(const BoxCopyArgs& a, double v, hipStream_t s) { DISPATCH1(box_fill_k, a, v); }
void launch_box_add(const BoxCopyArgs& a, double v, hipStream_t s) { DISPATCH1(box_add_k, a, v); }
void launch_box_hash(const BoxCopyArgs& a, double offset, double scale, idx_t vid, idx_t slot, idx_t gx,
                     idx_t gy, idx_t gz, hipStream_t s) {
    DISPATCH1(box_hash_k, a, offset, scale, vid, slot, gx, gy, gz);
}

// ---- reduction: one block-level partial per block, combined with double atomics (few blocks).
__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_prod(double v) {
    for (int o = 32; o > 0; o >>= 1) v *= __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ void atomic_mul_f64(double* p, double v) {
    unsigned long long* ip = (unsigned long long*)p;
    unsigned long long old = *ip, assumed;
    do { assumed = old; old = atomicCAS(ip, assumed, __double_as_longlong(__longlong_as_double(assumed) * v)); } while (assumed != old);
}
__device__ __forceinline__ void atomic_max_f64(double* p, double v) {
    unsigned long long* ip = (unsigned long long*)p;
    unsigned long long old = *ip, assumed;
    do { assumed = old; if (__longlong_as_double(assumed) >= v) break; old = atomicCAS(ip, assumed, __double_as_longlong(v)); } while (assumed != old);
}
__device__ __forceinline__ void atomic_min_f64(double* p, double v) {
    unsigned long long* ip = (unsigned long long*)p;
    unsigned long long old = *ip, assumed;
    do { assumed = old; if (__longlong_as_double(assumed) <= v) break; old = atomicCAS(ip, assumed, __double_as_longlong(v)); } while (assumed != old);
}

template <typename TV>
__global__ void box_reduce_k(const BoxCopyArgs a, double* out) {
    double s = 0, s2 = 0, pr = 1, mx = -INFINITY, mn = INFINITY;
    const idx_t k = (idx_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < a.n[2])
        for (idx_t i = blockIdx.z; i < a.n[0]; i += gridDim.z)
            for (idx_t j = blockIdx.y; j < a.n[1]; j += gridDim.y) {
                double v = (double)*box_elem<TV>(a, i, j, k);
                s += v; s2 += v * v; pr *= v; mx = fmax(mx, v); mn = fmin(mn, v);
            }
    s = wave_sum(s); s2 = wave_sum(s2); pr = wave_prod(pr); mx = wave_max(mx); mn = wave_min(mn);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&out[0], s); atomicAdd(&out[1], s2);
        atomic_mul_f64(&out[2], pr); atomic_max_f64(&out[3], mx); atomic_min_f64(&out[4], mn);
    }
}
void launch_box_reduce(const BoxCopyArgs& a, double* out5, hipStream_t s) {
    double init[5] = {0.0, 0.0, 1.0, -INFINITY, INFINITY};
    YKH_HIP(hipMemcpyAsync(out5, init, sizeof(init), hipMemcpyHostToDevice, s));
    YKH_HIP(hipStreamSynchronize(s));   // `init` is a stack temporary
    DISPATCH1(box_reduce_k, a, out5);
}

template <typename TV>
__global__ void box_compare_k(const BoxCopyArgs a, const BoxCopyArgs b, double eps, unsigned long long* count) {
    unsigned long long bad = 0;
    const idx_t k = (idx_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < a.n[2])
        for (idx_t i = blockIdx.z; i < a.n[0]; i += gridDim.z)
            for (idx_t j = blockIdx.y; j < a.n[1]; j += gridDim.y) {
                double v = (double)*box_elem<TV>(a, i, j, k);
                double r = (double)*box_elem<TV>(b, i, j, k);
                // within_tolerance (realv.hpp:979-994): relative when |ref| > 1, else absolute
                double d = fabs(v - r);
                if (fabs(r) > 1.0) d /= fabs(r);
                bool ok = (v == r) || (d <= eps);
                if (isnan(v) != isnan(r)) ok = false;
                bad += ok ? 0 : 1;
            }
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_down(bad, o, 64);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}
void launch_box_compare(const BoxCopyArgs& a, const BoxCopyArgs& b, double eps, unsigned long long* count,
                        hipStream_t s) {
    DISPATCH1(box_compare_k, a, b, eps, count);
}


// ------------------------------------------------------------------ pads of a step slot -> another slot
// Copies every allocated element of a dense [a0][a1][a2] slot that lies OUTSIDE the domain box (the pads / halos) from
// src to dst; rows inside the domain in x and y only have their two z pads copied.  Used by Solution::run_fused(): the
// scratch slot a fused pass writes into must carry the pads of the slot whose role it takes.
template <typename T>
__global__ void __launch_bounds__(256) copy_pads_k(const T* __restrict__ src, T* __restrict__ dst, idx_t a1, idx_t a2,
                                                   idx_t p0, idx_t p1, idx_t p2, idx_t n0, idx_t n1, idx_t n2) {
    const idx_t i = blockIdx.y, j = blockIdx.x;
    const bool inside_xy = i >= p0 && i < p0 + n0 && j >= p1 && j < p1 + n1;
    const idx_t row = (i * a1 + j) * a2;
    if (!inside_xy) {
        for (idx_t k = threadIdx.x; k < a2; k += blockDim.x) dst[row + k] = src[row + k];
    } else {
        for (idx_t k = threadIdx.x; k < p2; k += blockDim.x) dst[row + k] = src[row + k];
        for (idx_t k = p2 + n2 + threadIdx.x; k < a2; k += blockDim.x) dst[row + k] = src[row + k];
    }
}
void launch_copy_pads(const void* src, void* dst, int elem_bytes, const idx_t alloc[3], const idx_t pad_l[3], const idx_t dom[3],
                      hipStream_t s) {
    dim3 grid((unsigned)alloc[1], (unsigned)alloc[0]);
    if (elem_bytes == 4)
        hipLaunchKernelGGL(copy_pads_k<float>, grid, dim3(256), 0, s, (const float*)src, (float*)dst, alloc[1], alloc[2], pad_l[0], pad_l[1],
                           pad_l[2], dom[0], dom[1], dom[2]);
    else
        hipLaunchKernelGGL(copy_pads_k<double>, grid, dim3(256), 0, s, (const double*)src, (double*)dst, alloc[1], alloc[2], pad_l[0], pad_l[1],
                           pad_l[2], dom[0], dom[1], dom[2]);
}

// ------------------------------------------------------------------ streaming-bandwidth probe
// What this device delivers right now to a 16-byte-per-lane streaming kernel with the stencil's read:write mix --
// printed by bench.py next to the roofline fraction (a box in a low-power state or with slow HBM shows up here).
typedef float f4 __attribute__((ext_vector_type(4)));
// Round 4: the first version kept ONE 16-byte load per lane in flight per array (load, wait, store, next)
// and read 5.0 TB/s for a copy on a box where the stencil kernel itself moved 6.27 TB/s -- a probe slower than the kernel it is
// meant to put into perspective.  Now every lane issues UNR independent 16-byte loads per array (one-touch: non-temporal)
// before it touches any of them, i.e. 2048 workgroups x 256 lanes x UNR x 16 B = 128 KiB in flight per CU and array, and
// stores non-temporally.
template <int KIND, int UNR>
__global__ void __launch_bounds__(256) bw_probe_k(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c,
                                                  f4* __restrict__ d, size_t n) {
    const size_t chunk = (size_t)blockDim.x * UNR, stride = (size_t)gridDim.x * chunk;
    f4 acc = f4(0.f);
    for (size_t i0 = (size_t)blockIdx.x * chunk + threadIdx.x; i0 < n; i0 += stride) {
        f4 x[UNR], y[UNR], z[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const size_t i = i0 + (size_t)u * blockDim.x;
            if (i < n) {
                x[u] = __builtin_nontemporal_load(&a[i]);
                if constexpr (KIND == 1) { y[u] = __builtin_nontemporal_load(&b[i]); z[u] = __builtin_nontemporal_load(&c[i]); }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const size_t i = i0 + (size_t)u * blockDim.x;
            if (i < n) {
                if constexpr (KIND == 0) __builtin_nontemporal_store(x[u], &d[i]);
                else if constexpr (KIND == 1) __builtin_nontemporal_store(x[u] + y[u] * z[u], &d[i]);
                else acc += x[u];
            }
        }
    }
    if constexpr (KIND == 2) if (acc.x == 123.456f) d[0] = acc;
}
// ------------------------------------------------------------------ halo pack / unpack: all slabs of an exchange in one launch
// Register budget: <= 16 VGPRs.  The marching twins that run a decomposed rank's launches use 242-248 VGPRs, which leaves 16 per lane of
// every SIMD: a wave of a 16-VGPR kernel runs BESIDE a resident marching workgroup, one of 24 VGPRs waits until the launch is over
// (tools/microbench/waiter_cost.hip: 1024 workgroups done after 0.024 ms vs after the hogs' 1.0 ms).  So: one segment per blockIdx.y
// (no search, the segment's numbers are scalars), 32-bit indices inside a segment, byte strides prepared on the host.
constexpr int HALO_SEGS = 40;
struct HaloSegDev {
    char* var0; char* buf;           // var address of the slab's first element; packed buffer of the segment
    long long sxb, syb, sub;         // byte strides of a plane, a row, and a UNIT (vec elements) in the var
    unsigned n1, n2u;                // rows per plane; units per row (n2 / vec)
    unsigned units;                  // units in the segment
    unsigned unit_bytes;             // 16, or the element size where a slab's rows cannot be moved as vectors
};
struct HaloMoveArgs { int nseg; HaloSegDev seg[HALO_SEGS]; };

template <bool PACK>
__global__ void __launch_bounds__(256) halo_move_k(const HaloMoveArgs a) {
    const HaloSegDev& g = a.seg[blockIdx.y];
    const unsigned n2u = g.n2u, n1 = g.n1, units = g.units, ub = g.unit_bytes;
    for (unsigned r = blockIdx.x * 256u + threadIdx.x; r < units; r += gridDim.x * 256u) {
        const unsigned rows = r / n2u, kv = r - rows * n2u;
        const unsigned i = rows / n1, j = rows - i * n1;
        char* vp = g.var0 + ((long long)i * g.sxb + (long long)j * g.syb + (long long)kv * g.sub);
        char* bp = g.buf + (unsigned long long)r * ub;
        if (ub == 16) {
            if (PACK) *reinterpret_cast<uint4*>(bp) = *reinterpret_cast<const uint4*>(vp);
            else *reinterpret_cast<uint4*>(vp) = *reinterpret_cast<const uint4*>(bp);
        } else if (ub == 4) {
            if (PACK) *reinterpret_cast<unsigned*>(bp) = *reinterpret_cast<const unsigned*>(vp);
            else *reinterpret_cast<unsigned*>(vp) = *reinterpret_cast<const unsigned*>(bp);
        } else {
            if (PACK) *reinterpret_cast<unsigned long long*>(bp) = *reinterpret_cast<const unsigned long long*>(vp);
            else *reinterpret_cast<unsigned long long*>(vp) = *reinterpret_cast<const unsigned long long*>(bp);
        }
    }
}

void launch_halo_move(const std::vector<HaloSeg>& segs, bool pack, int elem_bytes, hipStream_t st) {
    if (elem_bytes != 4 && elem_bytes != 8) YKH_THROW("halo pack: element size must be 4 or 8 bytes");
    const unsigned V = 16u / (unsigned)elem_bytes;
    size_t k = 0;
    while (k < segs.size()) {
        HaloMoveArgs a;
        a.nseg = 0;
        unsigned long long most = 0;
        for (; k < segs.size() && a.nseg < HALO_SEGS; k++) {
            const HaloSeg& h = segs[k];
            if (h.n[0] <= 0 || h.n[1] <= 0 || h.n[2] <= 0) continue;
            HaloSegDev& d = a.seg[a.nseg];
            // 16-byte units where every row of the slab starts and ends on a vector boundary on both sides
            const bool vec_ok = h.sz == 1 && h.n[2] % (int)V == 0 && (((long long)h.lo[2]) % (long long)V + V) % V == 0 &&
                                h.sy % V == 0 && h.sx % V == 0 && ((uintptr_t)h.var_base % 16) == 0 && ((uintptr_t)h.buf % 16) == 0;
            const unsigned vec = vec_ok ? V : 1;
            d.var0 = (char*)h.var_base + ((long long)h.lo[0] * h.sx + (long long)h.lo[1] * h.sy + (long long)h.lo[2] * h.sz) * elem_bytes;
            d.buf = (char*)h.buf;
            d.sxb = h.sx * elem_bytes; d.syb = h.sy * elem_bytes; d.sub = h.sz * (long long)vec * elem_bytes;
            d.n1 = (unsigned)h.n[1];
            d.n2u = (unsigned)h.n[2] / vec;
            const unsigned long long units = (unsigned long long)h.n[0] * d.n1 * d.n2u;
            if (units >= (1ull << 32)) YKH_THROW("halo pack: a slab of more than 2^32 units");
            d.units = (unsigned)units;
            d.unit_bytes = vec * (unsigned)elem_bytes;
            most = std::max(most, units);
            a.nseg++;
        }
        if (a.nseg == 0 || most == 0) continue;
        const unsigned bx = (unsigned)std::min<unsigned long long>((most + 255) / 256, 4096);       // (grid-strided beyond that)
        if (pack) hipLaunchKernelGGL(halo_move_k<true>, dim3(bx, (unsigned)a.nseg), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(halo_move_k<false>, dim3(bx, (unsigned)a.nseg), dim3(256), 0, st, a);
        YKH_HIP(hipGetLastError());
    }
}

// ------------------------------------------------------------------ the mirror transport's stand-in for a copy engine + a link
// linear_copy_k: a device-to-device copy in <= 16 VGPRs, so that (like an SDMA engine, unlike the runtime's blit kernels) it runs
// BESIDE resident marching workgroups and takes bandwidth, not CUs.  hold_until_k: one wave that ends `ticks` (100 MHz) after the
// first copy of the exchange started -- the exchange then lasts what a link of the given speed would take for its largest message
// (a rank's faces travel on different links at the same time).  Instrument only (yk_env_init_mirror, tools/overlap_probe.py).
__global__ void __launch_bounds__(256) linear_copy_k(uint4* d, const uint4* s, unsigned n16, unsigned* d4, const unsigned* s4, unsigned n4, unsigned long long* t0) {
    if (t0 && blockIdx.x == 0 && threadIdx.x == 0) (void)atomicCAS(t0, 0ull, (unsigned long long)wall_clock64());
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) d[i] = s[i];
    if (blockIdx.x == 0 && threadIdx.x < n4) d4[threadIdx.x] = s4[threadIdx.x];      // (a tail of < 16 bytes, in 4-byte words)
}
__global__ void __launch_bounds__(64) hold_until_k(unsigned long long* t0, unsigned long long ticks) {
    if (threadIdx.x == 0) {
        const unsigned long long start = __hip_atomic_load(t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (start) while ((unsigned long long)wall_clock64() - start < ticks) __builtin_amdgcn_s_sleep(16);
        __hip_atomic_store(t0, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
void launch_linear_copy(void* dst, const void* src, size_t bytes, unsigned long long* t0, hipStream_t st) {
    if (!bytes) return;
    if (bytes % 4 || ((uintptr_t)dst | (uintptr_t)src) % 16 || bytes >= (1ull << 35)) YKH_THROW("linear copy: unaligned or oversized message");
    const unsigned n16 = (unsigned)(bytes / 16), n4 = (unsigned)((bytes % 16) / 4);
    const unsigned blocks = std::max(1u, std::min((n16 + 255u) / 256u, 2048u));
    hipLaunchKernelGGL(linear_copy_k, dim3(blocks), dim3(256), 0, st, (uint4*)dst, (const uint4*)src, n16, (unsigned*)((char*)dst + (size_t)n16 * 16),
                       (const unsigned*)((const char*)src + (size_t)n16 * 16), n4, t0);
    YKH_HIP(hipGetLastError());
}
void launch_hold_until(unsigned long long* t0, double seconds, hipStream_t st) {
    hipLaunchKernelGGL(hold_until_k, dim3(1), dim3(64), 0, st, t0, (unsigned long long)(std::max(0.0, seconds) * 1.0e8));
    YKH_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ stream-ordered waits on words in device memory
// wait_words_kernel: ONE wave; lane i polls word i until it has reached its value (wrap-safe `>=`), then the kernel ends and the
// stream goes on.  Used (a) by the comm stream to wait for the shell blocks of a planned launch that is still running on the
// compute stream (Solution::launch_planned: block_done() publishes the epoch), and (b) by the IPC transport for the flags its
// peers write into this rank's mailbox (ykh_ipc.cpp).  It occupies one wave slot and issues one 4-byte system-scope load per
// lane and microsecond: nothing a marching kernel notices.  A waiter that is never released would hang the stream (and, with
// it, the box): after `spins` polls (~1 us each) it gives up, raises *err and lets the stream continue -- the host reports
// the error at its next synchronisation point.
struct WaitWords { const unsigned* p[32]; unsigned v[32]; int n; };
__global__ void __launch_bounds__(64) wait_words_kernel(WaitWords w, unsigned* err, unsigned spins) {
    const int i = threadIdx.x;
    bool ok = i >= w.n;
    // a waiter before this one has already given up (the error word stays raised until the host has seen it): the exchange is lost,
    // do not make the stream sit through one time-out per remaining wait of the run -- a failing run costs ONE time-out
    if (err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
    for (unsigned k = 0; k < spins; k++) {
        if (!ok) ok = (int)(__hip_atomic_load(w.p[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - w.v[i]) >= 0;
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(32);
    }
    if (!ok && err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // what was released before the words were raised is visible to what follows
}
void launch_wait_words(int n, const unsigned* const* ptrs, const unsigned* vals, unsigned* err, double timeout_s, hipStream_t s) {
    if (n <= 0) return;
    for (int base = 0; base < n; base += 32) {
        WaitWords w;
        w.n = std::min(32, n - base);
        for (int i = 0; i < 32; i++) { w.p[i] = ptrs[base + std::min(i, w.n - 1)]; w.v[i] = vals[base + std::min(i, w.n - 1)]; }
        const unsigned spins = (unsigned)std::min(4.0e9, std::max(1.0e3, timeout_s * 1.0e6));
        hipLaunchKernelGGL(wait_words_kernel, dim3(1), dim3(64), 0, s, w, err, spins);
    }
    YKH_HIP(hipGetLastError());
}
// set_words_kernel: lane i stores value i to word i with system scope, after a system-scope release (what this stream did before
// -- a copy into a peer's buffer -- is visible to whoever sees the word).
struct SetWords { unsigned* p[32]; unsigned v[32]; int n; };
__global__ void __launch_bounds__(64) set_words_kernel(SetWords w) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int i = threadIdx.x;
    if (i < w.n) __hip_atomic_store(w.p[i], w.v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_set_words(int n, unsigned* const* ptrs, const unsigned* vals, hipStream_t s) {
    for (int base = 0; base < n; base += 32) {
        SetWords w;
        w.n = std::min(32, n - base);
        for (int i = 0; i < 32; i++) { w.p[i] = ptrs[base + std::min(i, w.n - 1)]; w.v[i] = vals[base + std::min(i, w.n - 1)]; }
        hipLaunchKernelGGL(set_words_kernel, dim3(1), dim3(64), 0, s, w);
    }
    YKH_HIP(hipGetLastError());
}

double probe_bandwidth(int kind, size_t bytes, int reps) {
    if (kind < 0 || kind > 2) YKH_THROW("probe_bandwidth: kind must be 0 (copy), 1 (3 reads + 1 write) or 2 (read)");
    if (bytes < (1u << 20)) bytes = (size_t)1 << 30;
    if (reps < 1) reps = 3;
    bytes &= ~(size_t)4095;
    const int narr = kind == 0 ? 2 : (kind == 1 ? 4 : 1);
    struct Res {
        void* p[4] = {nullptr, nullptr, nullptr, nullptr}; hipStream_t st = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Res() { for (auto q : p) if (q) (void)hipFree(q); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
                 if (st) (void)hipStreamDestroy(st); }
    } r;
    YKH_HIP(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));
    YKH_HIP(hipEventCreate(&r.e0));
    YKH_HIP(hipEventCreate(&r.e1));
    for (int i = 0; i < narr; i++) { YKH_HIP(hipMalloc(&r.p[i], bytes)); YKH_HIP(hipMemsetAsync(r.p[i], 0, bytes, r.st)); }
    // kind 0: a -> d = p[1]; kind 1: a,b,c = p[0..2] -> d = p[3]; kind 2: a = p[0], d = p[0] (never written)
    const f4 *a = (const f4*)r.p[0], *b = (const f4*)r.p[1], *c = (const f4*)r.p[2];
    f4* d = (f4*)(kind == 0 ? r.p[1] : (kind == 1 ? r.p[3] : r.p[0]));
    const size_t n = bytes / sizeof(f4);
    int cus = 256;
    hipDeviceProp_t prop; int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    const dim3 grid((unsigned)(cus * 8)), block(256);
    double best = 0;
    for (int it = 0; it < reps + 1; it++) {
        YKH_HIP(hipEventRecord(r.e0, r.st));
        if (kind == 0) hipLaunchKernelGGL((bw_probe_k<0, 4>), grid, block, 0, r.st, a, b, c, d, n);
        else if (kind == 1) hipLaunchKernelGGL((bw_probe_k<1, 4>), grid, block, 0, r.st, a, b, c, d, n);
        else hipLaunchKernelGGL((bw_probe_k<2, 4>), grid, block, 0, r.st, a, b, c, d, n);
        YKH_HIP(hipGetLastError());
        YKH_HIP(hipEventRecord(r.e1, r.st));
        YKH_HIP(hipEventSynchronize(r.e1));
        float ms = 0;
        YKH_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
        if (it > 0 && ms > 0) best = std::max(best, (double)narr * (double)bytes / (ms * 1e-3) * 1e-9);
    }
    return best;
}

}  // namespace ykh
