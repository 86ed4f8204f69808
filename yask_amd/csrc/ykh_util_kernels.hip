// ykh_util_kernels.hip -- small data-movement kernels around the stencil sweep.
//
// GPU counterparts of the reference's device regions D3 (halo pack/unpack and slice copies,
// YkVarBase::_copy_vecs_in_slice, src/kernel/lib/yk_var.hpp:1900-2040) and D4 (var init,
// src/kernel/lib/generic_var.cpp:137-200), plus reductions (yk_var::reduce_elements_in_slice,
// src/kernel/lib/yk_var_apis.cpp) and the ref-vs-opt compare (YkVarBase::compare,
// src/kernel/lib/yk_var.cpp:401-477; tolerance rule src/kernel/lib/realv.hpp:974-994).
// All are HBM-bound copies: threads run along z (unit stride) so accesses coalesce.
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

}  // namespace ykh
