// mall_bw.hip -- how fast can the CUs stream a buffer that fits the 256 MiB Infinity Cache, compared with one that
// does not?  (Decides whether blocking two time steps through the Infinity Cache can beat the HBM roofline,
// DESIGN.md section 3.7.)  Build: hipcc -O3 --offload-arch=gfx950 mall_bw.hip -o mall_bw ; run: ./mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(512) read_k(const float4* __restrict__ p, size_t n4, float* out) {
    float4 acc = {0, 0, 0, 0};
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride * 4) {
        float4 a = p[i];
        float4 b = i + stride < n4 ? p[i + stride] : acc;
        float4 c = i + 2 * stride < n4 ? p[i + 2 * stride] : acc;
        float4 d = i + 3 * stride < n4 ? p[i + 3 * stride] : acc;
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
        acc.z += a.z + b.z + c.z + d.z; acc.w += a.w + b.w + c.w + d.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}
__global__ void __launch_bounds__(512) copy_k(const float4* __restrict__ p, float4* __restrict__ q, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) q[i] = p[i];
}

int main() {
    const size_t maxb = (size_t)8 << 30;
    float4 *a, *b; float* out;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&b, maxb)); CK(hipMalloc(&out, 64));
    CK(hipMemset(a, 0, maxb)); CK(hipMemset(b, 0, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes_mb[] = {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024, 4096, 8192};
    printf("%10s %14s %14s\n", "MiB", "read GB/s", "copy GB/s (r+w)");
    for (size_t mb : sizes_mb) {
        size_t bytes = mb << 20, n4 = bytes / 16;
        int reps = (int)((size_t)(64ull << 30) / bytes); if (reps > 400) reps = 400; if (reps < 4) reps = 4;
        float ms_r, ms_c;
        for (int w = 0; w < 3; w++) read_k<<<256 * 4, 512>>>(a, n4, out);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) read_k<<<256 * 4, 512>>>(a, n4, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_r, e0, e1));
        size_t half = n4 / 2;   // copy within the same footprint: read first half, write second half
        for (int w = 0; w < 3; w++) copy_k<<<256 * 8, 512>>>(a, a + half, half);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) copy_k<<<256 * 8, 512>>>(a, a + half, half);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_c, e0, e1));
        printf("%10zu %14.0f %14.0f\n", mb, (double)bytes * reps / ms_r * 1e-6, (double)bytes * reps / ms_c * 1e-6);
    }
    return 0;
}
