// waiter_cost.hip -- what does a wave that waits on a word cost a launch that needs every register of every CU?  
//
// The marching twins run 512-thread workgroups at 247-256 VGPRs: 2 waves per SIMD fill the 512-entry register file, one workgroup
// fills a CU.  A plan of 256 equal blocks then takes ONE round -- unless something else holds a slot.  This probe launches 256 (and
// 512) "hog" workgroups that spin for a fixed time at a fixed register count and times the launch
//   (a) alone,
//   (b) with a one-wave kernel polling a word on another stream (wait_words_kernel's method), resident before the launch,
//   (c) with a hipStreamWaitValue32() pending on another stream (signal memory; does the runtime wait on a CU or in the command processor?),
//   (d) with a hipStreamWaitEvent() pending on another stream (known to be free).
// The word is released afterwards (host store / hipStreamWriteValue32) and the waiting stream's follow-up kernel must then run.
// Build: hipcc -O3 --offload-arch=gfx950 waiter_cost.hip -o waiter_cost ; run: ./waiter_cost [spin_us]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NV>
__global__ void __launch_bounds__(512) hog(unsigned long long ticks, unsigned* sink) {
    if (NV >= 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    else if (NV >= 240) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    else asm volatile("v_mov_b32 v100, 0" ::: "v100");
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0xffffffffu) *sink = 1;
}
__global__ void __launch_bounds__(64) poll_word(const unsigned* w, unsigned v, unsigned spins, unsigned* out) {
    bool ok = false;
    for (unsigned k = 0; k < spins && !ok; k++) {
        ok = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= v;
        if (!ok) __builtin_amdgcn_s_sleep(32);
    }
    if (threadIdx.x == 0) *out = ok ? 1u : 2u;
}
__global__ void mark(unsigned* out, unsigned v) { *out = v; }
// a pack-like kernel: 256-thread workgroups (one wave per SIMD) at a given VGPR count, each busy for a few microseconds
template <int NV>
__global__ void __launch_bounds__(256) small(unsigned long long ticks) {
    if (NV > 16) asm volatile("v_mov_b32 v23, 0" ::: "v23");
    else if (NV > 8) asm volatile("v_mov_b32 v15, 0" ::: "v15");
    else asm volatile("v_mov_b32 v7, 0" ::: "v7");
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}
template <int NV> __global__ void __launch_bounds__(512) hog2(unsigned long long ticks) {
    if (NV >= 248) asm volatile("v_mov_b32 v247, 0" ::: "v247");
    else asm volatile("v_mov_b32 v239, 0" ::: "v239");
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
// does a small-register kernel on another stream run BESIDE resident hog workgroups (done in microseconds) or after them?
template <int HV, int SV>
static int coreside(int ncu) {
    hipStream_t cs, ws;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ws, hipStreamNonBlocking));
    hipEvent_t e0, e1, h1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&h1));
    hipLaunchKernelGGL(small<SV>, dim3(1024), dim3(256), 0, ws, 500ull);     // warm-up
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(hog2<HV>, dim3(ncu), dim3(512), 0, cs, 100000ull);      // 1000 us on every CU
    CK(hipEventRecord(h1, cs));
    CK(hipEventRecord(e0, ws));
    hipLaunchKernelGGL(small<SV>, dim3(1024), dim3(256), 0, ws, 500ull);     // 1024 workgroups x 5 us
    CK(hipEventRecord(e1, ws));
    CK(hipDeviceSynchronize());
    float ms = 0, hm = 0;
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&hm, e0, h1));
    printf("{\"hog_vgprs\": %d, \"small_kernel_vgprs\": %d, \"small_kernel_done_after_ms\": %.4f, \"hogs_done_after_ms\": %.4f, \"ran_beside_the_hogs\": %s}\n", HV, SV, ms, hm,
           ms < 0.6f * hm ? "true" : "false");
    fflush(stdout);
    return 0;
}

template <int NV>
static float time_hogs(int blocks, unsigned long long ticks, hipStream_t s, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(hog<NV>, dim3(blocks), dim3(512), 0, s, ticks, nullptr);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(hog<NV>, dim3(blocks), dim3(512), 0, s, ticks, nullptr);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

template <int NV>
static int run_case(const char* name, int ncu, unsigned long long ticks) {
    hipStream_t cs, ws, rs;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ws, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&rs, hipStreamNonBlocking));
    unsigned *dev_word, *out, *sigmem = nullptr;
    CK(hipMalloc(&dev_word, 64)); CK(hipMemset(dev_word, 0, 64));
    CK(hipHostMalloc(&out, 64, hipHostMallocMapped)); out[0] = out[1] = out[2] = 0;
    const int reps = 10;
    const float a1 = time_hogs<NV>(ncu, ticks, cs, reps), a2 = time_hogs<NV>(2 * ncu, ticks, cs, reps);
    // (b) polling wave
    hipLaunchKernelGGL(poll_word, dim3(1), dim3(64), 0, ws, dev_word, 1u, 20000000u, out);
    const float b1 = time_hogs<NV>(ncu, ticks, cs, reps), b2 = time_hogs<NV>(2 * ncu, ticks, cs, reps);
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, rs, dev_word, 1u);
    CK(hipStreamSynchronize(rs)); CK(hipStreamSynchronize(ws));
    // (c) hipStreamWaitValue32 on signal memory
    int can = 0;
    (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    float c1 = -1, c2 = -1;
    unsigned c_state = 0;
    if (can && hipExtMallocWithFlags((void**)&sigmem, 8, hipMallocSignalMemory) == hipSuccess) {
        *(volatile unsigned long long*)sigmem = 0;
        hipError_t e = hipStreamWaitValue32(ws, sigmem, 1, hipStreamWaitValueGte, 0xFFFFFFFFu);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, ws, out + 1, 7u);
            c1 = time_hogs<NV>(ncu, ticks, cs, reps); c2 = time_hogs<NV>(2 * ncu, ticks, cs, reps);
            const unsigned before = ((volatile unsigned*)out)[1];
            CK(hipStreamWriteValue32(rs, sigmem, 1, 0));
            CK(hipStreamSynchronize(rs)); CK(hipStreamSynchronize(ws));
            c_state = before * 100 + ((volatile unsigned*)out)[1];     // 7 = the follow-up ran only after the release
        } else { printf("  hipStreamWaitValue32: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); }
    } else (void)hipGetLastError();
    // (d) hipStreamWaitEvent on an event that is recorded later
    hipEvent_t gate;
    CK(hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    hipLaunchKernelGGL(poll_word, dim3(1), dim3(64), 0, rs, dev_word + 1, 1u, 20000000u, out + 2);   // keeps rs busy until released
    CK(hipEventRecord(gate, rs));
    CK(hipStreamWaitEvent(ws, gate, 0));
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, ws, out + 3, 9u);
    CK(hipMemset(dev_word + 1, 0xff, 4));       // release rs right away: the wait on ws is then a resolved dependency...
    CK(hipStreamSynchronize(rs));
    const float d1 = time_hogs<NV>(ncu, ticks, cs, reps), d2 = time_hogs<NV>(2 * ncu, ticks, cs, reps);
    CK(hipStreamSynchronize(ws));
    printf("{\"hog\": \"%s\", \"blocks\": [%d, %d], \"alone_ms\": [%.4f, %.4f], \"polling_wave_resident_ms\": [%.4f, %.4f], \"polling_wave_result\": %u,\n"
           " \"can_stream_wait_value\": %d, \"stream_wait_value_pending_ms\": [%.4f, %.4f], \"stream_wait_value_followup_before_after\": %u, \"after_event_wait_ms\": [%.4f, %.4f]}\n",
           name, ncu, 2 * ncu, a1, a2, b1, b2, out[0], can, c1, c2, c_state, d1, d2);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const double us = argc > 1 ? atof(argv[1]) : 200.0;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    const unsigned long long ticks = (unsigned long long)(us * 100.0);      // wall_clock64(): 100 MHz
    printf("# %s, %d CUs, hog workgroups spin %.0f us\n", p.name, ncu, us);
    if (run_case<256>("512 threads x 256 VGPRs (the _tl twin)", ncu, ticks)) return 1;
    if (run_case<240>("512 threads x 240 VGPRs", ncu, ticks)) return 1;
    if (run_case<101>("512 threads x 101 VGPRs", ncu, ticks)) return 1;
    if (coreside<248, 8>(ncu) || coreside<248, 16>(ncu) || coreside<248, 24>(ncu) || coreside<240, 16>(ncu) || coreside<240, 24>(ncu)) return 1;
    return 0;
}
