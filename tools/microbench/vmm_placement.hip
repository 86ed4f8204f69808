// vmm_placement.hip -- does the "placement lottery" follow PHYSICAL placement, and can the virtual-memory API control it?
//
// the same stencil kernel runs 3-4 % apart on different sets of freshly hipMalloc'ed arrays (each set
// stable to 0.1 %); prepare_solution() therefore draws several sets and keeps the fastest (-hip_placement_trials).  VERDICT r02
// weak #5 calls that a workaround and names the untried alternative: hipMemAddressReserve / hipMemCreate / hipMemMap, i.e.
// choosing which physical chunk backs which part of which array.  This microbenchmark asks, with the stencil's own memory
// pattern and no arithmetic (the level-1 sweep of mall_pipeline.hip: 256 workgroups, one 16 KB tile-plane per array per
// iteration, three arrays read, one of them written in place, two planes of loads in flight):
//   A  hipMalloc, K fresh sets held at the same time           -> does this kernel show the spread at all?
//   B  VMM, one physical handle per array                       -> the same thing through the other API
//   C  VMM, physical chunks of `chunk` MiB created array by array, mapped in creation order
//   D  VMM, chunks created round-robin over the arrays (a0 b0 c0 a1 b1 c1 ...): equal logical positions are physical neighbours
//   E  the chunks of C, array b re-mapped rotated by r chunks, r = 1 .. R: the RELATIVE physical offset of the streams changes,
//      nothing else does -> if the time follows r, placement can be chosen instead of drawn
// Every variant: `reps` sweeps after one warm-up, best and median ms per sweep.
//
// Build: hipcc -O3 --offload-arch=gfx950 vmm_placement.hip -o vmm_placement ; run: ./vmm_placement [planes=1024] [chunk_MiB=64] [K=6] [R=8]
// (written at the end of round 3, compiled, NOT yet run on a GPU: the round's GPU budget was spent)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d: %s\n", hipGetErrorString(e_), __LINE__, #x); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NT = 512;                 // threads per workgroup
constexpr int TILE_F4 = 1024;           // 16 KB per tile-plane per array
constexpr int TILES = 256;              // tiles per plane = workgroups
constexpr int PD = 2;                   // planes of loads in flight

// one sweep: b[x] = f(a[x], b[x], c[x]) plane by plane, every workgroup its own tile -- the stencil kernel's streams
__global__ void __launch_bounds__(NT) sweep_k(const f4* __restrict__ a, f4* __restrict__ b, const f4* __restrict__ c, int nx) {
    const size_t plane = (size_t)TILES * TILE_F4;
    const size_t t0 = (size_t)blockIdx.x * TILE_F4 + threadIdx.x;
    f4 ra[PD][2], rb[PD][2], rc[PD][2];
    auto load = [&](int x, int s) {
        if (x >= nx) return;
        const size_t o = (size_t)x * plane + t0;
        ra[s][0] = a[o]; ra[s][1] = a[o + NT];
        rb[s][0] = __builtin_nontemporal_load(&b[o]); rb[s][1] = __builtin_nontemporal_load(&b[o + NT]);
        rc[s][0] = __builtin_nontemporal_load(&c[o]); rc[s][1] = __builtin_nontemporal_load(&c[o + NT]);
    };
    for (int s = 0; s < PD; s++) load(s, s);
    for (int x = 0; x < nx; x += PD) {
#pragma unroll
        for (int s = 0; s < PD; s++) {
            if (x + s >= nx) break;
            const size_t o = (size_t)(x + s) * plane + t0;
            const f4 v0 = ra[s][0] + rb[s][0] * rc[s][0], v1 = ra[s][1] + rb[s][1] * rc[s][1];
            load(x + s + PD, s);
            __builtin_nontemporal_store(v0, &b[o]);
            __builtin_nontemporal_store(v1, &b[o + NT]);
        }
    }
}

struct Timing { float best, median; };
static int time_sweeps(const f4* a, f4* b, const f4* c, int nx, int reps, Timing* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int r = -1; r < reps; r++) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(sweep_k, dim3(TILES), dim3(NT), 0, 0, a, b, c, nx);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float m = 0;
        CK(hipEventElapsedTime(&m, e0, e1));
        if (r >= 0) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    out->best = ms.front(); out->median = ms[ms.size() / 2];
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

// three arrays of `bytes` each in ONE reserved virtual range, backed by physical chunks the caller orders
struct Vmm {
    char* va = nullptr;
    size_t bytes = 0, chunk = 0, nchunk = 0;            // per array
    std::vector<hipMemGenericAllocationHandle_t> h;     // [array][chunk] in CREATION order as given by `order`
    hipMemAllocationProp prop{};
    f4* arr(int i) const { return (f4*)(va + (size_t)i * bytes); }
};
// order: 0 = array by array, 1 = round-robin over the arrays
static int vmm_create(Vmm& v, int dev, size_t bytes, size_t chunk, int order) {
    v.prop = hipMemAllocationProp{};
    v.prop.type = hipMemAllocationTypePinned;
    v.prop.location.type = hipMemLocationTypeDevice;
    v.prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &v.prop, hipMemAllocationGranularityRecommended));
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    if (bytes % chunk) { printf("array size %zu is not a multiple of the chunk %zu\n", bytes, chunk); return 1; }
    v.bytes = bytes; v.chunk = chunk; v.nchunk = bytes / chunk;
    v.h.assign(3 * v.nchunk, hipMemGenericAllocationHandle_t{});
    CK(hipMemAddressReserve((void**)&v.va, 3 * bytes, chunk, nullptr, 0));
    if (order == 0) {
        for (size_t i = 0; i < 3 * v.nchunk; i++) CK(hipMemCreate(&v.h[i], chunk, &v.prop, 0));
    } else {
        for (size_t k = 0; k < v.nchunk; k++)
            for (int a = 0; a < 3; a++) CK(hipMemCreate(&v.h[(size_t)a * v.nchunk + k], chunk, &v.prop, 0));
    }
    return 0;
}
// map: chunk k of array a is backed by handle [a][(k + rot[a]) % nchunk]
static int vmm_map(Vmm& v, const int rot[3]) {
    for (int a = 0; a < 3; a++)
        for (size_t k = 0; k < v.nchunk; k++)
            CK(hipMemMap(v.va + (size_t)a * v.bytes + k * v.chunk, v.chunk, 0, v.h[(size_t)a * v.nchunk + (k + (size_t)rot[a]) % v.nchunk], 0));
    hipMemAccessDesc ad{};
    ad.location = v.prop.location;
    ad.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(v.va, 3 * v.bytes, &ad, 1));
    return 0;
}
static int vmm_unmap(Vmm& v) {
    CK(hipDeviceSynchronize());
    for (size_t i = 0; i < 3 * v.nchunk; i++) CK(hipMemUnmap(v.va + i * v.chunk, v.chunk));
    return 0;
}
static int vmm_destroy(Vmm& v) {
    for (auto& hh : v.h) CK(hipMemRelease(hh));
    CK(hipMemAddressFree(v.va, 3 * v.bytes));
    v = Vmm{};
    return 0;
}

int main(int argc, char** argv) {
    const int nx = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t chunk_mib = argc > 2 ? (size_t)atol(argv[2]) : 64;
    const int K = argc > 3 ? atoi(argv[3]) : 6, R = argc > 4 ? atoi(argv[4]) : 8;
    const int reps = 7;
    int dev = 0;
    CK(hipGetDevice(&dev));
    const size_t bytes = (size_t)nx * TILES * TILE_F4 * sizeof(f4);
    const double gb = 4.0 * (double)bytes * 1e-9;       // 3 reads + 1 write
    printf("sweep of %d planes x %d tiles x 16 KB: %.2f GiB per array, %.2f GB moved per sweep\n", nx, TILES, bytes / 1073741824.0, gb);
    auto report = [&](const char* what, int i, const Timing& t) {
        printf("%-58s %2d  best %.4f ms (%.0f GB/s)  median %.4f ms\n", what, i, t.best, gb / (t.best * 1e-3), t.median);
        fflush(stdout);
    };
    // warm the clocks
    {
        f4 *a, *b, *c;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes));
        CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 0, bytes));
        Timing t;
        for (int i = 0; i < 4; i++) if (time_sweeps(a, b, c, nx, reps, &t)) return 1;
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c));
    }
    // ---- A: hipMalloc, K sets alive together, timed interleaved twice
    {
        std::vector<f4*> p(3 * K, nullptr);
        for (auto& q : p) { CK(hipMalloc(&q, bytes)); CK(hipMemset(q, 0, bytes)); }
        for (int pass = 0; pass < 2; pass++)
            for (int k = 0; k < K; k++) {
                Timing t;
                if (time_sweeps(p[3 * k], p[3 * k + 1], p[3 * k + 2], nx, reps, &t)) return 1;
                report(pass ? "A hipMalloc set (second pass)" : "A hipMalloc set", k, t);
            }
        for (auto& q : p) CK(hipFree(q));
    }
    // ---- B: one physical handle per array
    {
        Vmm v;
        if (vmm_create(v, dev, bytes, bytes, 0)) return 1;
        const int rot[3] = {0, 0, 0};
        if (vmm_map(v, rot)) return 1;
        CK(hipMemset(v.va, 0, 3 * bytes));
        Timing t;
        if (time_sweeps(v.arr(0), v.arr(1), v.arr(2), nx, reps, &t)) return 1;
        report("B vmm, one handle per array", 0, t);
        if (vmm_unmap(v) || vmm_destroy(v)) return 1;
    }
    // ---- C / E: chunks created array by array; then array b rotated against a and c
    {
        Vmm v;
        if (vmm_create(v, dev, bytes, chunk_mib << 20, 0)) return 1;
        printf("chunk %zu MiB, %zu chunks per array\n", v.chunk >> 20, v.nchunk);
        for (int r = 0; r <= R && (size_t)r < v.nchunk; r++) {
            const int rot[3] = {0, r, 0};
            if (vmm_map(v, rot)) return 1;
            if (r == 0) CK(hipMemset(v.va, 0, 3 * bytes));
            Timing t;
            if (time_sweeps(v.arr(0), v.arr(1), v.arr(2), nx, reps, &t)) return 1;
            report(r == 0 ? "C vmm chunks, created array by array" : "E   ... array b rotated by r chunks, r =", r, t);
            if (vmm_unmap(v)) return 1;
        }
        if (vmm_destroy(v)) return 1;
    }
    // ---- D: chunks created round-robin over the arrays
    {
        Vmm v;
        if (vmm_create(v, dev, bytes, chunk_mib << 20, 1)) return 1;
        const int rot[3] = {0, 0, 0};
        if (vmm_map(v, rot)) return 1;
        CK(hipMemset(v.va, 0, 3 * bytes));
        Timing t;
        if (time_sweeps(v.arr(0), v.arr(1), v.arr(2), nx, reps, &t)) return 1;
        report("D vmm chunks, created round-robin over the arrays", 0, t);
        if (vmm_unmap(v) || vmm_destroy(v)) return 1;
    }
    return 0;
}
