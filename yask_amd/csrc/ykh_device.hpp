// ykh_device.hpp -- hand-shaped CDNA4 (gfx950) kernel templates for YASK stencil parts.
//
// This is the MI355X replacement for the reference's generated `calc_vectors` nano/pico-block loop
// (emitter src/compiler/lib/YaskKernel.cpp:591-719; device region D1 = `omp target teams
// distribute` over nano-blocks, src/kernel/Makefile:539-563) and of the masked peel/remainder
// machinery of StencilPartTmpl::calc_nano_block_opt (src/kernel/lib/stencil_calc.hpp:444-860),
// which on a GPU collapses to bounds predicates.
//
// A stencil *part* (struct emitted by the `cdna4_hip` compiler target, see gen/*.hpp) provides
//   - access groups (one base pointer per distinct (var, step-offset)),
//   - the list of read offsets, and
//   - `eval(A&)`: the equations, written against an accessor A with rd<g,dx,dy,dz>() / wr<g>().
// Two kernel shapes instantiate it:
//   naive_kernel   : one thread per point, every read is a (cached) global load. Always legal.
//   star25d_kernel : 2.5-D blocking for axis-aligned ("star") reads of one group:
//                    * a thread block owns a (y,z) tile and marches along x (the largest stride);
//                    * x-neighbours live in a per-thread register queue (depth xlo+xhi+1);
//                    * the centre plane (+ y/z halos) is staged in a double-buffered LDS slab; the
//                      tile interior comes from the queue registers, only halos are re-read
//                      from L2/HBM; y-neighbours are 16-byte LDS row reads, z-neighbours come
//                      from a per-row register window assembled from 16-byte LDS reads;
//                    * every global access is a 16-byte vector along the unit-stride dim z, and a
//                      wavefront (64 lanes) covers 1-2 whole tile rows, so loads are coalesced
//                      and the LDS reads are bank-conflict free;
//                    * next-plane loads are issued before the barrier so HBM latency overlaps the
//                      current plane's arithmetic.
//   No MFMA: the update is ~3.8 flop/byte, bound by HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include "ykh_meta.hpp"
#include "ykh_fn.hpp"

namespace ykh {

// s + v * ck as ONE fused multiply-add, spelled out.  The partial sums of the marching kernels start as a product
// (`sum = centre * c0`) and continue `sum += neighbour * ck`: under -ffp-contract=fast the first of those adds is mul + mul, and
// the compiler may fuse EITHER product into the add -- it did so differently in the even and the odd copies of an unrolled trip,
// so the last bit of a point depended on the parity of (x - block start): x-chunks of odd length, or two instantiations of one
// source, gave results 1 ulp apart at ~1 % of the points (round 3, profiles/r3_bitexact).  With the operation written out every
// copy rounds the same way, whatever the chunking.
template <class V, class T>
__device__ __forceinline__ V fmacc(V v, T ck, V s) { return __builtin_elementwise_fma(v, V(ck), s); }

// compile-time loop: f(integral_constant<int,0>) ... f(integral_constant<int,N-1>)
template <class F, int... I>
__host__ __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// ------------------------------------------------------------------ vector types (16 B along z)
template <typename T> struct vtraits;
template <> struct vtraits<float> {
    static constexpr int VZ = 4;
    typedef float vec __attribute__((ext_vector_type(4)));
};
template <> struct vtraits<double> {
    static constexpr int VZ = 2;
    typedef double vec __attribute__((ext_vector_type(2)));
};

// ------------------------------------------------------------------ kernel arguments
constexpr int MAX_GROUPS = 96;     // PartArgs stays under the 4 KiB kernel-argument limit (fsg_abc needs 93)

// All quantities are in *rank-local* element coordinates: index 0 is the first domain point of
// this rank in each dim; halos/pads have negative indices or indices >= the local domain size.
struct PartArgs {
    void* ptr[MAX_GROUPS];        // per access group: address of local element (0,0,0) in the right step slot
    idx_t gsx[MAX_GROUPS];        // per-group strides (0 for a dim the var does not have)
    idx_t gsy[MAX_GROUPS];
    int gsz[MAX_GROUPS];
    idx_t sx, sy;                 // strides of the shared full-3D layout (all x,y,z vars use it)
    int x0, x1, y0, y1, z0, z1;   // compute box [lo, hi)
    int ax0, ax1, ay0, ay1, az0, az1;   // allocated extent [lo, hi) of the shared layout (load clamps)
    int ntz, nty, nxc, xchunk;    // star25d tiling: tiles in z, y; chunks and chunk length in x
    int ofs_x, ofs_y, ofs_z;      // global index of local 0 (rank offset), for index expressions
    int glast_x, glast_y, glast_z;   // last index of the overall domain (first is 0), for IF_DOMAIN conditions
    idx_t t;                      // evaluation step
    int dom_x1, dom_y1, dom_z1;   // the rank's domain is [0, dom_*1) in local indices (the box x0..z1 may be a part of it)
    int lane_dim;                 // point kernels: domain dim the 64 lanes of a wave run along = the solution's innermost
                                  // (unit-stride) domain dim: 2 for x,y,z solutions, 1 for 2-D, 0 for 1-D ones
    // planned launches of a decomposed rank (ykh_plan.cpp plan_blocks, Solution::launch_planned): workgroup i marches blk[i]
    const BlockDesc* blk;         // null: the regular tiling of the box x0..z1
    unsigned* sig;                // [0] running count of finished signalling blocks, [1] published epoch, [2] waiter's error flag
    unsigned sig_goal;            // the block that raises sig[0] to this value publishes ...
    unsigned sig_epoch;           // ... this epoch in sig[1]: the comm stream's wait_epoch_kernel then lets the halo exchange start
    int xcd_map;                  // experiment knob (YASK_HIP_XCD_MAP, read by kernels of -DYKH_PROFILING builds only): 0 = XCD strips,
                                  // 1 = 4 (y) x 2 (z) XCD blocks, 2 = 2 (y) x 4 (z)  (profiles/r6_iso3dfd_fetch)
};

// The (y, z) tile and x range of the calling workgroup of a marching kernel.  Stores are clipped to the launch's box x0..z1
// either way (a planned launch covers the rank box; each of its blocks is one whole tile of the regular tiling, so a thread
// can only reach points of its own tile) -- the descriptor's y1 / z1 are for the host-side checks.
struct BlockBox { int xs, xe, yt0, zt0, flags; };
template <int VZ, int TZ, int TY, bool DESC>
__device__ __forceinline__ BlockBox block_box(const PartArgs& a) {
    BlockBox b;
    // (DESC is a compile-time flag: the descriptor-reading twin of a kernel shape is a separate instantiation, so that the
    //  regular kernel keeps its register allocation -- the twin needs ~5 more SGPRs, which tips shapes at the 256-VGPR limit
    //  into scratch)
    if constexpr (DESC) {
        // planned launch: everything comes from the descriptor (uniform: kept in SGPRs)
        const BlockDesc* d = a.blk + blockIdx.x;
        b.xs = __builtin_amdgcn_readfirstlane(d->x0); b.xe = __builtin_amdgcn_readfirstlane(d->x1);
        b.yt0 = __builtin_amdgcn_readfirstlane(d->y0);
        b.zt0 = __builtin_amdgcn_readfirstlane(d->z0) & ~(VZ - 1);
        b.flags = __builtin_amdgcn_readfirstlane(d->flags);
        return b;
    } else {
    // XCD-aware tile assignment: block i runs on XCD i % 8; each XCD owns a contiguous range of (y,z) tiles, so that
    // neighbouring tiles share one L2
    const int ntiles = a.ntz * a.nty * a.nxc;
    int bid = blockIdx.x;
    if ((ntiles & 7) == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);
    int tz_i = bid % a.ntz;
    int ty_i = (bid / a.ntz) % a.nty;
    const int xc_i = bid / (a.ntz * a.nty);
#ifdef YKH_PROFILING
    // round-6 experiment: each XCD owns a BLOCK of the (y, z) tile grid instead of a strip of whole tile rows -- fewer rows of
    // y halo shared between XCDs, but z seams whose 8-float halo costs a whole 128-byte line per row (refuted: profiles/r6_iso3dfd_fetch)
    if (a.xcd_map > 0 && a.nxc == 1) {
        const int by = a.xcd_map == 1 ? 4 : 2, bz = 8 / by;          // XCD grid: by along y, bz along z
        if (a.nty % by == 0 && a.ntz % bz == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, rz = a.ntz / bz, ry = a.nty / by;
            tz_i = (xcd % bz) * rz + j % rz;
            ty_i = (xcd / bz) * ry + j / rz;
            (void)ry;
        }
    }
#endif
    b.zt0 = (a.z0 & ~(VZ - 1)) + tz_i * TZ;
    b.yt0 = a.y0 + ty_i * TY;
    b.xs = a.x0 + xc_i * a.xchunk;
    b.xe = (b.xs + a.xchunk < a.x1) ? b.xs + a.xchunk : a.x1;
    b.flags = 0;
    return b;
    }
}
// End of a signalling block of a planned launch: its stores are made visible (system scope: the next reader may be a copy
// engine or a peer), then it counts itself done; the block that completes the count publishes the launch's epoch.
// Recipe of MI355X_MICROARCH.md "hand-off": stores -> barrier -> ONE lane: release fence -> s_waitcnt -> flag.
__device__ __forceinline__ void block_done(const PartArgs& a, int flags) {
    if (!(flags & BLOCK_SIGNALS) || !a.sig) return;          // (uniform)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned old = __hip_atomic_fetch_add(a.sig, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == a.sig_goal) {
            // every other signalling block released its data before its increment, which this thread has observed
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(a.sig + 1, a.sig_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// Soft LOCK-STEP of the workgroups that share an XCD (round 5; the "_ls<K>" shapes of the marching kernels).  Called at the top of every
// loop trip (`step` planes) with d = planes marched so far: in the trip in which the block crosses a multiple of LS_K planes, thread 0
// counts the block in at its XCD's counter (sig + 32 * (blockIdx.x % 8); an agent-scope atomic, i.e. an L2 atomic) and polls it with
// agent-scope (sc1, L1-bypassing) loads until all gridDim.x / 8 workgroups of the XCD have arrived -- at most 400 polls, after which the
// block stops taking part (`dead`; thread 0's copy is the one that is read).  The caller's per-plane barrier holds the other waves.
// sig = 8 counters zeroed by the host before the launch, or null (no lock-step: Solution::launch_part_variant decides).
// (A first version polled at workgroup scope: hipcc turned its fetch_add(0) into an sc0 LOAD, which hits the L1 and never sees the
//  other workgroups' increments -- every wait then ran into the poll limit, profiles/r5_3axis_lockstep.)
template <int LS_K>
__device__ __forceinline__ void xcd_lockstep(unsigned* sig, int d, int step, bool& dead) {
    if constexpr (LS_K > 0) {
        // multiples of LS_K reached by now and by the previous trip: a trip longer than LS_K planes (sweep shapes with long trips)
        // crosses several at once and counts in for each of them, so that the goal -- k arrivals per workgroup -- stays reachable
        // (with one arrival per trip and step > LS_K every block ran into the poll limit, silently)
        const int k = d / LS_K, k_prev = d >= step ? (d - step) / LS_K : 0;                     // uniform
        if (sig && !dead && k > k_prev && threadIdx.x == 0) {
            unsigned* c = sig + (blockIdx.x & 7) * 32;
            const unsigned goal = (unsigned)k * (gridDim.x >> 3);
            __hip_atomic_fetch_add(c, (unsigned)(k - k_prev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spin = 0;
            for (; spin < 400; spin++) {
                if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= goal) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (spin == 400) {
                dead = true;
                // c[1]: workgroups of this XCD that gave up (read back and reported under -trace, Solution::launch_part_variant)
                __hip_atomic_fetch_add(c + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
// point-kernel thread -> (x, y, z): lanes along `lane_dim`, 4 rows per block along the next outer dim, blockIdx.z
// over what is left.  With fewer than 3 domain dims the missing ones have extent 1 (a 2-D solution used to put its 64
// lanes on that size-1 z: 4 active lanes per 256-thread block and fully uncoalesced rows).
struct PointXYZ { int x, y, z; };
__device__ __forceinline__ PointXYZ point_of_thread(const PartArgs& a) {
    const int l = blockIdx.x * 64 + (threadIdx.x & 63), r = blockIdx.y * 4 + (threadIdx.x >> 6), p = blockIdx.z;
    const int ld = a.lane_dim;
    // (selects, not branches, and returned by value: out-parameters by reference made hipcc keep x, y, z in scratch)
    PointXYZ q;
    q.x = a.x0 + (ld == 2 ? p : (ld == 1 ? r : l));
    q.y = a.y0 + (ld == 2 ? r : (ld == 1 ? l : r));
    q.z = a.z0 + (ld == 2 ? l : p);
    return q;
}

// ------------------------------------------------------------------ compile-time read analysis
struct StarRange {
    int xlo, xhi, ylo, yhi, zlo, zhi;
    bool mixed;      // some read has more than one non-zero offset
    bool any;        // group is read at all
    bool center;     // group is read at (0,0,0)
};

template <class P>
constexpr StarRange analyze_group(int g) {
    StarRange r = {0, 0, 0, 0, 0, 0, false, false, false};
    for (int i = 0; i < P::n_reads; i++) {
        if (P::reads[i].g != g) continue;
        r.any = true;
        int dx = P::reads[i].dx, dy = P::reads[i].dy, dz = P::reads[i].dz;
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

// A part fits star25d when exactly one group has off-centre reads, none of them mixed, and every
// var touched uses the shared full-3D layout. Returns that group or -1.
template <class P>
constexpr int star_group() {
    int sg = -1;
    for (int g = 0; g < P::n_groups; g++) {
        StarRange r = analyze_group<P>(g);
        if (!r.any) continue;
        if (r.mixed) return -1;
        bool off = r.xlo || r.xhi || r.ylo || r.yhi || r.zlo || r.zhi;
        if (off) {
            if (sg >= 0) return -1;
            sg = g;
        }
    }
    return sg;
}

template <class P>
constexpr bool group_is_written(int g) {
    for (int i = 0; i < P::n_writes; i++)
        if (P::writes[i] == g) return true;
    return false;
}

// ------------------------------------------------------------------ naive kernel
template <class P>
struct NaiveAcc {
    typedef typename P::real_t T;
    typedef T V;
    const PartArgs& a;
    int x, y, z;
    idx_t c;          // x*sx + y*sy + z: the point's offset in every var over all domain dims (shared strides)
    // Vars over all domain dims share strides and pads: their accesses are (uniform base + uniform offset) + c, so a
    // part with 40 access groups does not need 120 strides in SGPRs (awp: they spilled, and so did the kernel).
    template <int G, int DX, int DY, int DZ>
    __device__ __forceinline__ V rd() const {
        const T* p = (const T*)a.ptr[G];
        if constexpr (P::group_full[G]) return (p + ((idx_t)DX * a.sx + (idx_t)DY * a.sy + DZ))[c];
        else {
            // a var over a subset of the domain dims (its strides are 0 in the dims it lacks): the point's own offset once per GROUP (the
            // same expression for every read of the group: one evaluation), the read's offset as a uniform term the scalar unit computes.
            // Written as (x + DX) * gsx + ... every read cost three 64-bit vector multiplies: test_partial_3d, 59 such reads per point,
            // spent more instructions on addresses than cube spends on its 125 additions (round 6).
            constexpr unsigned gd = GroupDims<P>::get(G);
            if constexpr (gd == 7) {
                const idx_t base = (idx_t)x * a.gsx[G] + (idx_t)y * a.gsy[G] + (idx_t)z * a.gsz[G];
                return p[base + ((idx_t)DX * a.gsx[G] + (idx_t)DY * a.gsy[G] + (idx_t)DZ * a.gsz[G])];
            } else {
                // the var's dims at compile time (the compiler target's group_dims): no multiply for a dim it lacks, none for z (a var
                // that has the innermost domain dim has it at stride 1, Var::compute_geometry); an x-only table is a uniform address
                idx_t o = 0;
                if constexpr ((gd & 1) != 0) o += (idx_t)(x + DX) * a.gsx[G];
                if constexpr ((gd & 2) != 0) o += (idx_t)(y + DY) * a.gsy[G];
                if constexpr ((gd & 4) != 0) o += z + DZ;
                return p[o];
            }
        }
    }
    template <int G>
    __device__ __forceinline__ void wr(V v) const {
        T* p = (T*)a.ptr[G];
        if constexpr (P::group_full[G]) p[c] = v;
        else p[(idx_t)x * a.gsx[G] + (idx_t)y * a.gsy[G] + (idx_t)z * a.gsz[G]] = v;
    }
    __device__ __forceinline__ void pin(V&) const {}   // scheduling hint of the generated code (see MarchAcc)
    // a - b and a / b of the generated code (the compiler target routes them through the accessor, YaskHip.cpp)
    template <class L, class R> __device__ __forceinline__ V sub(L l, R r) const { return V(l) - V(r); }
    template <class L, class R> __device__ __forceinline__ V div(L l, R r) const { return V(l) / V(r); }
    // global index of the point in domain dim D / the evaluation step, as values
    template <int D>
    __device__ __forceinline__ V idx() const { return V(D == 0 ? x + a.ofs_x : (D == 1 ? y + a.ofs_y : z + a.ofs_z)); }
    __device__ __forceinline__ V step() const { return V(a.t); }
    // scalar indices for IF_DOMAIN conditions
    template <int D>
    __device__ __forceinline__ long long sidx() const { return D == 0 ? x + a.ofs_x : (D == 1 ? y + a.ofs_y : z + a.ofs_z); }
    template <int D>
    __device__ __forceinline__ long long first_idx() const { return 0; }
    template <int D>
    __device__ __forceinline__ long long last_idx() const { return D == 0 ? a.glast_x : (D == 1 ? a.glast_y : a.glast_z); }
    __device__ __forceinline__ long long sstep() const { return a.t; }
};

template <class P>
__global__ void __launch_bounds__(256) naive_kernel(const PartArgs a) {
    const PointXYZ q = point_of_thread(a);
    const int x = q.x, y = q.y, z = q.z;
    if (z >= a.z1 || y >= a.y1 || x >= a.x1) return;
    NaiveAcc<P> acc{a, x, y, z, (idx_t)x * a.sx + (idx_t)y * a.sy + z};
    if constexpr (P::has_step_cond_dev) {
        if (!P::step_cond_dev(acc)) return;      // IF_STEP on var values (uniform over the launch)
    }
    if constexpr (P::has_domain_cond) {
        if (!P::cond(acc)) return;      // sub-domain parts: the predicate replaces the reference's BB lists
    }
    P::eval(acc);
}

// Bounding box of a sub-domain (IF_DOMAIN) condition over a box: out[0..2] = min x,y,z, out[3..5] = max x,y,z
// (local indices) of the points where the condition holds, out[6..7] = their number (one 64-bit counter).
// Counterpart of the reference's find_bounding_box() (src/kernel/lib/setup.cpp:1082-1169); run once by
// prepare_solution(), after which the part is launched over its box only.
template <class P>
__global__ void __launch_bounds__(256) cond_bb_kernel(const PartArgs a, int* out) {
    __shared__ int sm[7];
    if (threadIdx.x < 7) sm[threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : (threadIdx.x < 6 ? (int)0x80000000 : 0);
    __syncthreads();
    const PointXYZ q = point_of_thread(a);
    const int x = q.x, y = q.y, z = q.z;
    bool on = false;
    if constexpr (P::has_domain_cond) {
        if (z < a.z1 && y < a.y1 && x < a.x1) {
            NaiveAcc<P> acc{a, x, y, z, 0};
            on = P::cond(acc) != (a.nxc < 0);        // (nxc < 0: the points where the condition does NOT hold -- the "hole" of a ring)
        }
    }
    if (on) {
        atomicMin(&sm[0], x); atomicMin(&sm[1], y); atomicMin(&sm[2], z);
        atomicMax(&sm[3], x); atomicMax(&sm[4], y); atomicMax(&sm[5], z);
        atomicAdd(&sm[6], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && sm[6] > 0) {
        atomicMin(&out[0], sm[0]); atomicMin(&out[1], sm[1]); atomicMin(&out[2], sm[2]);
        atomicMax(&out[3], sm[3]); atomicMax(&out[4], sm[4]); atomicMax(&out[5], sm[5]);
        atomicAdd(reinterpret_cast<unsigned long long*>(&out[6]), (unsigned long long)sm[6]);
    }
}

// Per-index counts of the points of a box where the condition holds, along each domain dim: hist[0 .. nx) for x, then ny entries
// for y, then nz for z (zeroed by the caller; 3-D solutions, lanes along z).  A profile is piecewise constant between the planes
// that bound the condition's region: prepare_solution() cuts the bounding box there and ends up with the reference's list of FULL
// bounding boxes (non-overlapping, valid points only; StencilPartBase::find_bounding_boxes, src/kernel/lib/setup.cpp:1235-1500)
// without ever holding a per-point mask.
template <class P>
__global__ void __launch_bounds__(256) cond_profile_kernel(const PartArgs a, unsigned* hist) {
    __shared__ unsigned hz[64], hy[4], hx;
    if (threadIdx.x < 64) hz[threadIdx.x] = 0;
    if (threadIdx.x < 4) hy[threadIdx.x] = 0;
    if (threadIdx.x == 0) hx = 0;
    __syncthreads();
    const PointXYZ q = point_of_thread(a);
    bool on = false;
    if constexpr (P::has_domain_cond) {
        if (q.z < a.z1 && q.y < a.y1 && q.x < a.x1) {
            NaiveAcc<P> acc{a, q.x, q.y, q.z, 0};
            on = P::cond(acc);
        }
    }
    if (on) { atomicAdd(&hz[threadIdx.x & 63], 1u); atomicAdd(&hy[threadIdx.x >> 6], 1u); atomicAdd(&hx, 1u); }
    __syncthreads();
    const int nx = a.x1 - a.x0, ny = a.y1 - a.y0;
    if (threadIdx.x < 64 && hz[threadIdx.x]) atomicAdd(&hist[nx + ny + blockIdx.x * 64 + threadIdx.x], hz[threadIdx.x]);
    if (threadIdx.x < 4 && hy[threadIdx.x]) atomicAdd(&hist[nx + blockIdx.y * 4 + threadIdx.x], hy[threadIdx.x]);
    if (threadIdx.x == 0 && hx) atomicAdd(&hist[blockIdx.z], hx);
}

// ------------------------------------------------------------------ star25d kernel
enum { ROT_MOVE = 0, ROT_UNROLL = 1, ROT_TRIP = 2, ROT_TRIP2 = 3 };   // (the last two: starlin only, see ykh_starlin.hpp)

template <typename T, int S>
__device__ __forceinline__ typename vtraits<T>::vec zshift(typename vtraits<T>::vec lo, typename vtraits<T>::vec hi) {
    // elements S..S+VZ-1 of the concatenation (lo, hi)
    if constexpr (vtraits<T>::VZ == 4) {
        if constexpr (S == 0) return lo;
        else if constexpr (S == 1) return __builtin_shufflevector(lo, hi, 1, 2, 3, 4);
        else if constexpr (S == 2) return __builtin_shufflevector(lo, hi, 2, 3, 4, 5);
        else return __builtin_shufflevector(lo, hi, 3, 4, 5, 6);
    } else {
        if constexpr (S == 0) return lo;
        else return __builtin_shufflevector(lo, hi, 1, 2);
    }
}

template <class P, int TZL_, int TYL_, int RY_, int ROT_>
struct Star25dCfg {
    typedef typename P::real_t T;
    typedef vtraits<T> VT;
    static constexpr int VZ = VT::VZ;
    static constexpr int TZL = TZL_, TYL = TYL_, RY = RY_, ROT = ROT_;
    static constexpr int NT = TZL * TYL;
    static constexpr int SG = star_group<P>();
    static constexpr StarRange R = analyze_group<P>(SG < 0 ? 0 : SG);
    static constexpr int XL = -R.xlo, XH = R.xhi, YL = -R.ylo, YH = R.yhi, ZL = -R.zlo, ZH = R.zhi;
    static constexpr int NQ = XL + XH + 1;
    static constexpr int ZLV = (ZL + VZ - 1) / VZ, ZHV = (ZH + VZ - 1) / VZ;   // z halo in vectors
    static constexpr int TZ = TZL * VZ, TY = TYL * RY;
    static constexpr int LP = TZ + (ZLV + ZHV) * VZ;    // LDS row pitch (elements), multiple of VZ
    static constexpr int LROWS = TY + YL + YH;
    static constexpr int NHY = (YL + YH) * TZL;          // halo vectors above/below the tile
    static constexpr int NHZ = TY * (ZLV + ZHV);         // halo vectors left/right of the tile
    static constexpr int NH = NHY + NHZ;
    static constexpr int NHT = (NH + NT - 1) / NT;       // halo vectors per thread
    static constexpr int NW = ZLV + 1 + ZHV;             // z-window vectors per row
    static constexpr int NYR = YL + RY + YH;             // y rows visible to a thread
    static constexpr size_t lds_bytes = sizeof(T) * 2 * LROWS * LP;
};

template <class C, class P, int J>
struct StarAcc {
    typedef typename C::T T;
    typedef typename C::VT::vec V;
    static constexpr int VZ = C::VZ;
    const PartArgs& a;
    const V (&q)[C::NQ][C::RY];       // x queue, logical order (index XL = centre plane)
    const V (&yr)[C::NYR];            // rows ly*RY-YL .. ly*RY+RY-1+YH at the thread's z
    const V (&zw)[C::NW];             // z window of row j
    const V (&cen)[MAX_GROUPS];       // centre-only operands of row j, by group
    V (&out)[MAX_GROUPS];             // results of row j, by group
    template <int G, int DX, int DY, int DZ>
    __device__ __forceinline__ V rd() const {
        if constexpr (G == C::SG) {
            static_assert((DX != 0) + (DY != 0) + (DZ != 0) <= 1, "star25d: mixed offset");
            if constexpr (DY == 0 && DZ == 0) return q[C::XL + DX][J];
            else if constexpr (DZ == 0) return yr[C::YL + J + DY];
            else {
                constexpr int e = C::ZLV * VZ + DZ;      // first element within the window
                return zshift<T, e % VZ>(zw[e / VZ], zw[(e / VZ + 1) < C::NW ? (e / VZ + 1) : e / VZ]);
            }
        } else {
            static_assert(DX == 0 && DY == 0 && DZ == 0, "star25d: off-centre read of a non-star group");
            return cen[G];
        }
    }
    template <int G>
    __device__ __forceinline__ void wr(V v) { out[G] = v; }
    __device__ __forceinline__ void pin(V&) const {}
    // a - b and a / b of the generated code (the compiler target routes them through the accessor, YaskHip.cpp)
    template <class L, class R> __device__ __forceinline__ V sub(L l, R r) const { return V(l) - V(r); }
    template <class L, class R> __device__ __forceinline__ V div(L l, R r) const { return V(l) / V(r); }
};

template <typename T>
__device__ __forceinline__ typename vtraits<T>::vec ld_vec(const T* p) {
    return *reinterpret_cast<const typename vtraits<T>::vec*>(p);
}
template <typename T>
__device__ __forceinline__ void st_vec(T* p, typename vtraits<T>::vec v) {
    *reinterpret_cast<typename vtraits<T>::vec*>(p) = v;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ABL (ablation bits, profiling only): 1 = no halo loads, 2 = no centre-operand loads, 4 = no stores.
template <class P, int TZL, int TYL, int RY, int ROT, int ABL = 0>
__global__ void __launch_bounds__(TZL* TYL) star25d_kernel(const PartArgs a) {
    typedef Star25dCfg<P, TZL, TYL, RY, ROT> C;
    typedef typename C::T T;
    typedef typename C::VT::vec V;
    constexpr int VZ = C::VZ, NQ = C::NQ, XL = C::XL, XH = C::XH, YL = C::YL;
    constexpr int ZLV = C::ZLV, ZHV = C::ZHV, LP = C::LP, NT = C::NT, NHT = C::NHT;
    constexpr int SG = C::SG, NG = P::n_groups;
    static_assert(SG >= 0, "part is not star-shaped");
    static_assert(NG <= MAX_GROUPS, "too many access groups");

    extern __shared__ __attribute__((aligned(16))) unsigned char ykh_smem[];
    T* slab = reinterpret_cast<T*>(ykh_smem);

    // ---- tile assignment, XCD-aware: consecutive block ids round-robin over the 8 XCDs
    // (MI355X_MICROARCH.md "Workgroup dispatch"), so give each XCD a contiguous range of tiles:
    // (y,z)-neighbour tiles then share one L2 and re-use each other's halo lines.
    const int ntiles = a.ntz * a.nty * a.nxc;
    int bid = blockIdx.x;
    if ((ntiles & 7) == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);
    int tz_i = bid % a.ntz;
    int ty_i = (bid / a.ntz) % a.nty;
    const int xc_i = bid / (a.ntz * a.nty);
#ifdef YKH_PROFILING
    // round-6 experiment: each XCD owns a BLOCK of the (y, z) tile grid instead of a strip of whole tile rows -- fewer rows of
    // y halo shared between XCDs, but z seams whose 8-float halo costs a whole 128-byte line per row (refuted: profiles/r6_iso3dfd_fetch)
    if (a.xcd_map > 0 && a.nxc == 1) {
        const int by = a.xcd_map == 1 ? 4 : 2, bz = 8 / by;          // XCD grid: by along y, bz along z
        if (a.nty % by == 0 && a.ntz % bz == 0) {
            const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, rz = a.ntz / bz, ry = a.nty / by;
            tz_i = (xcd % bz) * rz + j % rz;
            ty_i = (xcd / bz) * ry + j / rz;
            (void)ry;
        }
    }
#endif

    const int tid = threadIdx.x;
    const int lz = tid % TZL, ly = tid / TZL;
    // tile origin; z origin rounded down to a vector boundary (local index 0 is vector-aligned)
    const int zt0 = (a.z0 & ~(VZ - 1)) + tz_i * C::TZ;
    const int yt0 = a.y0 + ty_i * C::TY;
    const int xs = a.x0 + xc_i * a.xchunk;
    const int xe = (xs + a.xchunk < a.x1) ? xs + a.xchunk : a.x1;
    if (xs >= xe) return;

    const int myz = zt0 + lz * VZ;
    const int zc = clampi(myz, a.az0, a.az1 - VZ);
    const T* __restrict__ sp = (const T*)a.ptr[SG];

    // per-row global offsets within a plane (clamped so that loads never leave the allocation)
    idx_t roff[RY];
    static_for<RY>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        int y = clampi(yt0 + ly * RY + j, a.ay0, a.ay1 - 1);
        roff[j] = (idx_t)y * a.sy + zc;
    });
    // halo vectors owned by this thread: plane offset + LDS element offset
    idx_t hoff[NHT];
    int hlds[NHT];
    static_for<NHT>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        int h = tid + k * NT;
        int row, zv;   // LDS row, LDS vector column
        if (h < C::NHY) {
            int r = h / TZL;
            row = r < YL ? r : r + C::TY;          // rows above, then rows below the tile
            zv = ZLV + h % TZL;
        } else {
            int hh = h - C::NHY;
            int r = hh / (ZLV + ZHV), c = hh % (ZLV + ZHV);
            row = YL + r;
            zv = c < ZLV ? c : c + TZL;            // left vectors, then right vectors
        }
        if (h >= C::NH) { row = 0; zv = 0; }
        int y = clampi(yt0 - YL + row, a.ay0, a.ay1 - 1);
        int z = clampi(zt0 - ZLV * VZ + zv * VZ, a.az0, a.az1 - VZ);
        hoff[k] = (idx_t)y * a.sy + z;
        hlds[k] = (h < C::NH) ? row * LP + zv * VZ : -1;
    });

    auto xplane = [&](int x) -> idx_t { return (idx_t)clampi(x, a.ax0, a.ax1 - 1) * a.sx; };

    V q[NQ][RY];
    V nxt[RY];
    V hreg[NHT];
    V cen_nxt[NG][RY];     // only entries of centre-read, non-star groups are ever touched

    auto load_centres = [&](idx_t pc) {
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g != SG && analyze_group<P>(g).any) {
                const T* gp = (const T*)a.ptr[g];
                static_for<RY>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if constexpr (ABL & 2) cen_nxt[g][j] = V(1); else cen_nxt[g][j] = ld_vec<T>(gp + pc + roff[j]);
                });
            }
        });
    };

    // ---- prologue: fill the queue with planes xs-XL .. xs+XH-1, prefetch plane xs+XH,
    // the halos of plane xs and the centre operands of plane xs.
    static_for<NQ - 1>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        idx_t po = xplane(xs - XL + i);
        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            q[i][j] = ld_vec<T>(sp + po + roff[j]);
        });
    });
    {
        idx_t po = xplane(xs + XH);
        static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; nxt[j] = ld_vec<T>(sp + po + roff[j]); });
        idx_t pc = xplane(xs);
        static_for<NHT>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (ABL & 1) hreg[k] = V(0); else hreg[k] = ld_vec<T>(sp + pc + hoff[k]);
        });
        load_centres(pc);
    }

    // One plane of work. `PH` rotates the physical queue so that no register moves are needed
    // when the x loop is unrolled NQ times (ROT_UNROLL); with ROT_MOVE, PH is always 0.
    auto plane = [&](int x, auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        T* sb = slab + (x & 1) * (C::LROWS * LP);
        // newest plane completes the queue
        static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; q[(PH + NQ - 1) % NQ][j] = nxt[j]; });
        // stage the centre plane: interior from registers, halos from the prefetched registers
        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            st_vec<T>(sb + (YL + ly * RY + j) * LP + (ZLV + lz) * VZ, q[(PH + XL) % NQ][j]);
        });
        static_for<NHT>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (hlds[k] >= 0) st_vec<T>(sb + hlds[k], hreg[k]);
        });
        // centre operands of this plane were prefetched last iteration
        V cen[NG][RY];
        static_for<NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if constexpr (g != SG && analyze_group<P>(g).any)
                static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; cen[g][j] = cen_nxt[g][j]; });
        });
        // prefetch for the next plane before the barrier (overlaps with this plane's math)
        if (x + 1 < xe) {
            idx_t po = xplane(x + 1 + XH), pc = xplane(x + 1);
            static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; nxt[j] = ld_vec<T>(sp + po + roff[j]); });
            static_for<NHT>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (!(ABL & 1)) hreg[k] = ld_vec<T>(sp + pc + hoff[k]);
            });
            load_centres(pc);
        }
        __syncthreads();

        // logical view of the queue for the accessor
        V ql[NQ][RY];
        static_for<NQ>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; ql[i][j] = q[(PH + i) % NQ][j]; });
        });
        // rows visible to this thread along y (own rows come from registers)
        V yr[C::NYR];
        const T* colp = sb + (ZLV + lz) * VZ;
        static_for<C::NYR>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            if constexpr (r >= YL && r < YL + RY) yr[r] = ql[XL][r - YL];
            else yr[r] = ld_vec<T>(colp + (ly * RY + r) * LP);
        });
        const idx_t pc = (idx_t)x * a.sx;
        static_for<RY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            V zw[C::NW];
            const T* rowp = sb + (YL + ly * RY + j) * LP + lz * VZ;
            static_for<C::NW>([&](auto wc) {
                constexpr int w = decltype(wc)::value;
                if constexpr (w == ZLV) zw[w] = ql[XL][j];
                else zw[w] = ld_vec<T>(rowp + w * VZ);
            });
            V cj[MAX_GROUPS], out[MAX_GROUPS];
            static_for<NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if constexpr (g != SG && analyze_group<P>(g).any) cj[g] = cen[g][j];
            });
            StarAcc<C, P, j> acc{a, ql, yr, zw, cj, out};
            P::eval(acc);
            // predicated stores (tile may overhang the compute box)
            const int y = yt0 + ly * RY + j;
            if (y < a.y1 && myz < a.z1 && myz + VZ > a.z0) {
                const idx_t o = pc + (idx_t)y * a.sy + myz;
                static_for<P::n_writes>([&](auto wc) {
                    constexpr int g = P::writes[decltype(wc)::value];
                    T* op = (T*)a.ptr[g] + o;
                    if constexpr (ABL & 4) { if (out[g][0] == T(123.456)) op[0] = out[g][0]; }
                    else if (myz >= a.z0 && myz + VZ <= a.z1) st_vec<T>(op, out[g]);
                    else
                        static_for<VZ>([&](auto ec) {
                            constexpr int e = decltype(ec)::value;
                            if (myz + e >= a.z0 && myz + e < a.z1) op[e] = out[g][e];
                        });
                });
            }
        });
    };

    if constexpr (ROT == ROT_MOVE) {
        for (int x = xs; x < xe; x++) {
            plane(x, std::integral_constant<int, 0>{});
            static_for<NQ - 1>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                static_for<RY>([&](auto jc) { constexpr int j = decltype(jc)::value; q[i][j] = q[i + 1][j]; });
            });
        }
    } else {
        // NQ planes per trip; the physical queue slot of logical entry i is (PH+i)%NQ, so the
        // queue rotates by renaming instead of by register moves.
        int x = xs;
        while (x < xe) {
            bool go = true;
            static_for<NQ>([&](auto phc) {
                if (go) {
                    if (x < xe) { plane(x, phc); x++; }
                    else go = false;
                }
            });
        }
    }
}

}  // namespace ykh
