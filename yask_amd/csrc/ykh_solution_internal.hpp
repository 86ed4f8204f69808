// ykh_solution_internal.hpp -- small helpers shared by the translation units that implement class Solution
// (ykh_solution.cpp, ykh_schedule.cpp, ykh_temporal.cpp, ykh_tune.cpp).
#pragma once
#include "ykh_runtime.hpp"

namespace ykh {

static inline idx_t ceil_div(idx_t a, idx_t b) { return (a + b - 1) / b; }

// Sets a member flag for a scope and puts the old value back on every way out (a throwing launch must not leave
// `in_outer_loop`, `launching_exterior`, ... set for the calls that follow).
template <typename T>
struct ScopedSet {
    T& ref; T old;
    ScopedSet(T& r, T v) : ref(r), old(r) { ref = v; }
    ~ScopedSet() { ref = old; }
    ScopedSet(const ScopedSet&) = delete;
    ScopedSet& operator=(const ScopedSet&) = delete;
};

// grid of the point kernels / cond_bb_kernel over a box (see point_of_thread(), ykh_device.hpp)
static inline dim3 point_grid(const Box& b, int lane_dim) {
    const idx_t n[3] = {b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2]};
    if (lane_dim == 2) return dim3((unsigned)ceil_div(n[2], 64), (unsigned)ceil_div(n[1], 4), (unsigned)n[0]);
    if (lane_dim == 1) return dim3((unsigned)ceil_div(n[1], 64), (unsigned)ceil_div(n[0], 4), (unsigned)n[2]);
    return dim3((unsigned)ceil_div(n[0], 64), (unsigned)ceil_div(n[1], 4), (unsigned)n[2]);
}

}  // namespace ykh
