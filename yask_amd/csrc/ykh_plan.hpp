// ykh_plan.hpp -- device-free decomposition planning (see ykh_plan.cpp).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "ykh_meta.hpp"

namespace ykh {

struct PlanError : public std::runtime_error {
    explicit PlanError(const std::string& m) : std::runtime_error(m) {}
};

struct PlanNeighbor { int rank; int ofs[MAX_DOMAIN_DIMS]; int l1; };

struct RankPlan {
    // in: requested sizes / grid (0 = derive); out: resolved values
    idx_t global_size[MAX_DOMAIN_DIMS] = {0, 0, 0};
    idx_t rank_size[MAX_DOMAIN_DIMS] = {0, 0, 0};
    idx_t num_ranks[MAX_DOMAIN_DIMS] = {0, 0, 0};
    idx_t rank_index[MAX_DOMAIN_DIMS] = {0, 0, 0};
    // out
    idx_t local_size[MAX_DOMAIN_DIMS] = {0, 0, 0};
    idx_t rank_ofs[MAX_DOMAIN_DIMS] = {0, 0, 0};
    std::vector<PlanNeighbor> neighbors;
};

struct VarGeom {
    bool uses_domain[MAX_DOMAIN_DIMS];
    idx_t dom_size[MAX_DOMAIN_DIMS];
    idx_t halo_l[MAX_DOMAIN_DIMS], halo_r[MAX_DOMAIN_DIMS];
    int l1_norm;
    idx_t wext[MAX_DOMAIN_DIMS] = {0, 0, 0};     // wave-front extension: what travels to a neighbour is this much wider (Solution::wf_ext_)
};

void compact_factors(idx_t N, int nd, idx_t* f);
void plan_rank(RankPlan& p, int ndd, int nranks, int rank, const std::vector<std::string>& dim_names, bool rank_index_set);
bool plan_halo_slab(int ndd, const idx_t* num_ranks, const idx_t* rank_index, const VarGeom& v,
                    const PlanNeighbor& nb, bool sending, idx_t* lo, idx_t* n);

// Wave-front temporal tiling along x (Solution::run_wavefront): the launches, in order, that apply `nphases` (step, stage)
// phases to [lo, hi) slab by slab, phase p on the slab shifted by -p * angle.
struct WavefrontLaunch { idx_t phase, lo, hi; };
std::vector<WavefrontLaunch> plan_wavefront(idx_t lo, idx_t hi, idx_t width, idx_t angle, idx_t nphases);

// ---- planned launches of a decomposed rank (ykh_plan.cpp)
struct BlockPlanIn {
    idx_t n[MAX_DOMAIN_DIMS] = {1, 1, 1};                 // rank box [0, n)
    bool has_lo[MAX_DOMAIN_DIMS] = {false, false, false}; // a neighbour rank on the low / high side of dim d
    bool has_hi[MAX_DOMAIN_DIMS] = {false, false, false};
    idx_t width[MAX_DOMAIN_DIMS] = {0, 0, 0};             // what a neighbour needs of my boundary in dim d (halo width or -min_exterior)
    idx_t ty = 1, tz = 1;                                 // tile of the marching kernel (y rows, z elements)
    idx_t overhead = 8;                                   // plane-iterations a block spends before its first output plane
    idx_t ncu = 256;                                      // workgroups resident at a time (one per CU)
    double shell_frac = 0.45;                             // aim: the shell is done after this fraction of the launch
    int mode = 0;                                         // 0: rounds -- every tile cut at the same planes, equal blocks, shell blocks first;
                                                          // 3: the blocks of 0 in regular-launch order (diagnostic); 4: the two x-halves of the
                                                          // pipelined half-exchange schedule (plan_halves below; NOT a shell-first plan)
    idx_t min_len = 16;                                   // no block shorter than this (unless its whole range is)
};
struct BlockPlan {
    std::vector<BlockDesc> blocks;       // in dispatch order: signalling (shell) blocks first
    idx_t n_signal = 0;
    idx_t makespan = 0, shell_done = 0;  // simulated, in plane-iterations
    idx_t undivided = 0;                 // the same box as ONE regular launch (tiles x best uniform chunks), simulated the same way
    int mode_used = 0;                   // 3 rounds, 4 halves
    idx_t cut = 0;                       // mode 4: blocks [0, cut) are half A (the first launch), [cut, size) half B
};
BlockPlan plan_blocks(const BlockPlanIn& in);

// ---- pipelined half-exchanges (Solution::run, -hip_halves): the rank box is cut at x = q1 and x = q2 into an OUTER half
// A = [0, q1) u [q2, nx) and an INNER half B = [q1, q2); both are functions of nx alone, so that ranks that are y / z
// neighbours of each other (same x index, hence the same nx) cut their faces at the same planes.
//   halves_split       false when the box is too short in x for it (q1 must hold the x halo a neighbour needs)
//   halves_slab_ranges the part of a halo slab's x extent [lo, lo + n) that travels with half h: up to two ranges; slabs of a
//                      neighbour that is offset in x (x faces, edges, corners) travel whole with half A
bool halves_split(idx_t nx, idx_t xwidth, idx_t* q1, idx_t* q2);
int halves_slab_ranges(int half, bool x_neighbor, idx_t lo, idx_t n, idx_t q1, idx_t q2, idx_t out_lo[2], idx_t out_n[2]);

}  // namespace ykh
