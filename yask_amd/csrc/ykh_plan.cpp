// ykh_plan.cpp -- domain decomposition as pure index arithmetic (no HIP calls, usable without a GPU).
//
// Re-statement, for one process per GPU, of the reference's rank set-up:
//   rank grid            <- StencilContext::setup_rank (src/kernel/lib/setup.cpp:169-524),
//                           most-compact factorisation (src/common/tuple.cpp:355-430);
//   rank id -> coords    <- unlayout with the first domain dim varying fastest (setup.cpp:242-243);
//   local sizes/offsets  <- ceil(global/nranks), remainder on the last rank (setup.cpp:453-495);
//   neighbours           <- up to 3^N-1 (MPIInfo, settings.hpp:331-433);
//   halo slabs           <- alloc_mpi_data buffer geometry (src/kernel/lib/alloc.cpp:456-859).
// Solution::setup_rank()/alloc_halo_buffers() and the C ABI's yk_plan_* entry points share this code,
// so the multi-process CPU tests exercise exactly what runs on the GPUs.
#include <algorithm>
#include <cstdlib>
#include <sstream>

#include "ykh_plan.hpp"

namespace ykh {

static inline idx_t ceil_div(idx_t a, idx_t b) { return (a + b - 1) / b; }

// Most-compact factorisation: minimise the largest factor; candidates are visited with the factor
// of dim 1 varying fastest and dim 0 derived (src/common/tuple.cpp:355-430), first best wins.
void compact_factors(idx_t N, int nd, idx_t* f) {
    idx_t given = 1;
    bool all = true;
    for (int d = 0; d < nd; d++) { if (f[d] > 0) given *= f[d]; else all = false; }
    if (all && given == N) return;
    std::vector<idx_t> facts;
    for (idx_t n = 1; n <= N; n++) if (N % n == 0) facts.push_back(n);
    for (int keep = 1; keep >= 0; keep--) {
        idx_t best[MAX_DOMAIN_DIMS] = {0, 0, 0};
        idx_t best_max = -1;
        std::vector<size_t> ix(nd, 0);
        while (true) {
            idx_t can[MAX_DOMAIN_DIMS] = {1, 1, 1};
            for (int d = 1; d < nd; d++) can[d] = (keep && f[d] > 0) ? f[d] : facts[ix[d]];
            idx_t rest = 1;
            for (int d = 1; d < nd; d++) rest *= can[d];
            if (keep && f[0] > 0) can[0] = f[0];
            else can[0] = (N % rest == 0) ? N / rest : 0;
            idx_t prod = 1, mx = 0;
            for (int d = 0; d < nd; d++) { prod *= can[d]; mx = std::max(mx, can[d]); }
            if (can[0] > 0 && prod == N && (best_max < 0 || mx < best_max)) {
                best_max = mx;
                for (int d = 0; d < nd; d++) best[d] = can[d];
            }
            int d = 1;
            for (; d < nd; d++) {
                if (keep && f[d] > 0) continue;
                if (++ix[d] < facts.size()) break;
                ix[d] = 0;
            }
            if (d >= nd) break;
        }
        if (best_max >= 0) { for (int d = 0; d < nd; d++) f[d] = best[d]; return; }
    }
    throw PlanError("cannot factor " + std::to_string(N) + " ranks over the domain dims");
}

void plan_rank(RankPlan& p, int ndd, int nranks, int rank, const std::vector<std::string>& dim_names,
               bool rank_index_set) {
    idx_t f[MAX_DOMAIN_DIMS] = {1, 1, 1};
    for (int d = 0; d < ndd; d++) f[d] = p.num_ranks[d];
    compact_factors(nranks, ndd, f);
    idx_t prod = 1;
    for (int d = 0; d < ndd; d++) prod *= f[d];
    auto dn = [&](int d) { return d < (int)dim_names.size() ? dim_names[d] : std::string(1, char('x' + d)); };
    if (prod != nranks) {
        std::ostringstream os;
        os << prod << " rank(s) requested (";
        for (int d = 0; d < ndd; d++) os << (d ? " * " : "") << dn(d) << "=" << f[d];
        os << "), but " << nranks << " rank(s) are active";
        throw PlanError(os.str());
    }
    for (int d = 0; d < ndd; d++) p.num_ranks[d] = f[d];
    // rank id -> coords with the first domain dim varying fastest (Tuple::unlayout, first_inner)
    if (!rank_index_set) {
        idx_t me = rank;
        for (int d = 0; d < ndd; d++) { p.rank_index[d] = me % p.num_ranks[d]; me /= p.num_ranks[d]; }
    }
    for (int d = 0; d < ndd; d++)
        if (p.rank_index[d] < 0 || p.rank_index[d] >= p.num_ranks[d])
            throw PlanError("rank index of " + std::to_string(p.rank_index[d]) + " is not within allowed range [0 ... " +
                            std::to_string(p.num_ranks[d] - 1) + "] in '" + dn(d) + "' dimension on rank " + std::to_string(rank));
    // sizes: either global or local given per dim (setup.cpp:453-495)
    for (int d = 0; d < ndd; d++) {
        if (p.global_size[d] > 0 && p.rank_size[d] == 0) {
            idx_t base = ceil_div(p.global_size[d], p.num_ranks[d]);
            idx_t last = p.global_size[d] - base * (p.num_ranks[d] - 1);
            if (last <= 0)
                throw PlanError("global-domain size " + std::to_string(p.global_size[d]) + " in '" + dn(d) +
                                "' cannot be split over " + std::to_string(p.num_ranks[d]) + " ranks");
            p.local_size[d] = (p.rank_index[d] == p.num_ranks[d] - 1) ? last : base;
            p.rank_ofs[d] = p.rank_index[d] * base;
        } else if (p.rank_size[d] > 0) {
            p.local_size[d] = p.rank_size[d];
            // every rank in a grid line must use the same size for the offsets to be derivable locally
            p.rank_ofs[d] = p.rank_index[d] * p.rank_size[d];
            p.global_size[d] = p.rank_size[d] * p.num_ranks[d];
        } else {
            throw PlanError("both local-domain size and global-domain size are zero in '" + dn(d) +
                            "' dimension on rank " + std::to_string(rank) + "; specify one, and the other will be calculated");
        }
    }
    // neighbours
    p.neighbors.clear();
    int nloop[3] = {ndd > 0 ? 3 : 1, ndd > 1 ? 3 : 1, ndd > 2 ? 3 : 1};
    for (int a = 0; a < nloop[0]; a++)
        for (int b = 0; b < nloop[1]; b++)
            for (int c = 0; c < nloop[2]; c++) {
                int o[3] = {nloop[0] > 1 ? a - 1 : 0, nloop[1] > 1 ? b - 1 : 0, nloop[2] > 1 ? c - 1 : 0};
                if (!o[0] && !o[1] && !o[2]) continue;
                idx_t co[3];
                bool ok = true;
                for (int d = 0; d < ndd; d++) {
                    co[d] = p.rank_index[d] + o[d];
                    if (co[d] < 0 || co[d] >= p.num_ranks[d]) ok = false;
                }
                if (!ok) continue;
                idx_t id = 0;
                for (int d = ndd - 1; d >= 0; d--) id = id * p.num_ranks[d] + co[d];
                PlanNeighbor nb;
                nb.rank = (int)id;
                nb.l1 = 0;
                for (int d = 0; d < MAX_DOMAIN_DIMS; d++) { nb.ofs[d] = d < ndd ? o[d] : 0; nb.l1 += std::abs(nb.ofs[d]); }
                p.neighbors.push_back(nb);
            }
}

// What travels between this rank and the neighbour at offset nb.ofs for one var:
//   in each dim d with ofs[d] = -1 my first halo_r[d] domain points (they fill the neighbour's right
//   halo), with ofs[d] = +1 my last halo_l[d] points, with ofs[d] = 0 my whole extent in d -- extended
//   into my own halo at a *global* boundary when the var is read diagonally (l1_norm > 1), as
//   alloc.cpp:544-553 does, so that corner cells end up identical to a single-rank run.
bool plan_halo_slab(int ndd, const idx_t* num_ranks, const idx_t* rank_index, const VarGeom& v,
                    const PlanNeighbor& nb, bool sending, idx_t* lo, idx_t* n) {
    if (v.l1_norm < nb.l1) return false;
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++) {
        const int o = nb.ofs[d];
        if (d >= ndd || !v.uses_domain[d]) {
            if (o != 0) return false;   // var does not extend in a dim the neighbour is offset in
            lo[d] = 0; n[d] = 1;
            continue;
        }
        const idx_t sz = v.dom_size[d];
        if (o == 0) {
            idx_t l = 0, h = sz;
            if (v.l1_norm > 1) {
                if (rank_index[d] == 0) l -= v.halo_l[d];
                if (rank_index[d] == num_ranks[d] - 1) h += v.halo_r[d];
            }
            lo[d] = l; n[d] = h - l;
        } else if (sending) {
            idx_t w = ((o < 0) ? v.halo_r[d] : v.halo_l[d]) + v.wext[d];
            lo[d] = (o < 0) ? 0 : sz - w;
            n[d] = w;
        } else {
            idx_t w = ((o < 0) ? v.halo_l[d] : v.halo_r[d]) + v.wext[d];
            lo[d] = (o < 0) ? -w : sz;
            n[d] = w;
        }
        if (n[d] <= 0) return false;
    }
    return true;
}


// The reference's wave-front schedule (StencilContext::calc_mega_block / shift_mega_block, src/kernel/lib/context.cpp:
// 482-745,1181-1525) in one dim: slabs of `width` start at lo, lo+width, ... until the LAST phase of a slab has covered hi;
// phase p of a slab covers [x0 - p*angle, x0 + width - p*angle) clipped to [lo, hi).  With angle >= the stencil's reach in
// x, phase p+1 finds its inputs computed (this slab or an earlier one) and what it overwrites in place is no longer needed.
std::vector<WavefrontLaunch> plan_wavefront(idx_t lo, idx_t hi, idx_t width, idx_t angle, idx_t nphases) {
    std::vector<WavefrontLaunch> out;
    if (hi <= lo || nphases < 1) return out;
    angle = std::max<idx_t>(1, angle);
    width = std::max(width, angle);
    for (idx_t x0 = lo; x0 < hi + (nphases - 1) * angle; x0 += width)
        for (idx_t p = 0; p < nphases; p++) {
            const idx_t a = std::max(lo, x0 - p * angle), b = std::min(hi, x0 + width - p * angle);
            if (b > a) out.push_back({p, a, b});
        }
    return out;
}

// ------------------------------------------------------------------ planned launch of a decomposed rank
// The reference computes the exterior of a rank first, starts the halo exchange, and computes the interior while the messages
// travel (StencilContext::run_solution, src/kernel/lib/context.cpp:377-478; `-min_exterior`, settings.hpp:245).  Round 2 did
// that with separate launches -- slabs, then the interior in pieces -- and paid 1.2-1.5x the undivided sweep on a 512^3 block:
// thin slabs leave CUs idle or fall back to the point kernel, every launch boundary drains the chip, every interior piece
// re-runs the marching kernel's prologue.  Here the whole rank box is ONE launch of the marching kernel whose workgroups are
// handed (tile, x-range) descriptors in a planned order:
//   1. x-face shell: for every (y, z) tile the `width` planes next to an x neighbour (thin blocks);
//   2. y/z shell: the tiles that touch a y or z face with a neighbour -- WHOLE tiles of the regular tiling, so they cost what
//      they cost in an undivided sweep -- cut into x-chunks short enough to finish after ~shell_frac of the launch;
//   3. the interior tiles, in x-pieces sized so that every CU finishes at the same time: CUs that were idle from the start
//      get long pieces, CUs that first ran shell blocks get the rest (list scheduling against the simulated CU time line).
// Blocks are dispatched in index order, so the shell runs first and the interior backfills; shell blocks count themselves
// done on the device and the last one releases the exchange (Solution::launch_planned).  Same kernel, same per-point
// arithmetic as a one-rank sweep: results are bit-identical whatever the plan.
namespace {
struct CuLine {          // availability times of the CUs; blocks go to the CU that frees first (the hardware's own rule)
    std::vector<idx_t> t;
    explicit CuLine(idx_t n) : t((size_t)n, 0) {}
    size_t earliest() const { return (size_t)(std::min_element(t.begin(), t.end()) - t.begin()); }
    idx_t run(idx_t len) { const size_t c = earliest(); const idx_t s = t[c]; t[c] = s + len; return s; }
    idx_t makespan() const { return *std::max_element(t.begin(), t.end()); }
};
struct TileBox { idx_t y0, y1, z0, z1; };
// blocks go round-robin over 8 strips of the tile-ordered list: block i of a launch runs on XCD i % 8, so each XCD's L2
// serves a contiguous strip of neighbouring tiles (the regular launches' mapping, ykh_starlin.hpp)
template <class T>
std::vector<T> deal_over_xcds(const std::vector<T>& v) {
    const size_t n = v.size();
    if (n < 16) return v;
    std::vector<T> out;
    out.reserve(n);
    size_t lo[9];
    for (int k = 0; k <= 8; k++) lo[k] = n * (size_t)k / 8;
    for (size_t r = 0; out.size() < n; r++)
        for (int k = 0; k < 8; k++)
            if (lo[k] + r < lo[k + 1]) out.push_back(v[lo[k] + r]);
    return out;
}
}  // namespace

// Mode 0 ("rounds").  What round 3's first planner (thin x slabs, greedy per-CU budgets or uniform interior chunks: deleted in round 5)
// got wrong, measured on the GPU (gpurun_out/r3a, 1.3-1.6x):
// pieces sized per CU start at different times and places, so the ~250 concurrent workgroups of a launch march through ~250
// different x planes -- and the kernel runs 1.5x slower per plane than in a regular launch, where all tiles of a chunk sweep
// the same planes at the same time (DRAM pages, TLB reach and the L2 sharing of halo lines all depend on it; round 2 had seen
// 10 % from mere drift).  So: EVERY tile is cut at the same k planes (the regular launch's own structure, k chosen so that the
// launch takes at least two rounds of workgroups), all blocks have the same length and therefore start and end together round
// after round, and only the ORDER is planned: first every block a neighbour needs something of -- all chunks of the y/z shell
// tiles, and the first / last chunk of every tile where an x neighbour exists (its face planes are complete when the chunk
// ends) -- then the rest.  No thin slabs, no extra prologues beyond those of the k chunks.
static BlockPlan plan_rounds(const BlockPlanIn& in) {
    BlockPlan out;
    const idx_t nx = in.n[0], ny = in.n[1], nz = in.n[2];
    const idx_t o = std::max<idx_t>(0, in.overhead), minlen = std::max<idx_t>(1, in.min_len);
    std::vector<TileBox> tiles;
    std::vector<char> shell;
    auto touches = [&](idx_t a, idx_t b, idx_t n, int d) {
        const idx_t w = std::max<idx_t>(1, in.width[d]);
        return (in.has_lo[d] && a < w) || (in.has_hi[d] && b > n - w);
    };
    for (idx_t y0 = 0; y0 < ny; y0 += in.ty)
        for (idx_t z0 = 0; z0 < nz; z0 += in.tz) {
            TileBox tb{y0, std::min(y0 + in.ty, ny), z0, std::min(z0 + in.tz, nz)};
            tiles.push_back(tb);
            shell.push_back(touches(tb.y0, tb.y1, ny, 1) || touches(tb.z0, tb.z1, nz, 2));
        }
    const idx_t wx = std::max<idx_t>(1, in.width[0]);
    auto build = [&](idx_t k, std::vector<BlockDesc>* blocks, idx_t* n_sig, idx_t* shell_done, idx_t* makespan) {
        std::vector<BlockDesc> p1, p2;
        for (idx_t c = 0; c < k; c++) {
            const idx_t a = nx * c / k, b = nx * (c + 1) / k;
            // (an x neighbour needs my first / last wx planes: every chunk that holds some of them)
            const bool xface = (in.has_lo[0] && a < wx) || (in.has_hi[0] && b > nx - wx);
            for (size_t i = 0; i < tiles.size(); i++) {
                BlockDesc d;
                d.x0 = (int)a; d.x1 = (int)b; d.y0 = (int)tiles[i].y0; d.y1 = (int)tiles[i].y1; d.z0 = (int)tiles[i].z0; d.z1 = (int)tiles[i].z1;
                d.flags = (shell[i] || xface) ? BLOCK_SIGNALS : 0;
                d.start = 0;
                (d.flags ? p1 : p2).push_back(d);
            }
        }
        std::vector<BlockDesc> all = p1;
        all.insert(all.end(), p2.begin(), p2.end());
        if (in.mode == 3) {
            // diagnostic: the same blocks in the order of a regular launch (chunk-major), shell blocks NOT first -- what does the
            // shell-first order cost by itself?  (tools/decomp_cost.py)
            all.clear();
            for (idx_t c = 0; c < k; c++) {
                for (auto& d : p1) if (d.x0 == (int)(nx * c / k)) all.push_back(d);
                for (auto& d : p2) if (d.x0 == (int)(nx * c / k)) all.push_back(d);
            }
        }
        // simulate the dispatch; within each round of ncu blocks, deal the blocks over the XCD strips
        CuLine cl(in.ncu);
        idx_t sd = 0;
        for (auto& d : all) {
            d.start = (int)cl.run((d.x1 - d.x0) + o);
            if (d.flags) sd = std::max<idx_t>(sd, d.start + (d.x1 - d.x0) + o);
        }
        if (blocks) {
            blocks->clear();
            for (size_t r0 = 0; r0 < all.size(); r0 += (size_t)in.ncu) {
                std::vector<BlockDesc> seg(all.begin() + r0, all.begin() + std::min(all.size(), r0 + (size_t)in.ncu));
                // (the shell blocks of a segment stay in front of its other blocks: dealt separately)
                std::vector<BlockDesc> a1, a2;
                for (auto& d : seg) (d.flags ? a1 : a2).push_back(d);
                a1 = deal_over_xcds(a1); a2 = deal_over_xcds(a2);
                blocks->insert(blocks->end(), a1.begin(), a1.end());
                blocks->insert(blocks->end(), a2.begin(), a2.end());
            }
        }
        *n_sig = (idx_t)p1.size(); *shell_done = p1.empty() ? 0 : sd; *makespan = cl.makespan();
    };
    // candidates: k chunks per tile.  Among those within 20 % of the shortest launch: the shortest whose shell is done by shell_frac
    // of it; if none is, the one with the earliest shell.
    struct Cand { idx_t k, span; double frac; };
    std::vector<Cand> cands;
    idx_t shortest = -1;
    for (idx_t k = 1; k <= 64; k++) {
        if (k > 1 && nx / k < minlen) break;
        idx_t ns, sd, ms;
        build(k, nullptr, &ns, &sd, &ms);
        cands.push_back({k, ms, ms > 0 ? (double)sd / (double)ms : 1.0});
        if (shortest < 0 || ms < shortest) shortest = ms;
    }
    idx_t best_k = 1, best_span = -1;
    double best_frac = 2.0;
    bool best_ok = false;
    for (const Cand& c : cands) {
        if (c.span * 5 > shortest * 6) continue;
        const bool ok = c.frac <= in.shell_frac + 1e-9;
        bool better;
        if (ok != best_ok) better = ok;
        else if (ok) better = best_span < 0 || c.span < best_span;
        else better = c.frac < best_frac - 0.02 || (c.frac < best_frac + 0.02 && (best_span < 0 || c.span < best_span));
        if (better) { best_k = c.k; best_span = c.span; best_frac = c.frac; best_ok = ok; }
    }
    build(best_k, &out.blocks, &out.n_signal, &out.shell_done, &out.makespan);
    out.mode_used = 3;
    // the undivided box as one regular launch, for reference
    idx_t best = -1;
    for (idx_t k = 1; k <= 64; k++) {
        const idx_t len = (nx + k - 1) / k;
        if (k > 1 && len < 32) break;
        CuLine cl(in.ncu);
        for (idx_t c = 0; c * len < nx; c++)
            for (size_t i = 0; i < tiles.size(); i++) cl.run(std::min(len, nx - c * len) + o);
        if (best < 0 || cl.makespan() < best) best = cl.makespan();
    }
    out.undivided = best;
    return out;
}

// ------------------------------------------------------------------ mode 4: the two x-halves of the pipelined half-exchange schedule
// What the shell-first order of mode 0 costs (1.05-1.13x the undivided sweep, against 1.01-1.02x for the SAME
// blocks in regular-launch order) is the order itself: shell tiles marching without their neighbours.  Mode 4 keeps the regular
// order -- every tile, chunk after chunk -- and moves the overlap into TIME instead: the box is two launches, the outer x-half
// A = [0, q1) u [q2, nx) and the inner half B = [q1, q2), and each launch is followed by the exchange of ITS part of the faces
// (x faces and the A part of the y / z faces after A; the B part after B), which travels while the OTHER half is computed:
// A's halos are needed again by the next step's A launch, one whole B launch later (Solution::run, halves).  No block is
// special, so none signals; `cut` separates the two launches.
bool halves_split(idx_t nx, idx_t xwidth, idx_t* q1, idx_t* q2) {
    const idx_t a = nx / 4;
    if (a < 1 || a < xwidth || nx - 2 * a < 1) return false;
    *q1 = a; *q2 = nx - a;
    return true;
}
int halves_slab_ranges(int half, bool x_neighbor, idx_t lo, idx_t n, idx_t q1, idx_t q2, idx_t out_lo[2], idx_t out_n[2]) {
    const idx_t hi = lo + n;
    int k = 0;
    auto add = [&](idx_t a, idx_t b) { if (b > a) { out_lo[k] = a; out_n[k] = b - a; k++; } };
    if (x_neighbor) { if (half == 0) add(lo, hi); return k; }
    if (half == 0) { add(lo, std::min(hi, q1)); add(std::max(lo, q2), hi); }
    else add(std::max(lo, q1), std::min(hi, q2));
    return k;
}
static BlockPlan plan_halves(const BlockPlanIn& in) {
    BlockPlan out;
    const idx_t nx = in.n[0], ny = in.n[1], nz = in.n[2];
    const idx_t o = std::max<idx_t>(0, in.overhead), minlen = std::max<idx_t>(1, in.min_len);
    idx_t q1 = 0, q2 = 0;
    if (!halves_split(nx, std::max<idx_t>(1, in.width[0]), &q1, &q2)) throw PlanError("plan_blocks: the box is too short in x for two halves");
    std::vector<TileBox> tiles;
    for (idx_t y0 = 0; y0 < ny; y0 += in.ty)
        for (idx_t z0 = 0; z0 < nz; z0 += in.tz) tiles.push_back(TileBox{y0, std::min(y0 + in.ty, ny), z0, std::min(z0 + in.tz, nz)});
    // kq pieces per outer quarter, 2 kq for the inner half: all pieces about nx / (4 kq) planes long
    auto build = [&](idx_t kq, std::vector<BlockDesc>* blocks, idx_t* cut, idx_t* span_a, idx_t* span_b) {
        std::vector<BlockDesc> half[2];
        auto pieces = [&](int h, idx_t a, idx_t b, idx_t k) {
            for (idx_t c = 0; c < k; c++) {
                const idx_t x0 = a + (b - a) * c / k, x1 = a + (b - a) * (c + 1) / k;
                if (x1 <= x0) continue;
                for (const TileBox& tb : tiles) {
                    BlockDesc d;
                    d.x0 = (int)x0; d.x1 = (int)x1; d.y0 = (int)tb.y0; d.y1 = (int)tb.y1; d.z0 = (int)tb.z0; d.z1 = (int)tb.z1;
                    d.flags = 0; d.start = 0;
                    half[h].push_back(d);
                }
            }
        };
        pieces(0, 0, q1, kq);
        pieces(0, q2, nx, kq);
        pieces(1, q1, q2, 2 * kq);
        idx_t span[2];
        for (int h = 0; h < 2; h++) {
            CuLine cl(in.ncu);
            for (auto& d : half[h]) d.start = (int)cl.run((d.x1 - d.x0) + o);
            span[h] = cl.makespan();
        }
        *span_a = span[0]; *span_b = span[1];
        *cut = (idx_t)half[0].size();
        if (blocks) {
            blocks->clear();
            for (int h = 0; h < 2; h++)
                for (size_t r0 = 0; r0 < half[h].size(); r0 += (size_t)in.ncu) {
                    // (each launch starts again at XCD 0; within a round of workgroups the blocks are dealt over the XCD strips)
                    std::vector<BlockDesc> seg(half[h].begin() + r0, half[h].begin() + std::min(half[h].size(), r0 + (size_t)in.ncu));
                    seg = deal_over_xcds(seg);
                    for (auto& d : seg) { if (h == 1) d.start += (int)span[0]; blocks->push_back(d); }
                }
        }
    };
    idx_t best_kq = 1, best = -1;
    for (idx_t kq = 1; kq <= 16; kq++) {
        if (kq > 1 && q1 / kq < minlen) break;
        idx_t c, sa, sb;
        build(kq, nullptr, &c, &sa, &sb);
        if (best < 0 || sa + sb < best) { best = sa + sb; best_kq = kq; }
    }
    idx_t sa = 0, sb = 0;
    build(best_kq, &out.blocks, &out.cut, &sa, &sb);
    out.n_signal = 0;
    out.shell_done = sa;              // (what a neighbour needs of half A is complete when its launch ends)
    out.makespan = sa + sb;
    out.mode_used = 4;
    idx_t und = -1;
    for (idx_t k = 1; k <= 64; k++) {
        const idx_t len = (nx + k - 1) / k;
        if (k > 1 && len < 32) break;
        CuLine cl(in.ncu);
        for (idx_t c = 0; c * len < nx; c++)
            for (size_t i = 0; i < tiles.size(); i++) cl.run(std::min(len, nx - c * len) + o);
        if (und < 0 || cl.makespan() < und) und = cl.makespan();
    }
    out.undivided = und;
    return out;
}

BlockPlan plan_blocks(const BlockPlanIn& in) {
    BlockPlan out;
    const idx_t nx = in.n[0], ny = in.n[1], nz = in.n[2];
    if (nx < 1 || ny < 1 || nz < 1 || in.ty < 1 || in.tz < 1 || in.ncu < 1) throw PlanError("plan_blocks: bad box or tile");
    if (in.mode == 4) return plan_halves(in);
    if (in.mode != 0 && in.mode != 3) throw PlanError("plan_blocks: mode must be 0 (rounds, shell first), 3 (the same blocks in regular order) or 4 (halves)");
    return plan_rounds(in);
}

}  // namespace ykh
