// ykh_schedule.cpp -- how ONE step of a decomposed rank is issued (part of class Solution, split off ykh_solution.cpp in round 4):
// exterior / interior split, planned launches (shell blocks first, the exchange released from the device), pipelined
// half-exchanges (-hip_halves), and the per-phase timers behind yk_stats.  The reference's counterpart is the exterior-first
// order of StencilContext::run_solution with adv_halo_exchange (src/kernel/lib/context.cpp:377-478, halo.cpp:494-574).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "ykh_runtime.hpp"
#include "ykh_solution_internal.hpp"

namespace ykh {

// ------------------------------------------------------------------ exterior / interior split of a decomposed run
// Interior box of a rank with neighbours on the given sides (alloc.cpp:686-723 `mpi_interior`): the exterior is what the
// neighbours need, computed first.  Width per dim = the halo (or -min_exterior); in z one marching tile (see prepare()).
Box Solution::interior_for(const bool* has_lo, const bool* has_hi) const {
    Box ib = rank_box();
    for (int d = 0; d < ndd; d++) {
        idx_t w = std::max<idx_t>(std::max(shared_pad_l_[d], shared_pad_r_[d]), min_exterior);
        if (d == 2 && ndd == 3 && min_exterior == 0 && !impl.parts.empty() && part_variant[0] >= 0) {
            const KernelVariant& kv = impl.parts[0].variants[part_variant[0]];
            const idx_t tz = kv.star && kv.rx == 0 ? kv.tz : 0;
            if (tz > w && local_size[2] >= ((has_lo[d] ? 1 : 0) + (has_hi[d] ? 1 : 0) + 1) * tz) w = tz;
        }
        if (has_lo[d]) ib.lo[d] += w;
        if (has_hi[d]) ib.hi[d] -= w;
    }
    return ib;
}
// exterior slabs first (context.cpp:377-444) ...
void Solution::launch_exterior(const StageMeta& sm, idx_t t, const Box& ib) {
    ScopedSet<bool> ext(launching_exterior, true);
    Box rem = rank_box();
    for (int d = 0; d < ndd; d++) {
        if (ib.lo[d] > rem.lo[d]) {
            Box s = rem; s.hi[d] = ib.lo[d];
            for (int k = 0; k < sm.n_parts; k++) launch_part(sm.parts[k], t, s, compute_stream);
            rem.lo[d] = ib.lo[d];
        }
        if (ib.hi[d] < rem.hi[d]) {
            Box s = rem; s.lo[d] = ib.hi[d];
            for (int k = 0; k < sm.n_parts; k++) launch_part(sm.parts[k], t, s, compute_stream);
            rem.hi[d] = ib.hi[d];
        }
    }
}
// ... then the interior, while the halos travel: ONE launch (round 2 split it along x to let RCCL's kernels in at the boundaries,
// each split re-running the prologue: iso3dfd 512^3 interior 0.355 / 0.387 / 0.472 ms at 1 / 2 / 4 splits; the copy-based
// transport needs no CU, and the schedules that overlap by construction -- halves, planned -- replaced the idea).
void Solution::launch_interior(const StageMeta& sm, idx_t t, const Box& ib) {
    ScopedSet<bool> inter(launching_interior, true);
    for (int k = 0; k < sm.n_parts; k++) launch_part(sm.parts[k], t, ib, compute_stream);
}
// What the compute side of one step costs a rank with neighbours on the given sides -- the same launches run() issues,
// without any communication -- against the undivided box.  tools/decomp_cost.py; ms[0] = exterior, ms[1] = interior,
// ms[2] = whole box in one piece.
void Solution::time_decomposed_step(const bool* has_lo, const bool* has_hi, int reps, float* ms) {
    if (!prepared) YKH_THROW("time_decomposed_step() called without calling prepare_solution() first");
    const Box ib = interior_for(has_lo, has_hi), rb = rank_box();
    if (ib.empty()) YKH_THROW("time_decomposed_step(): no interior left");
    hipEvent_t e[4];
    for (auto& x : e) YKH_HIP(hipEventCreate(&x));
    float acc[3] = {0, 0, 0};
    bool all_planned = true;
    for (int st = 0; st < meta->n_stages; st++) all_planned &= planned_part(meta->stages[st]) >= 0;
    // -hip_halves: the two half-launches of the pipelined schedule (ms[0] = the outer half, ms[1] = the inner one)
    idx_t hq1 = 0, hq2 = 0;
    const int plan_mode_used = (halves && halves_geometry(&hq1, &hq2)) ? 4 : -1;
    for (int r = -1; r < reps; r++) {          // r = -1: warm-up
        // (stage by stage as run() issues them; ms[0] = the exterior of the LAST stage, ms[0] + ms[1] = the whole step)
        YKH_HIP(hipEventRecord(e[0], compute_stream));
        if (all_planned) {
            // planned launches: ms[0] = from the start of the last stage's launch until its shell blocks have published their
            // epoch (a waiter on the comm stream, as in run()), ms[1] = from there to the end of the launch
            for (int st = 0; st < meta->n_stages; st++) {
                const StageMeta& sm = meta->stages[st];
                LaunchPlan* lp = get_launch_plan(planned_part(sm), has_lo, has_hi, false, plan_mode_used);
                launch_planned(planned_part(sm), r + 1, *lp, true, compute_stream);
                if (shell_event_pending) {
                    shell_event_pending = false;
                    YKH_HIP(hipStreamWaitEvent(comm_stream, ev_shell, 0));
                } else {              // (a plan without shell blocks: the launch itself)
                    YKH_HIP(hipEventRecord(ev_shell, compute_stream));
                    YKH_HIP(hipStreamWaitEvent(comm_stream, ev_shell, 0));
                }
                if (st == meta->n_stages - 1) YKH_HIP(hipEventRecord(e[1], comm_stream));
            }
            YKH_HIP(hipEventRecord(e[2], compute_stream));
            for (int st = 0; st < meta->n_stages; st++)
                for (int k = 0; k < meta->stages[st].n_parts; k++) launch_part(meta->stages[st].parts[k], r + 1, rb, compute_stream);
            YKH_HIP(hipEventRecord(e[3], compute_stream));
            YKH_HIP(hipEventSynchronize(e[3]));
            YKH_HIP(hipStreamSynchronize(comm_stream));
            if (r < 0) continue;
            for (int i = 0; i < 3; i++) { float m = 0; YKH_HIP(hipEventElapsedTime(&m, e[i], e[i + 1])); acc[i] += m; }
            continue;
        }
        for (int st = 0; st < meta->n_stages; st++) {
            const StageMeta& sm = meta->stages[st];
            launch_exterior(sm, r + 1, ib);
            if (st == meta->n_stages - 1) YKH_HIP(hipEventRecord(e[1], compute_stream));
            launch_interior(sm, r + 1, ib);
        }
        YKH_HIP(hipEventRecord(e[2], compute_stream));
        for (int st = 0; st < meta->n_stages; st++)
            for (int k = 0; k < meta->stages[st].n_parts; k++) launch_part(meta->stages[st].parts[k], r + 1, rb, compute_stream);
        YKH_HIP(hipEventRecord(e[3], compute_stream));
        YKH_HIP(hipEventSynchronize(e[3]));
        if (r < 0) continue;
        for (int i = 0; i < 3; i++) { float m = 0; YKH_HIP(hipEventElapsedTime(&m, e[i], e[i + 1])); acc[i] += m; }
    }
    for (int i = 0; i < 3; i++) ms[i] = acc[i] / (reps > 0 ? reps : 1);
    for (auto& x : e) (void)hipEventDestroy(x);
}

// ------------------------------------------------------------------ planned launches (ykh_plan.cpp plan_blocks)
void Solution::neighbor_sides(bool* has_lo, bool* has_hi) const {
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++) {
        has_lo[d] = d < ndd && env->nranks > 1 && rank_index[d] > 0;
        has_hi[d] = d < ndd && env->nranks > 1 && rank_index[d] < num_ranks[d] - 1;
    }
}
void Solution::drop_launch_plans() {
    for (auto& lp : launch_plans) if (lp && lp->dev) (void)hipFree(lp->dev);
    launch_plans.clear();
    planned_cache_.clear();
}
// A stage runs as a planned launch when it is ONE part on a marching kernel that reads block descriptors, over a plain
// 3-D box: no scratch children (they would have to be evaluated per block), no sub-domain box, no per-point predicate.
int Solution::planned_part(const StageMeta& sm) const {
    if (!planned_launch || ndd != 3 || has_outer || force_scalar || sm.n_parts != 1) return -1;
    const int part = sm.parts[0];
    const PartMeta& pm = *impl.parts[part].meta;
    if (pm.is_scratch || pm.has_step_cond || pm.has_step_cond_dev || part_needs_predicate(part)) return -1;
    if ((size_t)part < part_has_bb.size() && part_has_bb[part]) return -1;
    return planned_variant_of(part) >= 0 ? part : -1;
}
int Solution::planned_variant_of(int part) const {
    if (planned_cache_.size() != impl.parts.size()) planned_cache_.assign(impl.parts.size(), -2);
    if (planned_cache_[part] != -2) return planned_cache_[part];
    auto remember = [&](int r) { planned_cache_[part] = r; return r; };
    const PartImpl& pi = impl.parts[part];
    int v = part_variant[part];
    if (v < 0) return -1;
    // the part runs on its static default: the shape named for planned launches (same arithmetic, room for the twin's registers)
    if (variant_override.empty()) {
        if (v == pi.default_variant && pi.planned_variant >= 0) v = pi.planned_variant;
        else if (v == pi.exact_div_variant && pi.planned_exact_variant >= 0) v = pi.planned_exact_variant;
    }
    const KernelVariant& kv = pi.variants[v];
    if (!kv.launch_desc || !kv.star || kv.rx != 0) return remember(-1);
    if (kv.func_desc) {          // a twin that spilled registers is never worth it
        hipFuncAttributes at;
        if (hipFuncGetAttributes(&at, kv.func_desc) == hipSuccess) { if (at.localSizeBytes > 0) return remember(-1); }
        else (void)hipGetLastError();
    }
    return remember(v);
}
Solution::LaunchPlan* Solution::get_launch_plan(int part, const bool* has_lo, const bool* has_hi, bool wide_shell, int mode) {
    const KernelVariant& kv = impl.parts[part].variants[planned_variant_of(part)];
    if (mode < 0) mode = 0;
    std::ostringstream ks;
    ks << part << ':' << planned_variant_of(part) << '/' << local_size[0] << 'x' << local_size[1] << 'x' << local_size[2] << '/';
    for (int d = 0; d < 3; d++) ks << (has_lo[d] ? 'l' : '-') << (has_hi[d] ? 'h' : '-');
    ks << '/' << mode << '/' << min_exterior << '/' << env->num_cus << '/' << (wide_shell ? wf_ext_[0] + wf_ext_[1] * 1000 + wf_ext_[2] * 1000000 : 0);
    const std::string key = ks.str();
    for (auto& lp : launch_plans) if (lp->key == key) return lp.get();
    BlockPlanIn in;
    for (int d = 0; d < 3; d++) {
        in.n[d] = local_size[d]; in.has_lo[d] = has_lo[d]; in.has_hi[d] = has_hi[d];
        // (what a neighbour needs of my boundary: the halo -- plus the wave-front extension in a multi-rank -Mbt group)
        in.width[d] = std::max<idx_t>(std::max(shared_pad_l_[d], shared_pad_r_[d]) + (wide_shell ? wf_ext_[d] : 0), min_exterior);
    }
    in.ty = kv.ty; in.tz = kv.tz;
    in.overhead = kv.xover > 0 ? kv.xover : std::max<idx_t>(1, shared_pad_r_[0] + 1);
    in.ncu = std::max(1, env->num_cus);
    in.shell_frac = 0.55;           // the shell in the first of two rounds of equal blocks (35 %: three rounds, measured no better)
    in.mode = mode;
    auto lp = std::make_unique<LaunchPlan>();
    lp->key = key;
    try { lp->plan = plan_blocks(in); } catch (const PlanError& e) { YKH_THROW(e.what()); }
    if (lp->plan.blocks.empty()) YKH_THROW("planned launch: empty plan");
    YKH_HIP(hipMalloc(&lp->dev, lp->plan.blocks.size() * sizeof(BlockDesc)));
    YKH_HIP(hipMemcpyAsync(lp->dev, lp->plan.blocks.data(), lp->plan.blocks.size() * sizeof(BlockDesc), hipMemcpyHostToDevice, compute_stream));
    YKH_HIP(hipStreamSynchronize(compute_stream));
    if (env->trace)
        fprintf(stderr, "planned launch %s: %zu blocks (%lld signalling), simulated shell done at %lld, end at %lld, undivided %lld plane-iterations\n",
                key.c_str(), lp->plan.blocks.size(), (long long)lp->plan.n_signal, (long long)lp->plan.shell_done,
                (long long)lp->plan.makespan, (long long)lp->plan.undivided);
    {
        // where the shell ends: behind the last signalling block, rounded up to whole rounds of CUs (the blocks of a round end together)
        size_t last = 0;
        for (size_t i = 0; i < lp->plan.blocks.size(); i++) if (lp->plan.blocks[i].flags & BLOCK_SIGNALS) last = i + 1;
        const size_t ncu = (size_t)std::max(1, env->num_cus);
        lp->cut = std::min(lp->plan.blocks.size(), (last + ncu - 1) / ncu * ncu);
        if (lp->plan.mode_used == 4) lp->cut = (size_t)lp->plan.cut;       // the two halves: the planner's own cut
    }
    launch_plans.push_back(std::move(lp));
    return launch_plans.back().get();
}
void Solution::launch_planned(int part, idx_t t, LaunchPlan& lp, bool signal, hipStream_t s) {
    const KernelVariant& kv = impl.parts[part].variants[planned_variant_of(part)];
    if (lp.plan.mode_used == 4) {        // (the two halves back to back, without their exchanges: time_decomposed_step())
        launch_planned_half(part, t, lp, 0, s);
        if (signal) { YKH_HIP(hipEventRecord(ev_shell, s)); shell_event_pending = true; }
        launch_planned_half(part, t, lp, 1, s);
        return;
    }
    PartArgs a;
    fill_part_args(part, t, rank_box(), a);
    a.blk = lp.dev;
    if (signal && lp.plan.n_signal > 0) {
        // TWO launches, cut at the end of the round that holds the last shell block, with an event between them that releases the
        // exchange.  (Round 3 also had a one-launch form whose shell blocks raised a device-side signal that a resident one-wave
        // waiter polled: the waiter took a wave slot on some CU, a marching block needs ALL registers of a CU, and the 256th block of
        // a round then ran alone afterwards -- +0.1 ms on a 0.46 ms launch, profiles/r3_overlap.  Deleted in round 5 together with
        // the in-line pack between the two launches: the pack kernel fits in the 16 VGPRs a marching twin leaves.)
        kv.launch_desc(a, dim3((unsigned)lp.cut, 1, 1), s);
        YKH_HIP(hipGetLastError());
        YKH_HIP(hipEventRecord(ev_shell, s));
        shell_event_pending = true;
        if (lp.cut < lp.plan.blocks.size()) {
            a.blk = lp.dev + lp.cut;
            kv.launch_desc(a, dim3((unsigned)(lp.plan.blocks.size() - lp.cut), 1, 1), s);
            YKH_HIP(hipGetLastError());
        }
        return;
    }
    kv.launch_desc(a, dim3((unsigned)lp.plan.blocks.size(), 1, 1), s);
    YKH_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ pipelined half-exchanges (-hip_halves)
// The reference keeps its MPI requests moving while it computes (adv_halo_exchange, src/kernel/lib/halo.cpp:494-574, called per
// micro-block, context.cpp:1037-1040).  Planned launches (above) get their overlap from ORDER -- shell blocks first -- and pay for
// it: tiles that march without their neighbours fetch the shared lines twice (1.05-1.13x the undivided sweep, DESIGN.md section
// 4.1; the same blocks in regular order: 1.01-1.02x).  Here the order stays regular and the overlap comes from a software pipeline
// over half-launches H0, H1, H2, ... (outer half A = [0, q1) u [q2, nx), inner half B = [q1, q2), A, B, ...):
//     compute stream:  H_i            H_i+1                      H_i+2
//     comm stream:            E_i: pack, send ... arrive, unpack --^ (H_i+2 waits for E_i; H_i+1 does not)
// E_i carries what the neighbours need of H_i's planes: with face-only reads (l1_norm <= 1) the y / z halo a half reads lies in
// that half's own planes, and the x faces lie in A.  Who needs E_i?  The next launch over the same planes, H_i+2 -- one whole
// launch later.  Hazards: E_i's unpack writes halo cells of H_i's planes while H_i+1 runs; H_i+1 loads halo cells of its own
// planes only (the marching kernels load prologue / tail planes without their halos, or never use them in a stored value), and
// centre-only operands nowhere near a halo.  One exchange is in flight at a time: send / receive buffers and the transports'
// per-exchange state are those of the whole-face path.
bool Solution::halves_geometry(idx_t* q1, idx_t* q2) const {
    if (ndd != 3 || has_outer || wf_multi()) return false;
    return halves_split(local_size[0], std::max<idx_t>(1, std::max(shared_pad_l_[0], shared_pad_r_[0])), q1, q2);
}
bool Solution::halves_active() const {
    if (!halves || !halves_geom_ok_ || !overlap_comms || !planned_launch || env->nranks <= 1 || !do_halo_exchange) return false;
    if (std::max<idx_t>(mega_block_size[0], block_size[0]) > 1) return false;          // (wave-front groups exchange once per group)
    for (int st = 0; st < meta->n_stages; st++)
        if (planned_part(meta->stages[st]) < 0) return false;
    return true;
}
void Solution::launch_planned_half(int part, idx_t t, LaunchPlan& lp, int half, hipStream_t s) {
    const KernelVariant& kv = impl.parts[part].variants[planned_variant_of(part)];
    const size_t first = half == 0 ? 0 : lp.cut, count = half == 0 ? lp.cut : lp.plan.blocks.size() - lp.cut;
    if (count == 0) return;
    PartArgs a;
    fill_part_args(part, t, rank_box(), a);
    a.blk = lp.dev + first;
    kv.launch_desc(a, dim3((unsigned)count, 1, 1), s);
    YKH_HIP(hipGetLastError());
}
// behind the launch of `half` (ev_shell has been recorded there): the comm stream waits for it, packs and sends that half's faces
void Solution::halves_start(int half) {
    exch_half_ = half;
    shell_event_pending = true;
    exchange_halos(0, 0, /*start_only=*/true, false);
    exch_half_ = -1;
    halves_in_flight_ = true;
    halves_flight_half_ = half;
    halves_flight_phase_ = cur_phase;
}
// the exchange in flight: the comm stream waits for its messages and unpacks them, the compute stream waits for the unpack --
// i.e. whatever is launched AFTER this call sees the halos, what was launched before does not wait
void Solution::halves_finish() {
    if (!halves_in_flight_) return;
    PhaseEvents* mine = cur_phase;
    cur_phase = halves_flight_phase_;          // (transfer / unpack times belong to the phase set of the launch that produced the data)
    exch_half_ = halves_flight_half_;
    try { exchange_halos(0, 0, false, /*finish_only=*/true); } catch (...) { exch_half_ = -1; cur_phase = mine; halves_in_flight_ = false; throw; }
    exch_half_ = -1;
    cur_phase = mine;
    halves_in_flight_ = false;
    // (the inner half completes the faces: nothing is dirty any more -- also when this rank had no message in this exchange)
    if (halves_flight_half_ == 1)
        for (auto& v : vars) v->set_dirty_all(false);
}
void Solution::run_stage_halves(const StageMeta& sm, int /*st*/, idx_t t, const bool* has_lo, const bool* has_hi) {
    const int part = planned_part(sm);
    LaunchPlan* lp = get_launch_plan(part, has_lo, has_hi, false, /*mode=*/4);
    for (int h = 0; h < 2; h++) {
        cur_phase = phase_next();
        phase_mark(PH_EXT0, compute_stream);
        launch_planned_half(part, t, *lp, h, compute_stream);
        YKH_HIP(hipEventRecord(ev_shell, compute_stream));
        phase_mark(PH_INT1, compute_stream);
        halves_finish();                       // the PREVIOUS half's exchange (its data are first read by the launch after this one)
        phase_mark(PH_WAIT1, compute_stream);  // INT1 -> WAIT1: what the compute stream will idle behind this launch = exchange not hidden
        // (host bookkeeping, after the previous stage's flags have been cleared: what this stage writes is dirty for both halves)
        if (h == 0) note_stage_written(sm, t);
        halves_start(h);                       // (marks EXT1 on the comm stream, behind its wait for this launch: EXT0 -> EXT1 = the launch)
        cur_phase = nullptr;
    }
}

// ------------------------------------------------------------------ phase timers
// The reference times halo pack / unpack / wait and exterior / interior evaluation with host timers
// (src/kernel/lib/context.hpp:319-328); here the phases are asynchronous, so each (step, stage) of a multi-rank run
// gets HIP events on the stream the phase runs on; they are read once run() has drained the streams.
Solution::PhaseEvents* Solution::phase_next() {
    if (!phase_timers) return nullptr;
    // A ring: run_solution(0, 99999) must not create 800 000 events.  Set number k of a run lives in slot k % PHASE_RING; before
    // a slot is reused its times are folded into `stats` (its events are PHASE_RING stages old: the wait is almost never one).
    const size_t slot = phase_used % PHASE_RING;
    phase_pool.reserve(PHASE_RING);          // (sets are handed out by pointer, and the halves schedule holds two at a time: never re-allocate)
    if (slot == phase_pool.size()) {
        PhaseEvents ph;
        for (int i = 0; i < PH_N; i++) { ph.e[i] = nullptr; ph.rec[i] = false; }
        for (int i = 0; i < PH_N; i++) YKH_HIP(hipEventCreate(&ph.e[i]));
        phase_pool.push_back(ph);
    } else if (phase_used >= PHASE_RING) {
        PhaseEvents& old = phase_pool[slot];
        for (int i = 0; i < PH_N; i++) if (old.rec[i]) YKH_HIP(hipEventSynchronize(old.e[i]));
        phase_fold(old);
    }
    phase_used++;
    PhaseEvents* ph = &phase_pool[slot];
    for (int i = 0; i < PH_N; i++) ph->rec[i] = false;
    return ph;
}
void Solution::phase_mark(int which, hipStream_t st) {
    if (!cur_phase) return;
    YKH_HIP(hipEventRecord(cur_phase->e[which], st));
    cur_phase->rec[which] = true;
}
void Solution::phase_fold(const PhaseEvents& ph) {
    auto span = [&](int a, int b) -> double {
        if (!ph.rec[a] || !ph.rec[b]) return 0.0;
        float ms = 0;
        if (hipEventElapsedTime(&ms, ph.e[a], ph.e[b]) != hipSuccess) { (void)hipGetLastError(); return 0.0; }
        return ms > 0 ? ms * 1e-3 : 0.0;
    };
    stats.exterior_secs += span(PH_EXT0, PH_EXT1);
    stats.interior_secs += span(PH_EXT1, PH_INT1);
    stats.halo_wait_secs += span(PH_INT1, PH_WAIT1);
    const double pack = span(PH_PACK0, PH_PACK1), xfer = span(PH_PACK1, PH_XFER1), unpack = span(PH_XFER1, PH_UNPACK1);
    stats.halo_pack_secs += pack; stats.halo_xfer_secs += xfer; stats.halo_unpack_secs += unpack;
    stats.halo_secs += pack + xfer + unpack;
}
// (called when run() has drained the streams: every set still in the ring is complete)
void Solution::phase_collect() {
    const size_t live = std::min<size_t>(phase_used, phase_pool.size());
    for (size_t i = 0; i < live; i++) phase_fold(phase_pool[i]);
    for (auto& ph : phase_pool) for (int i = 0; i < PH_N; i++) ph.rec[i] = false;
    phase_used = 0;
}

}  // namespace ykh
