// ykh_temporal.cpp -- more than one time step per sweep (part of class Solution, split off ykh_solution.cpp in round 4): the
// two-steps-per-pass kernel driver (-hip_fuse_steps), wave-front tiling at launch granularity (-Mbt / -bt; the reference's
// calc_mega_block / shift_mega_block, src/kernel/lib/context.cpp:482-745,1181-1525) and wave-fronts across ranks (setup.cpp:717-805).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "ykh_runtime.hpp"
#include "ykh_solution_internal.hpp"

namespace ykh {

// ------------------------------------------------------------------ two steps per pass, fused on chip
// Temporal blocking where the chip can hold it (ykh_starlin2.hpp): S(t+2) is computed from S(t) in
// one sweep, S(t+1) stays in registers / LDS.  The kernel writes out of place, so the passes alternate between the
// var's own slot and a scratch slot; the scratch starts as a copy of the slot (its pads are the slot's pads, which a
// single rank never updates) and the layout is restored at the end.  The LAST pass also stores S(t+1), so that after
// run_solution() both step slots hold what a plain run leaves there.
bool Solution::can_fuse() const {
    if (fuse_steps < 2 || env->nranks != 1 || ndd != 3 || force_scalar) return false;
    if (impl.parts.size() != 1 || !impl.parts[0].fused2.launch || meta->n_stages != 1) return false;
    const PartMeta& pm = *impl.parts[0].meta;
    for (auto& v : vars)
        if (v->meta == &meta->vars[pm.groups[0].var]) return v->nslots == 2 && v->is_allocated();
    return false;
}

void Solution::launch_fused(idx_t t, const void* src, void* slot_b, void* dst, bool store_b) {
    const Fused2Variant& f = impl.parts[0].fused2;
    const Box rb = rank_box();
    PartArgs a;
    fill_part_args(0, t, rb, a);
    a.ptr[0] = const_cast<void*>(src);
    a.ptr[1] = slot_b;
    a.ptr[2] = dst;
    const idx_t zb = rb.lo[2] & ~(idx_t)(f.vz - 1);
    a.ntz = (int)ceil_div(rb.hi[2] - zb, f.tzi);
    a.nty = (int)ceil_div(rb.hi[1] - rb.lo[1], f.tyi);
    const idx_t nx = rb.hi[0] - rb.lo[0], tiles = (idx_t)a.ntz * a.nty, cus = std::max(1, env->num_cus);
    // x-chunks: fill the CUs in whole rounds; every chunk runs 4*xr+1 extra planes to fill the two pipelines
    idx_t best_n = 1;
    double best_eff = -1;
    const idx_t extra = 4 * f.xr + 1;
    for (idx_t n = 1; n <= 32; n++) {
        idx_t len = ceil_div(nx, n);
        if (n > 1 && len < 48) break;
        idx_t blocks = tiles * ceil_div(nx, len);
        double fill = (double)blocks / (double)(ceil_div(blocks, cus) * cus);
        double eff = fill * (double)len / (double)(len + extra);
        if (eff > best_eff * 1.02) { best_eff = eff; best_n = n; }
    }
    idx_t xc = xchunk_override > 0 ? xchunk_override : ceil_div(nx, best_n);
    xc = std::max<idx_t>(1, std::min(xc, nx));
    a.xchunk = (int)xc;
    a.nxc = (int)ceil_div(nx, xc);
    f.launch(a, dim3((unsigned)((idx_t)a.ntz * a.nty * a.nxc)), compute_stream, store_b);
    YKH_HIP(hipGetLastError());
}

void Solution::run_fused(idx_t t0, idx_t npairs, idx_t dir) {
    const PartMeta& pm = *impl.parts[0].meta;
    Var* v = nullptr;
    for (auto& vv : vars) if (vv->meta == &meta->vars[pm.groups[0].var]) v = vv.get();
    if (!v) YKH_THROW("fused run: var not found");
    const size_t slot_bytes = (size_t)v->slot_elems * elem_bytes(), org = (size_t)v->origin_elems * elem_bytes();
    if (!v->scratch) YKH_HIP(hipMalloc(&v->scratch, slot_bytes));
    char* slot_a = (char*)v->dptr + (size_t)v->slot_of(t0) * slot_bytes;            // holds S(t0), S(t0+2), ...
    char* slot_b = (char*)v->dptr + (size_t)v->slot_of(t0 + dir) * slot_bytes;      // the in-between steps' slot
    // The passes alternate between the slot and the scratch and must END in the slot.  Even number of passes: the scratch
    // only needs the slot's pads (its domain is overwritten by the first pass).  Odd: the scratch becomes a full copy of
    // the slot and the FIRST pass reads the copy and writes into the slot -- no copy back at the end either way.
    idx_t alloc[3], padl[3], dom[3];
    for (int d = 0; d < 3; d++) { padl[d] = v->pad_l[d]; dom[d] = v->dom_size[d]; alloc[d] = v->pad_l[d] + v->dom_size[d] + v->pad_r[d]; }
    char* cur = slot_a;
    char* other = (char*)v->scratch;
    if (npairs % 2 == 0) launch_copy_pads(slot_a, v->scratch, elem_bytes(), alloc, padl, dom, compute_stream);
    else {
        YKH_HIP(hipMemcpyAsync(v->scratch, slot_a, slot_bytes, hipMemcpyDeviceToDevice, compute_stream));
        std::swap(cur, other);
    }
    for (idx_t k = 0; k < npairs; k++) {
        const idx_t t = t0 + dir * 2 * k;
        launch_fused(t, cur + org, slot_b + org, other + org, /*store_b=*/k == npairs - 1);
        std::swap(cur, other);
        stats.fused_passes++;
        v->update_valid_step(t + dir); v->update_valid_step(t + 2 * dir);
    }
    if (cur != slot_a) YKH_THROW("fused run: internal error (result is not in the var's slot)");
    v->set_dirty_all(true);
}

// ------------------------------------------------------------------ wave-front temporal tiling
// The reference's temporal wave-fronts (StencilContext::calc_mega_block / shift_mega_block, src/kernel/lib/context.cpp:
// 482-745,1181-1525; angles from setup.cpp:863-1020) at launch granularity: the rank is cut into x-slabs; for one slab
// after the other, phase p = (step, stage) number p of the group is evaluated on the slab shifted by -p * angle, where
// angle = the widest x-halo of the solution.  Phase p+1 then finds every input it reads at x +- halo already computed by
// phase p (this slab or an earlier one), and what it overwrites in place (2-slot write-back, 1-slot in-place vars) is
// no longer needed by phase p of the next slab, whose reads start at x1 - p*angle - halo >= x1 - (p+1)*angle.
// Every launch is an ordinary kernel launch over a box, so the result is bit-identical to plain sweeps.
void Solution::run_wavefront(idx_t t0, idx_t nsteps, idx_t dir) {
    const Box rb = rank_box();
    const idx_t nx = rb.hi[0] - rb.lo[0];
    const idx_t ang = std::max<idx_t>(1, std::max(shared_pad_l_[0], shared_pad_r_[0]));
    const idx_t nphase = nsteps * meta->n_stages;
    idx_t w = mega_block_size[1] > 0 ? mega_block_size[1] : 128;
    w = std::max<idx_t>(w, ang);
    for (const WavefrontLaunch& wl : plan_wavefront(rb.lo[0], rb.hi[0], w, ang, nphase)) {       // ykh_plan.cpp
        const idx_t t = t0 + dir * (wl.phase / meta->n_stages);
        const StageMeta& sm = meta->stages[wl.phase % meta->n_stages];
        Box b = rb;
        b.lo[0] = wl.lo;
        b.hi[0] = wl.hi;
        for (int k = 0; k < sm.n_parts; k++) launch_part(sm.parts[k], t, b, compute_stream);
    }
    (void)nx;
    // bookkeeping once per step: written vars become valid at the output step
    for (idx_t s = 0; s < nsteps; s++) {
        const idx_t t = t0 + dir * s;
        for (int st = 0; st < meta->n_stages; st++) {
            const StageMeta& sm = meta->stages[st];
            for (int k = 0; k < sm.n_parts; k++) {
                const PartMeta& pm = *impl.parts[sm.parts[k]].meta;
                if (pm.is_scratch || (pm.step_cond && !pm.step_cond(t))) continue;
                for (int wv = 0; wv < pm.n_writes; wv++) {
                    const AccessGroup& ag = pm.groups[pm.writes[wv]];
                    for (auto& v : vars)
                        if (v->meta == &meta->vars[ag.var]) {
                            if (ag.has_step) { v->update_valid_step(t + ag.dt); v->set_dirty(true, t + ag.dt); }
                            else v->set_dirty_all(true);
                        }
                }
            }
        }
    }
}

// ------------------------------------------------------------------ wave-front tiling across ranks
// The reference's temporal wave-fronts work across ranks by extending every rank's evaluation into its neighbours' domains
// (left/right_wf_exts = angle x shifts, setup.cpp:717-805; the diagram in run_solution, context.cpp:286-346): the points
// between "end (rank)" and "end (ext)" are computed by both ranks, and halos are exchanged once per group of -Mbt steps.
// Here: a group of n steps is P = n x stages phases; phase p runs the stage's parts over the rank box grown by
// wf_angle x (P-1-p) towards every side that has a neighbour.  Phase p+1's reads (reach: the halo = the angle) then lie
// inside what phase p computed, the last phase covers exactly the rank box, and nothing has to travel inside the group.
// The widened halos (wf_ext = angle x (P-1) beyond the stencil halo, all 26 neighbours: extended boxes have edges and
// corners) are exchanged once per group -- half the messages per step at n = 2, against ~(1 + ext/size)^3 of redundant
// arithmetic.  Every launch is an ordinary launch of the same kernel: bit-identical to plain multi-rank sweeps.
// The last phase runs as a planned launch where the stage allows it (shell = what the neighbours need, now halo + wf_ext
// wide, first; the exchange starts while the interior is still being computed).
void Solution::run_wavefront_multi(idx_t t0, idx_t nsteps, idx_t dir, const bool* has_lo, const bool* has_hi, bool exchange) {
    const idx_t nphase = nsteps * meta->n_stages;
    const Box rb = rank_box();
    for (idx_t p = 0; p < nphase; p++) {
        const idx_t t = t0 + dir * (p / meta->n_stages);
        const int st = (int)(p % meta->n_stages);
        const StageMeta& sm = meta->stages[st];
        const idx_t e = nphase - 1 - p;
        Box b = rb;
        for (int d = 0; d < ndd; d++) {
            if (wf_ext_[d] <= 0) continue;
            if (has_lo[d]) b.lo[d] -= wf_angle_[d] * e;
            if (has_hi[d]) b.hi[d] += wf_angle_[d] * e;
        }
        const bool last = p == nphase - 1;
        const int pl_part = (last && exchange && overlap_comms) ? planned_part(sm) : -1;
        cur_phase = (exchange && last) ? phase_next() : nullptr;
        if (pl_part >= 0) {
            LaunchPlan* lp = get_launch_plan(pl_part, has_lo, has_hi, /*wide_shell=*/true);
            phase_mark(PH_EXT0, compute_stream);
            launch_planned(pl_part, t, *lp, /*signal=*/true, compute_stream);
        } else {
            phase_mark(PH_EXT1, compute_stream);
            for (int k = 0; k < sm.n_parts; k++) launch_part(sm.parts[k], t, b, compute_stream);
        }
        note_stage_written(sm, t);
        if (last && exchange) {
            exchange_halos(t, st, /*start_only=*/true, false);
            phase_mark(PH_INT1, compute_stream);
            exchange_halos(t, st, false, /*finish_only=*/true);
            phase_mark(PH_WAIT1, compute_stream);
        }
        cur_phase = nullptr;
    }
}

}  // namespace ykh
