// ykh_halo.cpp -- per-GPU domain decomposition: halo pack -> transport -> unpack.
//
// GPU re-design of StencilContext::exchange_halos (src/kernel/lib/halo.cpp:80-491) and of the
// buffer geometry computed by alloc_mpi_data (src/kernel/lib/alloc.cpp:456-859):
//   * neighbours: up to 3^N-1, pruned per var by its L1 norm (faces only when l1_norm == 1);
//   * what is sent to the neighbour at offset o: in each dim d with o[d] = -1 my first halo_r[d]
//     domain points (they fill the neighbour's right halo), with o[d] = +1 my last halo_l[d] points,
//     with o[d] = 0 my whole extent in d -- extended into my own halo at a *global* boundary when the
//     var is read diagonally (l1_norm > 1), as alloc.cpp:544-553 does, so that corner cells end up
//     identical to a single-rank run;
//   * only (var, step-slot) pairs marked dirty are exchanged (src/kernel/lib/yk_var.cpp:122-152);
//   * unlike the reference (one MPI message per var per neighbour, tag = var ordinal) all dirty slabs
//     for one neighbour travel in ONE message: xGMI is point-to-point, so fewer/larger transfers win;
//   * pack and unpack are box-copy kernels on the communication stream; the transport is stream
//     ordered (RCCL ncclSend/ncclRecv inside one group, or a host callback), so run() can overlap the
//     whole exchange with the interior kernel on the compute stream.
#include <algorithm>
#include <cstring>

#include "ykh_runtime.hpp"

namespace ykh {

static bool slab_for(const Solution& s, const Var& v, const Solution::Neighbor& nb, bool sending, Slab& out) {
    if (v.fixed_size || !v.is_allocated()) return false;
    VarGeom g;
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++) {
        g.uses_domain[d] = v.uses_domain[d]; g.dom_size[d] = v.dom_size[d];
        g.halo_l[d] = v.halo_l[d]; g.halo_r[d] = v.halo_r[d];
    }
    g.l1_norm = v.l1_norm;
    if (s.wf_multi()) {
        // wave-front tiling across ranks: every var a phase may read on its extended box needs its neighbours' data
        // wf_ext further out -- vars without a halo included -- and the extended boxes have edges and corners
        for (int d = 0; d < MAX_DOMAIN_DIMS; d++) g.wext[d] = v.uses_domain[d] ? s.wf_ext(d) : 0;
        g.l1_norm = MAX_DOMAIN_DIMS;
    }
    PlanNeighbor pn;
    pn.rank = nb.rank; pn.l1 = nb.l1;
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++) pn.ofs[d] = nb.ofs[d];
    if (!plan_halo_slab(s.ndd, s.num_ranks, s.rank_index, g, pn, sending, out.lo, out.n)) return false;
    out.elems = out.n[0] * out.n[1] * out.n[2] * v.misc_elems;
    return true;
}

void Solution::alloc_halo_buffers() {
    halo_built_direct_ok = env->direct_halo_ok;
    if (env->nranks <= 1) return;
    for (auto& nb : neighbors) {
        auto x = std::make_unique<NeighborXfer>();
        x->nb = nb;
        size_t sb = 0, rb = 0;
        for (size_t i = 0; i < vars.size(); i++) {
            Slab sl;
            sl.var = (int)i;
            if (slab_for(*this, *vars[i], nb, true, sl)) { x->send.push_back(sl); sb += (size_t)sl.elems * vars[i]->nslots; }
            if (slab_for(*this, *vars[i], nb, false, sl)) { x->recv.push_back(sl); rb += (size_t)sl.elems * vars[i]->nslots; }
        }
        x->send_cap = sb * elem_bytes();
        x->recv_cap = rb * elem_bytes();
        // the same slabs cut at the planes of the two x-halves (pipelined half-exchanges, Solution::run_stage_halves): both ends of a
        // link cut at the same planes (q1, q2 depend on the x extent alone, which y / z neighbours share)
        idx_t q1 = 0, q2 = 0;
        if (halves_geometry(&q1, &q2))
            for (int dirn = 0; dirn < 2; dirn++) {
                const std::vector<Slab>& whole = dirn ? x->recv : x->send;
                for (int h = 0; h < 2; h++) {
                    std::vector<Slab>& part = dirn ? x->recv_h[h] : x->send_h[h];
                    for (const Slab& sl : whole) {
                        idx_t l[2], m[2];
                        const int k = halves_slab_ranges(h, nb.ofs[0] != 0, sl.lo[0], sl.n[0], q1, q2, l, m);
                        for (int i = 0; i < k; i++) {
                            Slab c = sl;
                            c.lo[0] = l[i]; c.n[0] = m[i];
                            c.elems = c.n[0] * c.n[1] * c.n[2] * vars[sl.var]->misc_elems;
                            part.push_back(c);
                        }
                    }
                }
            }
        // in-place transfer? (decided from geometry only, so both ends of a link agree)
        x->direct = direct_halo && env->direct_halo_ok && !wf_multi() && ndd == 3 && nb.ofs[0] != 0 && nb.ofs[1] == 0 && nb.ofs[2] == 0 &&
                    x->send.size() == x->recv.size() && !x->send.empty();
        for (auto* lst : {&x->send, &x->recv})
            for (const Slab& sl : *lst) {
                const Var& v = *vars[sl.var];
                if (!(v.uses_domain[0] && v.uses_domain[1] && v.uses_domain[2]) || v.l1_norm > 1 || v.misc_elems != 1)
                    x->direct = false;
                // an in-place message is named (HaloMsg::key) by solution ordinal, var and step slot in fixed fields of 16 slots and
                // 255 vars: anything wider takes the packed path (one message per neighbour) instead of a colliding key
                if (v.nslots > 16 || sl.var >= 255) x->direct = false;
            }
        for (size_t i = 0; x->direct && i < x->send.size(); i++)
            if (x->send[i].var != x->recv[i].var) x->direct = false;      // asymmetric halos: keep the packed path
        if (x->direct) { xfers.push_back(std::move(x)); continue; }
        if (x->send_cap) YKH_HIP(hipMalloc(&x->send_buf, x->send_cap));
        if (x->recv_cap) YKH_HIP(hipMalloc(&x->recv_buf, x->recv_cap));
        if (x->send_cap || x->recv_cap) xfers.push_back(std::move(x));
    }
}

void Solution::free_halo_buffers() {
    if (env && env->exch_reset && !xfers.empty()) env->exch_reset(env->user);       // (a transport may hold mappings of these buffers)
    for (auto& x : xfers) {
        if (x->send_buf) (void)hipFree(x->send_buf);
        if (x->recv_buf) (void)hipFree(x->recv_buf);
    }
    xfers.clear();
}

// Every dirty (var, slot) slab of `slabs`, laid out one after the other in the contiguous buffer: appended to `segs`
// (launch_halo_move moves all segments of an exchange in one launch); returns the bytes of the message.
static size_t collect_slabs(Solution& s, const std::vector<Slab>& slabs, void* buf, std::vector<HaloSeg>& segs) {
    size_t ofs = 0;
    const int eb = s.elem_bytes();
    for (const Slab& sl : slabs) {
        Var& v = *s.vars[sl.var];
        for (int slot = 0; slot < v.nslots; slot++) {
            if (!v.dirty[slot]) continue;
            // misc indices are laid out outside the domain dims: one segment per misc plane
            std::vector<idx_t> mofs = {0};
            for (size_t p = 0; p < v.dims.size(); p++) {
                if (v.dims[p].type != DIM_MISC) continue;
                std::vector<idx_t> nxt;
                for (idx_t m = 0; m <= v.dims[p].last_misc - v.dims[p].first_misc; m++)
                    for (idx_t b : mofs) nxt.push_back(b + m * v.misc_stride[p]);
                mofs.swap(nxt);
            }
            for (idx_t mo : mofs) {
                HaloSeg h;
                h.var_base = (char*)v.dptr + ((size_t)slot * v.slot_elems + v.origin_elems + mo) * eb;
                h.buf = (char*)buf + ofs;
                h.sx = v.stride[0]; h.sy = v.stride[1]; h.sz = v.stride[2];
                for (int d = 0; d < 3; d++) { h.lo[d] = (int)sl.lo[d]; h.n[d] = (int)sl.n[d]; }
                segs.push_back(h);
                ofs += (size_t)(sl.n[0] * sl.n[1] * sl.n[2]) * eb;
            }
        }
    }
    return ofs;
}

// The messages of one exchange and their packing: every dirty (var, slot) slab of every neighbour goes into that neighbour's send
// buffer (one vectorised launch for all of them, on `st`); x faces of "direct" neighbours travel straight from the var's planes.
void Solution::exchange_build_and_pack(hipStream_t st) {
    std::vector<HaloMsg>& msgs = pending_msgs;
    std::vector<HaloSeg> segs;
    const int half = exch_half_;          // -1: whole faces; 0 / 1: the slabs of one x-half (pipelined half-exchanges)
    for (auto& x : xfers) {
        if (x->direct) {
            // one message per dirty (var, slot): whole planes (with their y/z pads) straight from / into the var
            x->send_now = x->recv_now = 0;
            if (half == 1) continue;      // (x faces travel with the outer half)
            for (size_t i = 0; i < x->send.size(); i++) {
                const Slab &ss = x->send[i], &rs = x->recv[i];
                Var& v = *vars[ss.var];
                for (int slot = 0; slot < v.nslots; slot++) {
                    if (!v.dirty[slot]) continue;
                    auto plane_ptr = [&](idx_t xl) {
                        return (char*)v.dptr + ((size_t)slot * v.slot_elems + v.origin_elems + xl * v.stride[0] -
                                                v.pad_l[1] * v.stride[1] - v.pad_l[2] * v.stride[2]) * elem_bytes();
                    };
                    HaloMsg m;
                    m.peer = x->nb.rank;
                    m.send_buf = plane_ptr(ss.lo[0]); m.send_bytes = (size_t)(ss.n[0] * v.stride[0]) * elem_bytes();
                    m.recv_buf = plane_ptr(rs.lo[0]); m.recv_bytes = (size_t)(rs.n[0] * v.stride[0]) * elem_bytes();
                    m.tag = (x->nb.ofs[0] + 1) * 9 + 4;
                    m.key = ordinal * 4096 + (ss.var * 16 + slot) + 1;
                    msgs.push_back(m);
                    x->send_now += m.send_bytes; x->recv_now += m.recv_bytes;
                }
            }
            continue;
        }
        x->send_now = collect_slabs(*this, half < 0 ? x->send : x->send_h[half], x->send_buf, segs);
        // receive size: same rule evaluated on my recv slabs (neighbour's dirty flags mirror mine)
        size_t r = 0;
        for (const Slab& sl : (half < 0 ? x->recv : x->recv_h[half])) {
            Var& v = *vars[sl.var];
            for (int slot = 0; slot < v.nslots; slot++)
                if (v.dirty[slot]) r += (size_t)sl.elems * elem_bytes();
        }
        x->recv_now = r;
        if (!x->send_now && !x->recv_now) continue;
        HaloMsg m;
        m.peer = x->nb.rank;
        m.send_buf = x->send_buf; m.recv_buf = x->recv_buf;
        m.send_bytes = x->send_now; m.recv_bytes = x->recv_now;
        // tag encodes the direction so that both messages between a pair of ranks are distinct
        m.tag = (x->nb.ofs[0] + 1) * 9 + (x->nb.ofs[1] + 1) * 3 + (x->nb.ofs[2] + 1);
        m.key = ordinal * 4096;
        msgs.push_back(m);
    }
    launch_halo_move(segs, /*pack=*/true, elem_bytes(), st);
}

// start_only : pack + begin transport on the comm stream (after everything queued on the compute stream);
// finish_only: wait for arrival, unpack, and make the compute stream wait for the unpack.
void Solution::exchange_halos(idx_t /*t_written*/, int /*stage*/, bool start_only, bool finish_only) {
    if (env->nranks <= 1 || xfers.empty()) return;
    if (!env->exch_start) YKH_THROW("multi-rank solution has no halo-exchange transport installed in its env");
    std::vector<HaloMsg>& msgs = pending_msgs;      // messages between a start and the matching finish
    if (!finish_only) {
        const bool evented = shell_event_pending;
        shell_event_pending = false;
        {
            bool any = false;
            for (auto& v : vars)
                for (char d : v->dirty) any |= (d != 0);
            msgs.clear();
            if (!any) return;
            if (evented) {
                // planned launch in two parts: the comm stream waits for the event behind the shell's rounds
                YKH_HIP(hipStreamWaitEvent(comm_stream, ev_shell, 0));
                phase_mark(PH_EXT1, comm_stream);
            } else {
                // comm stream waits for the kernels that produced the data
                YKH_HIP(hipEventRecord(ev_a, compute_stream));
                YKH_HIP(hipStreamWaitEvent(comm_stream, ev_a, 0));
            }
            phase_mark(PH_PACK0, comm_stream);
            exchange_build_and_pack(comm_stream);
            phase_mark(PH_PACK1, comm_stream);
        }
        for (const HaloMsg& m : msgs) {
            stats.halo_bytes_sent += (idx_t)m.send_bytes; stats.halo_bytes_recv += (idx_t)m.recv_bytes;
            stats.halo_msgs_sent += m.send_bytes ? 1 : 0;
        }
        if (!msgs.empty() && env->exch_start(env->user, (int)msgs.size(), msgs.data(), (void*)comm_stream) != 0)
            YKH_THROW("halo-exchange transport failed to start");
    }
    if (!start_only) {
        if (msgs.empty()) return;
        if (env->exch_wait && env->exch_wait(env->user, (int)msgs.size(), msgs.data(), (void*)comm_stream) != 0)
            YKH_THROW("halo-exchange transport failed while waiting");
        phase_mark(PH_XFER1, comm_stream);
        std::vector<HaloSeg> segs;
        for (auto& x : xfers)
            if (x->recv_now && !x->direct) collect_slabs(*this, exch_half_ < 0 ? x->recv : x->recv_h[exch_half_], x->recv_buf, segs);
        launch_halo_move(segs, /*pack=*/false, elem_bytes(), comm_stream);
        phase_mark(PH_UNPACK1, comm_stream);
        YKH_HIP(hipEventRecord(ev_b, comm_stream));
        YKH_HIP(hipStreamWaitEvent(compute_stream, ev_b, 0));
        // (the outer half's exchange leaves the flags up: the inner half's part of the same faces has yet to travel)
        if (exch_half_ != 0)
            for (auto& v : vars) v->set_dirty_all(false);
        msgs.clear();
    }
}

void Solution::exchange_halos_all() {
    if (!prepared) YKH_THROW("exchange_halos() called without calling prepare_solution() first");
    // Dirty flags are per rank: another rank may have changed data through the API (set_element ...) while this
    // one did not.  Every rank must post the same messages, so everything counts as possibly dirty -- the
    // reference's set_all_neighbor_vars_dirty() (context.cpp:234, halo.cpp:84-161 keeps self/others flags).
    for (auto& v : vars) v->before_device_use();      // raw buffers handed out: the caller may have written through them
    exch_half_ = -1;                                   // whole faces
    if (env->nranks > 1 && halo_built_direct_ok != env->direct_halo_ok) {
        // a transport was installed after prepare_solution() and it differs in whether peers may map var storage: the in-place
        // x transfers were decided for the old one (every rank installs the same transport, so every rank rebuilds here)
        free_halo_buffers();
        alloc_halo_buffers();
        drop_step_graphs();
        drop_launch_plans();
    }
    if (env->nranks > 1 && !xfers.empty() && env->exch_begin && env->exch_begin(env->user) != 0)
        YKH_THROW("halo-exchange transport failed to agree on its state across the ranks");
    if (env->nranks > 1)
        for (auto& v : vars) v->set_dirty_all(true);
    exchange_halos(0, 0, true, false);
    exchange_halos(0, 0, false, true);
    if (env->nranks > 1)
        for (auto& v : vars) v->after_device_write();
}

}  // namespace ykh
