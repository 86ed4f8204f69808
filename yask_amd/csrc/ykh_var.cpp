// ykh_var.cpp -- device-resident YASK var.
//
// GPU re-design of YkVarBase/YkVarImpl + GenericVar (src/kernel/lib/yk_var.{hpp,cpp},
// yk_var_apis.cpp, generic_var.{hpp,cpp}): same API semantics (global indices, valid-step window,
// pads >= halos, strict / non-strict slices, dirty flags), but storage is a single hipMalloc'd
// array per var laid out [step slot][misc..][x][y][z] with z unit-stride and 256-byte aligned rows,
// so that a wavefront reads whole cache lines and 16-byte vector accesses are always aligned.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <sstream>

#include "ykh_runtime.hpp"

namespace ykh {

static inline idx_t imod_flr(idx_t a, idx_t b) { idx_t m = a % b; return m < 0 ? m + b : m; }
static inline idx_t round_up(idx_t v, idx_t m) { return (v + m - 1) / m * m; }

Var::Var(Solution* s, const VarMeta* m, int ord) : soln(s), meta(m), name(m->name), ordinal(ord) {
    init_dims_from_meta();
}

void Var::init_dims_from_meta() {
    const SolnMeta* sm = soln->meta;
    dims.clear();
    for (int i = 0; i < meta->ndims; i++) {
        const DimMeta& d = sm->dims[meta->dims[i]];
        VarDim vd;
        vd.name = d.name;
        vd.type = d.type;
        vd.domain_idx = d.domain_idx;
        if (d.type == DIM_STEP) { has_step = true; step_posn = i; }
        if (d.type == DIM_MISC) { vd.first_misc = meta->misc_first[i]; vd.last_misc = meta->misc_last[i]; }
        if (d.type == DIM_OUTER) {       // stored like a misc dim; the index range (pads included) comes with the sizes
            vd.type = DIM_MISC; vd.is_outer = true; vd.domain_idx = -1;
            vd.outer_halo_l = meta->outer_halo_l; vd.outer_halo_r = meta->outer_halo_r;
        }
        if (d.type == DIM_DOMAIN) {
            uses_domain[d.domain_idx] = true;
            halo_l[d.domain_idx] = meta->halo_l[d.domain_idx];
            halo_r[d.domain_idx] = meta->halo_r[d.domain_idx];
        }
        dims.push_back(vd);
    }
    nslots = has_step ? std::max(1, meta->step_alloc) : 1;
    l1_norm = meta->l1_norm;
    is_written = meta->is_written;
    dirty.assign(nslots, 0);
}

// User-created var (yk_solution::new_var / new_fixed_size_var, src/kernel/lib/new_var.cpp).
Var::Var(Solution* s, const std::string& nm, const std::vector<std::string>& dnames, int ord,
         const std::vector<idx_t>* sizes)
    : soln(s), meta(nullptr), name(nm), ordinal(ord) {
    if (sizes) {
        fixed_size = true;
        fixed_sizes = *sizes;
        if (sizes->size() != dnames.size())
            YKH_THROW("attempting to create var '" + nm + "' with " + std::to_string(dnames.size()) +
                      " dimension names but " + std::to_string(sizes->size()) + " dimension sizes");
    }
    for (size_t i = 0; i < dnames.size(); i++) {
        for (size_t j = 0; j < i; j++)
            if (dnames[i] == dnames[j])
                YKH_THROW("cannot create var '" + nm + "': dimension '" + dnames[i] + "' used more than once");
        VarDim vd;
        vd.name = dnames[i];
        vd.domain_idx = -1;
        if (dnames[i] == s->step_dim_name) {
            vd.type = DIM_STEP;
            has_step = true;
            step_posn = (int)i;
        } else {
            vd.type = DIM_MISC;
            for (int d = 0; d < s->ndd; d++)
                if (s->domain_dim_names[d] == dnames[i]) { vd.type = DIM_DOMAIN; vd.domain_idx = d; uses_domain[d] = true; }
            if (vd.type == DIM_MISC && s->has_outer && dnames[i] == s->outer_dim_name && !sizes) vd.is_outer = true;
            if (vd.type == DIM_MISC) {
                vd.first_misc = 0;
                vd.last_misc = sizes ? (*sizes)[i] - 1 : 0;
            }
        }
        dims.push_back(vd);
    }
    nslots = 1;
    if (has_step) nslots = sizes ? (int)std::max<idx_t>(1, (*sizes)[step_posn]) : 1;
    dynamic_step_alloc = true;
    dirty.assign(nslots, 0);
}

Var::~Var() {
    // leave the fuse group; the storage lives on while another var of the group holds it
    if (fuse_group) {
        auto& g = *fuse_group;
        g.erase(std::remove(g.begin(), g.end(), this), g.end());
        fuse_group.reset();
    }
    drop_storage_refs();
}

int Var::elem_bytes() const { return soln->elem_bytes(); }

void Var::compute_geometry() {
    const int eb = elem_bytes();
    const idx_t zal = 256 / eb;   // innermost-row alignment in elements (256 B)
    // innermost used domain dim gets the aligned pitch
    int inner = -1;
    for (int d = soln->ndd - 1; d >= 0; d--)
        if (uses_domain[d]) { inner = d; break; }
    idx_t alloc[MAX_DOMAIN_DIMS] = {1, 1, 1};
    for (size_t p = 0; p < dims.size(); p++) {
        if (dims[p].is_outer) {          // outer domain dim: allocated range = domain + pads, like YkVarBase::resize
            const idx_t pl = std::max<idx_t>(dims[p].outer_halo_l, soln->min_pad[3]) + soln->extra_pad[3];
            const idx_t pr = std::max<idx_t>(dims[p].outer_halo_r, soln->min_pad[3]) + soln->extra_pad[3];
            dims[p].first_misc = -pl;
            dims[p].last_misc = soln->local_size[3] - 1 + pr;
        }
        if (dims[p].type != DIM_DOMAIN) continue;
        int d = dims[p].domain_idx;
        if (fixed_size) {
            dom_size[d] = fixed_sizes[p];
            rank_ofs[d] = 0;
            pad_l[d] = pad_r[d] = 0;
        } else {
            dom_size[d] = soln->local_size[d];
            rank_ofs[d] = soln->rank_ofs[d];
            // pad = max(halo, min pads) + extra pad (YkVarBase::resize, src/kernel/lib/yk_var.cpp:239-326);
            // additionally every var shares the solution-wide maximum so that all full-dim vars
            // have identical strides (one address computation serves all of them in a kernel).
            // (+ the wave-front extension of a multi-rank -Mbt run, the reference's left/right_wf_exts, yk_var.cpp:239-326)
            pad_l[d] = std::max<idx_t>({halo_l[d], min_pad_l[d], soln->min_pad[d], soln->shared_pad_l(d)}) + soln->extra_pad[d] + soln->wf_ext(d);
            pad_r[d] = std::max<idx_t>({halo_r[d], min_pad_r[d], soln->min_pad[d], soln->shared_pad_r(d)}) + soln->extra_pad[d] + soln->wf_ext(d);
        }
        if (d == inner) {
            pad_l[d] = round_up(pad_l[d], zal);
            idx_t vz = 16 / eb;
            idx_t pr = round_up(pad_r[d], vz);
            idx_t pitch = round_up(pad_l[d] + dom_size[d] + pr, zal);
            pad_r[d] = pitch - pad_l[d] - dom_size[d];
        }
        alloc[d] = pad_l[d] + dom_size[d] + pad_r[d];
    }
    // strides, innermost domain dim first
    idx_t st = 1;
    for (int d = soln->ndd - 1; d >= 0; d--) {
        if (uses_domain[d]) { stride[d] = st; st *= alloc[d]; }
        else { stride[d] = 0; dom_size[d] = 1; pad_l[d] = pad_r[d] = 0; rank_ofs[d] = 0; }
    }
    // misc dims outside the domain dims (innermost misc dim = last declared)
    misc_stride.assign(dims.size(), 0);
    misc_elems = 1;
    for (int p = (int)dims.size() - 1; p >= 0; p--) {
        if (dims[p].type != DIM_MISC) continue;
        misc_stride[p] = st;
        idx_t n = dims[p].last_misc - dims[p].first_misc + 1;
        st *= n;
        misc_elems *= n;
    }
    slot_elems = round_up(st, zal);
    origin_elems = 0;
    for (int d = 0; d < soln->ndd; d++) origin_elems += pad_l[d] * stride[d];
}

std::shared_ptr<void> Var::own_allocation(void* alloc) {
    return std::shared_ptr<void>(alloc, [](void* p) { if (p) (void)hipFree(p); });
}
// The storage of this var -- and of every var fused with it -- becomes [data, data + nbytes) inside `alloc`.
void Var::adopt_storage(std::shared_ptr<void> owner, void* alloc, void* data, size_t nbytes) {
    auto set = [&](Var& m) { m.alloc_owner = owner; m.alloc_ptr = alloc; m.dptr = data; m.alloc_bytes = nbytes; };
    if (fuse_group) {
        // Vars fused while unprepared may have ended up with different geometry (pads / sizes of their own solutions): every
        // member indexes this ONE allocation with its own strides, so all of them must describe the same layout -- the
        // reference demands identical layouts when it fuses into a solution var (yk_var_apis.cpp:344-351).  ADVICE r03.
        for (Var* m : *fuse_group) {
            if (m == this || !m->soln->prepared) continue;      // (a member whose solution is not prepared yet has no geometry of its own so far)
            bool same = m->nslots == nslots && m->slot_elems == slot_elems && m->origin_elems == origin_elems && m->bytes() == bytes();
            for (int d = 0; d < MAX_DOMAIN_DIMS && same; d++) same = m->stride[d] == stride[d] && m->pad_l[d] == pad_l[d] && m->dom_size[d] == dom_size[d];
            if (!same)
                YKH_THROW("fuse_vars: storage layouts of the fused vars '" + name + "' and '" + m->name + "' differ (their solutions gave them different "
                          "sizes or pads after they were fused): they cannot share one allocation");
        }
        for (Var* m : *fuse_group) set(*m);
    } else set(*this);
}
void Var::drop_storage_refs() {
    alloc_owner.reset();
    alloc_ptr = nullptr;
    dptr = nullptr;
    alloc_bytes = 0;
    if (scratch) { (void)hipFree(scratch); scratch = nullptr; }
    mirror_.clear();
    raw_exposed_ = false;
}
// fuse_vars(): from here on this var and `src` are two names for one var (src/kernel/lib/yk_var_apis.cpp:334-367 points
// both API objects at one YkVarBase): one allocation, one valid-step window, one set of dirty flags.
void Var::fuse_with(Var& src) {
    if (&src == this || (fuse_group && fuse_group == src.fuse_group)) return;
    // leave my old group, release what I held
    if (fuse_group) {
        auto& g = *fuse_group;
        g.erase(std::remove(g.begin(), g.end(), this), g.end());
        fuse_group.reset();
    }
    drop_storage_refs();
    if (!src.fuse_group) { src.fuse_group = std::make_shared<std::vector<Var*>>(); src.fuse_group->push_back(&src); }
    fuse_group = src.fuse_group;
    fuse_group->push_back(this);
    alloc_owner = src.alloc_owner; alloc_ptr = src.alloc_ptr; dptr = src.dptr; alloc_bytes = src.alloc_bytes;
    first_valid_step = src.first_valid_step;
    dirty = src.dirty;
    dirty.resize((size_t)std::max(1, nslots), 1);       // (flags are indexed by MY slots; a longer or shorter source must not be read past its end)
    soln->note_storage_changed();
}

void Var::allocate() {
    release();
    size_t nb = bytes();
    if (nb == 0) nb = 256;
    void* ap = nullptr;
    YKH_HIP(hipMalloc(&ap, nb));
    adopt_storage(own_allocation(ap), ap, ap, nb);
    // Zero on the solution's own stream: hipMemset() runs on the NULL stream, which the solution's
    // non-blocking streams do not synchronise with -- a multi-GB memset was still clearing the tail of the
    // allocation while the first init kernel had already written it (seen at >= 512^3).
    YKH_HIP(hipMemsetAsync(dptr, 0, nb, soln->compute_stream));
    YKH_HIP(hipStreamSynchronize(soln->compute_stream));
    soln->note_storage_changed();
}

// release_storage(): "after fusing, calling release_storage() on this var or the source var will apply to both"
void Var::release() {
    const bool had = dptr != nullptr;
    if (fuse_group) for (Var* m : *fuse_group) m->drop_storage_refs();
    else drop_storage_refs();
    if (had) soln->note_storage_changed();
}

int Var::slot_of(idx_t t) const { return has_step ? (int)imod_flr(t, nslots) : 0; }

void* Var::slot_base(idx_t t) const {
    return (char*)dptr + ((size_t)slot_of(t) * slot_elems + origin_elems) * elem_bytes();
}

// Valid-step window (YkVarBase::update_valid_step, src/kernel/lib/yk_var.cpp:559-575).
void Var::update_valid_step(idx_t t) {
    if (!has_step) return;
    if (t < first_valid_step) first_valid_step = t;
    else if (t > last_valid_step()) first_valid_step = t - nslots + 1;
    if (fuse_group) for (Var* m : *fuse_group) m->first_valid_step = first_valid_step;
}

int Var::dim_posn(const std::string& dim, bool must_exist) const {
    for (size_t i = 0; i < dims.size(); i++)
        if (dims[i].name == dim) return (int)i;
    if (must_exist) YKH_THROW("dimension '" + dim + "' not found in var '" + name + "'");
    return -1;
}

idx_t Var::first_local_index(int p) const {
    const VarDim& d = dims[p];
    if (d.type == DIM_STEP) return first_valid_step;
    if (d.type == DIM_MISC) return d.first_misc;
    return rank_ofs[d.domain_idx] - pad_l[d.domain_idx];
}
idx_t Var::last_local_index(int p) const {
    const VarDim& d = dims[p];
    if (d.type == DIM_STEP) return last_valid_step();
    if (d.type == DIM_MISC) return d.last_misc;
    return rank_ofs[d.domain_idx] + dom_size[d.domain_idx] + pad_r[d.domain_idx] - 1;
}
idx_t Var::alloc_size(int p) const {
    const VarDim& d = dims[p];
    if (d.type == DIM_STEP) return nslots;
    return last_local_index(p) - first_local_index(p) + 1;
}

bool Var::indices_local(const std::vector<idx_t>& idx) const {
    if (idx.size() != dims.size()) return false;
    for (size_t p = 0; p < dims.size(); p++)
        if (idx[p] < first_local_index((int)p) || idx[p] > last_local_index((int)p)) return false;
    return true;
}

std::string Var::format_indices(const std::vector<idx_t>& idx) const {
    std::ostringstream os;
    for (size_t p = 0; p < dims.size() && p < idx.size(); p++) os << (p ? ", " : "") << dims[p].name << "=" << idx[p];
    return os.str();
}

void Var::check_indices(const std::vector<idx_t>& idx, const char* fn, bool strict, bool check_step,
                        bool* clipped) const {
    if (idx.size() != dims.size())
        YKH_THROW(std::string(fn) + " called with " + std::to_string(idx.size()) + " indices instead of " +
                  std::to_string(dims.size()) + " for var '" + name + "'");
    if (clipped) *clipped = false;
    for (size_t p = 0; p < dims.size(); p++) {
        if (dims[p].type == DIM_STEP && (!check_step || soln->step_wrap)) continue;
        idx_t lo = first_local_index((int)p), hi = last_local_index((int)p);
        if (idx[p] < lo || idx[p] > hi) {
            if (strict)
                YKH_THROW(std::string(fn) + ": index in dim '" + dims[p].name + "' is " + std::to_string(idx[p]) +
                          ", which is not in allowed range [" + std::to_string(lo) + "..." + std::to_string(hi) +
                          "] of var '" + name + "'");
            if (clipped) *clipped = true;
        }
    }
}

// Walk the (step, misc) combinations of a slice; for each, hand `fn` a BoxCopyArgs describing the
// domain-dim box on the var side and the matching strided view of the caller's row-major buffer.
template <class F>
idx_t Var::for_boxes(const std::vector<idx_t>& first, const std::vector<idx_t>& last, bool strict,
                     bool update_step, F&& fn) const {
    if (!dptr) YKH_THROW("call to access var '" + name + "' with no storage allocated");
    const size_t nd = dims.size();
    if (first.size() != nd || last.size() != nd)
        YKH_THROW("slice indices for var '" + name + "' must have " + std::to_string(nd) + " values");
    // clip (non-strict) or validate (strict)
    std::vector<idx_t> lo(first), hi(last);
    for (size_t p = 0; p < nd; p++) {
        if (dims[p].type == DIM_STEP && (update_step || soln->step_wrap)) continue;   // window follows the writes / wraps
        idx_t alo = first_local_index((int)p), ahi = last_local_index((int)p);
        if (strict) {
            if (lo[p] < alo || hi[p] > ahi)
                YKH_THROW("slice range [" + std::to_string(lo[p]) + "..." + std::to_string(hi[p]) + "] in dim '" +
                          dims[p].name + "' is not within allowed range [" + std::to_string(alo) + "..." +
                          std::to_string(ahi) + "] of var '" + name + "'");
        } else {
            lo[p] = std::max(lo[p], alo);
            hi[p] = std::min(hi[p], ahi);
        }
    }
    for (size_t p = 0; p < nd; p++)
        if (hi[p] < lo[p]) return 0;
    // buffer strides follow the *requested* slice extents (first..last), row-major in var dim order
    std::vector<idx_t> bstride(nd, 1);
    for (int p = (int)nd - 2; p >= 0; p--) bstride[p] = bstride[p + 1] * (last[p + 1] - first[p + 1] + 1);
    BoxCopyArgs a{};
    a.var_elem_bytes = elem_bytes();
    a.buf_elem_bytes = elem_bytes();
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++) { a.lo[d] = 0; a.n[d] = 1; a.bs[d] = 0; }
    a.sx = stride[0]; a.sy = stride[1]; a.sz = stride[2];
    idx_t nbox = 1;
    std::vector<int> outer;   // positions of step/misc dims
    for (size_t p = 0; p < nd; p++) {
        if (dims[p].type == DIM_DOMAIN) {
            int d = dims[p].domain_idx;
            a.lo[d] = lo[p] - rank_ofs[d];
            a.n[d] = hi[p] - lo[p] + 1;
            a.bs[d] = bstride[p];
            nbox *= a.n[d];
        } else outer.push_back((int)p);
    }
    idx_t total = 0;
    std::vector<idx_t> cur(nd);
    for (size_t p = 0; p < nd; p++) cur[p] = lo[p];
    while (true) {
        idx_t t = has_step ? cur[step_posn] : 0;
        if (has_step && update_step) const_cast<Var*>(this)->update_valid_step(t);
        idx_t var_ofs = (idx_t)slot_of(t) * slot_elems + origin_elems;
        idx_t buf_ofs = 0;
        for (size_t p = 0; p < nd; p++) {
            if (dims[p].type == DIM_MISC) var_ofs += (cur[p] - dims[p].first_misc) * misc_stride[p];
            if (dims[p].type != DIM_DOMAIN) buf_ofs += (cur[p] - first[p]) * bstride[p];
            else buf_ofs += (lo[p] - first[p]) * bstride[p];
        }
        a.var_base = (char*)dptr + (size_t)var_ofs * elem_bytes();
        fn(a, buf_ofs);
        total += nbox;
        // next (step, misc) combination
        int k = (int)outer.size() - 1;
        for (; k >= 0; k--) {
            int p = outer[k];
            if (++cur[p] <= hi[p]) break;
            cur[p] = lo[p];
        }
        if (k < 0) break;
    }
    return total;
}

static void* staging(size_t bytes, hipStream_t s) {
    // one grow-only device staging buffer per process (API calls are not thread-safe by contract)
    static void* buf = nullptr;
    static size_t cap = 0;
    if (bytes > cap) {
        if (buf) { (void)hipStreamSynchronize(s); (void)hipFree(buf); }
        cap = std::max(bytes, (size_t)1 << 20);
        YKH_HIP(hipMalloc(&buf, cap));
    }
    return buf;
}

idx_t Var::get_elements_in_slice(void* buf, size_t buf_elems, int buf_eb, const std::vector<idx_t>& first,
                                 const std::vector<idx_t>& last) const {
    before_device_use();
    if (!dptr) YKH_THROW("call to 'get_elements_in_slice' with no storage allocated for var '" + name + "'");
    size_t need = 1;
    for (size_t p = 0; p < dims.size() && p < first.size() && p < last.size(); p++) {
        if (last[p] < first[p]) return 0;
        need *= (size_t)(last[p] - first[p] + 1);
    }
    if (buf_elems < need)
        YKH_THROW("call to 'get_elements_in_slice' with buffer of size " + std::to_string(buf_elems) +
                  "; " + std::to_string(need) + " needed");
    hipStream_t s = soln->compute_stream;
    void* stg = staging(need * buf_eb, s);
    idx_t n = for_boxes(first, last, true, false, [&](BoxCopyArgs& a, idx_t bofs) {
        a.buf = (char*)stg + (size_t)bofs * buf_eb;
        a.buf_elem_bytes = buf_eb;
        launch_box_gather(a, s);
    });
    YKH_HIP(hipMemcpyAsync(buf, stg, need * buf_eb, hipMemcpyDeviceToHost, s));
    YKH_HIP(hipStreamSynchronize(s));
    return n;
}

idx_t Var::set_elements_in_slice(const void* buf, size_t buf_elems, int buf_eb, const std::vector<idx_t>& first,
                                 const std::vector<idx_t>& last) {
    before_device_use();
    if (!dptr) YKH_THROW("call to 'set_elements_in_slice' with no storage allocated for var '" + name + "'");
    size_t need = 1;
    for (size_t p = 0; p < dims.size() && p < first.size() && p < last.size(); p++) {
        if (last[p] < first[p]) return 0;
        need *= (size_t)(last[p] - first[p] + 1);
    }
    if (buf_elems < need)
        YKH_THROW("call to 'set_elements_in_slice' with buffer of size " + std::to_string(buf_elems) +
                  "; " + std::to_string(need) + " needed");
    hipStream_t s = soln->compute_stream;
    void* stg = staging(need * buf_eb, s);
    YKH_HIP(hipMemcpyAsync(stg, buf, need * buf_eb, hipMemcpyHostToDevice, s));
    idx_t n = for_boxes(first, last, true, true, [&](BoxCopyArgs& a, idx_t bofs) {
        a.buf = (char*)stg + (size_t)bofs * buf_eb;
        a.buf_elem_bytes = buf_eb;
        launch_box_scatter(a, s);
    });
    YKH_HIP(hipStreamSynchronize(s));
    set_dirty_all(true);
    after_device_write();
    return n;
}

idx_t Var::set_elements_in_slice_same(double v, const std::vector<idx_t>& first, const std::vector<idx_t>& last,
                                      bool strict) {
    before_device_use();
    hipStream_t s = soln->compute_stream;
    idx_t n = for_boxes(first, last, strict, true, [&](BoxCopyArgs& a, idx_t) { launch_box_fill(a, v, s); });
    YKH_HIP(hipStreamSynchronize(s));
    set_dirty_all(true);
    after_device_write();
    return n;
}

// Sets every allocated element including pads (GenericVar::set_elems_same, generic_var.cpp:137-165).
void Var::set_all_elements_same(double v) {
    if (!dptr) YKH_THROW("call to 'set_all_elements_same' with no storage allocated for var '" + name + "'");
    hipStream_t s = soln->compute_stream;
    BoxCopyArgs a{};
    a.var_base = dptr;
    a.var_elem_bytes = a.buf_elem_bytes = elem_bytes();
    a.sx = a.sy = 0; a.sz = 1;
    a.lo[0] = a.lo[1] = a.lo[2] = 0;
    a.n[0] = 1; a.n[1] = 1; a.n[2] = slot_elems * nslots;
    launch_box_fill(a, v, s);
    YKH_HIP(hipStreamSynchronize(s));
    set_dirty_all(true);
    after_device_write();
}

void Var::set_elements_hash(double offset, double scale, int hash_id) {
    if (!dptr) YKH_THROW("call to 'set_elements_hash' with no storage allocated for var '" + name + "'");
    hipStream_t s = soln->compute_stream;
    const size_t nd = dims.size();
    std::vector<idx_t> first(nd), last(nd);
    for (size_t p = 0; p < nd; p++) {
        if (dims[p].type == DIM_DOMAIN) {
            int d = dims[p].domain_idx;
            first[p] = rank_ofs[d] - halo_l[d];
            last[p] = rank_ofs[d] + dom_size[d] + halo_r[d] - 1;
        } else if (dims[p].is_outer) {
            first[p] = -dims[p].outer_halo_l;
            last[p] = soln->local_size[3] - 1 + dims[p].outer_halo_r;
        } else {
            first[p] = first_local_index((int)p);
            last[p] = last_local_index((int)p);
        }
    }
    // walk (step, misc) by hand to know slot/misc for the hash
    std::vector<idx_t> cur(first);
    while (true) {
        idx_t slot = has_step ? slot_of(cur[step_posn]) : 0;
        idx_t misc = 0;
        idx_t var_ofs = slot * slot_elems + origin_elems;
        for (size_t p = 0; p < nd; p++)
            if (dims[p].type == DIM_MISC) {
                misc = misc * 131 + cur[p];
                var_ofs += (cur[p] - dims[p].first_misc) * misc_stride[p];
            }
        BoxCopyArgs a{};
        a.var_elem_bytes = a.buf_elem_bytes = elem_bytes();
        a.var_base = (char*)dptr + (size_t)var_ofs * elem_bytes();
        a.sx = stride[0]; a.sy = stride[1]; a.sz = stride[2];
        idx_t g[3] = {0, 0, 0};
        for (int d = 0; d < MAX_DOMAIN_DIMS; d++) {
            a.lo[d] = uses_domain[d] ? -halo_l[d] : 0;
            a.n[d] = uses_domain[d] ? dom_size[d] + halo_l[d] + halo_r[d] : 1;
            g[d] = uses_domain[d] ? rank_ofs[d] - halo_l[d] : 0;
        }
        launch_box_hash(a, offset, scale, hash_id, slot + 1024 * misc, g[0], g[1], g[2], s);
        int p = (int)nd - 1;
        for (; p >= 0; p--) {
            if (dims[p].type == DIM_DOMAIN) continue;
            if (++cur[p] <= last[p]) break;
            cur[p] = first[p];
        }
        if (p < 0) break;
    }
    YKH_HIP(hipStreamSynchronize(s));
    set_dirty_all(true);
    after_device_write();
}

double Var::get_element(const std::vector<idx_t>& idx) const {
    if (!dptr) YKH_THROW("call to 'get_element' with no storage allocated for var '" + name + "'");
    check_indices(idx, "get_element", true, true);
    double out = 0;
    if (elem_bytes() == 4) { float f; get_elements_in_slice(&f, 1, 4, idx, idx); out = f; }
    else get_elements_in_slice(&out, 1, 8, idx, idx);
    return out;
}

idx_t Var::set_element(double v, const std::vector<idx_t>& idx, bool strict) {
    if (!dptr) YKH_THROW("call to 'set_element' with no storage allocated for var '" + name + "'");
    bool clipped = false;
    check_indices(idx, "set_element", strict, false, &clipped);
    if (clipped) return 0;
    return set_elements_in_slice_same(v, idx, idx, true);
}

idx_t Var::add_to_element(double v, const std::vector<idx_t>& idx, bool strict) {
    before_device_use();
    if (!dptr) YKH_THROW("call to 'add_to_element' with no storage allocated for var '" + name + "'");
    bool clipped = false;
    check_indices(idx, "add_to_element", strict, true, &clipped);
    if (clipped) return 0;
    hipStream_t s = soln->compute_stream;
    idx_t n = for_boxes(idx, idx, true, false, [&](BoxCopyArgs& a, idx_t) { launch_box_add(a, v, s); });
    YKH_HIP(hipStreamSynchronize(s));
    set_dirty_all(true);
    after_device_write();
    return n;
}

Var::Reduction Var::reduce_elements_in_slice(int mask, const std::vector<idx_t>& first,
                                             const std::vector<idx_t>& last, bool strict) const {
    before_device_use();
    hipStream_t s = soln->compute_stream;
    static double* dout = nullptr;
    if (!dout) YKH_HIP(hipMalloc(&dout, 5 * sizeof(double)));
    Reduction r;
    r.mask = mask;
    r.vmax = -INFINITY;
    r.vmin = INFINITY;
    idx_t n = for_boxes(first, last, strict, false, [&](BoxCopyArgs& a, idx_t) {
        launch_box_reduce(a, dout, s);
        double h[5];
        YKH_HIP(hipMemcpyAsync(h, dout, sizeof(h), hipMemcpyDeviceToHost, s));
        YKH_HIP(hipStreamSynchronize(s));
        r.sum += h[0]; r.sum_sq += h[1]; r.prod *= h[2];
        r.vmax = std::max(r.vmax, h[3]); r.vmin = std::min(r.vmin, h[4]);
    });
    r.n = n;
    return r;
}

idx_t Var::compare(const Var& ref, double eps) const {
    before_device_use();
    ref.before_device_use();
    if (dims.size() != ref.dims.size()) return 1;
    if (!dptr || !ref.dptr) return (dptr || ref.dptr) ? 1 : 0;
    hipStream_t s = soln->compute_stream;
    static unsigned long long* dcnt = nullptr;
    if (!dcnt) YKH_HIP(hipMalloc(&dcnt, sizeof(unsigned long long)));
    YKH_HIP(hipMemsetAsync(dcnt, 0, sizeof(unsigned long long), s));
    // in-domain points of every valid step & misc index (yk_var.cpp:440-449 compares domain only)
    const size_t nd = dims.size();
    std::vector<idx_t> first(nd), last(nd);
    for (size_t p = 0; p < nd; p++) {
        if (dims[p].type == DIM_DOMAIN) {
            int d = dims[p].domain_idx;
            first[p] = rank_ofs[d];
            last[p] = rank_ofs[d] + dom_size[d] - 1;
        } else {
            first[p] = first_local_index((int)p);
            last[p] = last_local_index((int)p);
        }
    }
    std::vector<BoxCopyArgs> mine, theirs;
    for_boxes(first, last, true, false, [&](BoxCopyArgs& a, idx_t) { mine.push_back(a); });
    ref.for_boxes(first, last, true, false, [&](BoxCopyArgs& a, idx_t) { theirs.push_back(a); });
    if (mine.size() != theirs.size()) return 1;
    for (size_t i = 0; i < mine.size(); i++) launch_box_compare(mine[i], theirs[i], eps, dcnt, s);
    unsigned long long h = 0;
    YKH_HIP(hipMemcpyAsync(&h, dcnt, sizeof(h), hipMemcpyDeviceToHost, s));
    YKH_HIP(hipStreamSynchronize(s));
    return (idx_t)h;
}

// (a fuse group is one var: data written through one name needs its halos exchanged under every name)
void Var::set_dirty(bool d, idx_t t) {
    const int sl = slot_of(t);
    if (fuse_group) { for (Var* m : *fuse_group) if (sl < (int)m->dirty.size()) m->dirty[sl] = d ? 1 : 0; }
    else dirty[sl] = d ? 1 : 0;
}
void Var::set_dirty_all(bool d) {
    if (fuse_group) { for (Var* m : *fuse_group) std::fill(m->dirty.begin(), m->dirty.end(), d ? 1 : 0); }
    else std::fill(dirty.begin(), dirty.end(), d ? 1 : 0);
}
bool Var::is_dirty(idx_t t) const { return dirty[slot_of(t)] != 0; }

// Host mirror for yk_var::get_raw_storage_buffer() (src/kernel/lib/yk_var.hpp:2624).  In the reference the raw pointer IS the
// storage: a caller may read or write through it at any time without telling the library.  Here the storage lives in HBM,
// so from the first get_raw_storage_buffer() call on a var until its storage is released (or release_raw_storage() is
// called) the library keeps a host copy in the same layout coherent with the device array around every API call that
// touches the var: host -> device before the call uses the data (the caller may have written through the pointer),
// device -> host after a call that changed it (run_solution, set_*, halo exchange).  That is two whole-var copies per call
// for exposed vars -- the price of the reference's contract for the rare caller that wants raw access, none for the others.
void* Var::host_mirror() {
    if (!dptr) return nullptr;
    if (raw_exposed_ && mirror_.size() == bytes()) { before_device_use(); return mirror_.data(); }   // (keeps the caller's edits)
    mirror_.assign(bytes(), 0);
    raw_exposed_ = true;
    after_device_write();
    return mirror_.data();
}
// (fused vars: a raw buffer handed out under one name is pushed / pulled whichever name the call goes through)
void Var::before_device_use() const {
    auto push = [](const Var& m) {
        if (!m.raw_exposed_ || !m.dptr || m.mirror_.size() != m.bytes()) return;
        YKH_HIP(hipStreamSynchronize(m.soln->compute_stream));
        YKH_HIP(hipMemcpyAsync(m.dptr, m.mirror_.data(), m.bytes(), hipMemcpyHostToDevice, m.soln->compute_stream));
        YKH_HIP(hipStreamSynchronize(m.soln->compute_stream));
    };
    if (fuse_group) for (const Var* m : *fuse_group) push(*m);
    else push(*this);
}
void Var::after_device_write() {
    auto pull = [](Var& m) {
        if (!m.raw_exposed_ || !m.dptr || m.mirror_.size() != m.bytes()) return;
        YKH_HIP(hipStreamSynchronize(m.soln->compute_stream));
        YKH_HIP(hipMemcpyAsync(m.mirror_.data(), m.dptr, m.bytes(), hipMemcpyDeviceToHost, m.soln->compute_stream));
        YKH_HIP(hipStreamSynchronize(m.soln->compute_stream));
    };
    if (fuse_group) for (Var* m : *fuse_group) pull(*m);
    else pull(*this);
}
// Kept for callers of round 1's extension: pushes the host copy now (before_device_use() does it anyway).
void Var::sync_mirror_to_device() {
    before_device_use();
    set_dirty_all(true);
}
void Var::release_raw_storage() {
    before_device_use();      // last edits made through the pointer
    raw_exposed_ = false;
    mirror_.clear();
    mirror_.shrink_to_fit();
}

}  // namespace ykh
