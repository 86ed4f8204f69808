// ykh_c_api.cpp -- extern "C" boundary of libyask_kernel.<stencil>.cdna4_hip.so
// (declarations and reference citations: include/yask_hip_c_api.h). Exceptions never cross the
// boundary: they are caught here and turned into an error code + yk_last_error().
#include "../../include/yask_hip_c_api.h"

#include <cstring>
#include <sstream>

#include "ykh_handles.hpp"
#include "ykh_runtime.hpp"

using namespace ykh;

static thread_local std::string g_err;
static thread_local int g_err_code = 0;

#define YK_TRY g_err_code = 0; try {
#define YK_CATCH(retval)                                                           \
    } catch (const std::exception& e) { g_err = e.what(); g_err_code = 1; return retval; } \
      catch (...) { g_err = "YASK error: unknown exception"; g_err_code = 1; return retval; }

static Var* V(yk_var_h v) {
    if (!v) YKH_THROW("null var handle");
    return reinterpret_cast<Var*>(v);
}
static Solution& S(yk_soln_h s) {
    if (!s || !s->soln) YKH_THROW("null solution handle");
    return *s->soln;
}
static std::vector<idx_t> vec(Var* v, const yk_idx_t* p) {
    if (v->dims.empty()) return {};                      // a scalar var: no indices (std::vector<idx_t>{}.data() may be null)
    if (!p) YKH_THROW("null index array");
    return std::vector<idx_t>(p, p + v->dims.size());
}
static int dom(Var* v, const char* dim, const char* fn) {
    int p = v->dim_posn(dim);
    if (v->dims[p].type != DIM_DOMAIN)
        YKH_THROW(std::string(fn) + ": dimension '" + dim + "' of var '" + v->name + "' is not a domain dimension");
    return v->dims[p].domain_idx;
}

extern "C" {

const char* yk_last_error(void) { return g_err.c_str(); }
int yk_last_error_code(void) { return g_err_code; }
void yk_clear_error(void) { g_err.clear(); g_err_code = 0; }

const char* yk_get_version_string(void) { static std::string v = version_string(); return v.c_str(); }

yk_env_h yk_new_env(void) {
    YK_TRY
    auto* e = new yk_env_s;
    e->env = std::make_shared<Env>();
    return e;
    YK_CATCH(nullptr)
}
void yk_free_env(yk_env_h env) { delete env; }

yk_soln_h yk_new_solution(yk_env_h env) {
    YK_TRY
    if (!env) YKH_THROW("null env handle");
    auto* s = new yk_solution_s;
    s->soln = std::make_shared<Solution>(env->env, ykh_solution_impl());
    return s;
    YK_CATCH(nullptr)
}
yk_soln_h yk_new_solution_from(yk_env_h env, yk_soln_h source) {
    YK_TRY
    yk_soln_h s = yk_new_solution(env);
    if (!s) return nullptr;
    Solution& d = *s->soln;
    Solution& o = S(source);
    for (int i = 0; i < MAX_API_DOMAIN_DIMS; i++) {
        d.global_size[i] = o.global_size[i]; d.rank_size[i] = o.rank_size[i];
        d.num_ranks[i] = o.num_ranks[i]; d.rank_index[i] = o.rank_index[i];
        d.min_pad[i] = o.min_pad[i]; d.extra_pad[i] = o.extra_pad[i];
    }
    for (int i = 0; i <= MAX_API_DOMAIN_DIMS; i++) { d.block_size[i] = o.block_size[i]; d.mega_block_size[i] = o.mega_block_size[i]; }
    d.rank_index_set = o.rank_index_set;
    d.overlap_comms = o.overlap_comms; d.min_exterior = o.min_exterior; d.do_halo_exchange = o.do_halo_exchange;
    d.auto_tune = o.auto_tune; d.force_scalar = o.force_scalar; d.variant_override = o.variant_override;
    d.xchunk_override = o.xchunk_override;
    // every cdna4_hip-specific setting as well: a validation copy must run the same configuration
    d.direct_halo = o.direct_halo; 
    d.thin_slab_point_kernel = o.thin_slab_point_kernel; d.tune_at_prepare = o.tune_at_prepare;
    d.auto_tune_trial_secs = o.auto_tune_trial_secs; d.step_wrap = o.step_wrap; d.step_timers = o.step_timers;
    d.ignored_opts = o.ignored_opts; d.fuse_steps = o.fuse_steps; d.step_graphs = o.step_graphs; d.fast_div = o.fast_div; d.placement_trials = o.placement_trials;
    return s;
    YK_CATCH(nullptr)
}
void yk_free_solution(yk_soln_h s) { delete s; }

int yk_env_get_num_ranks(yk_env_h e) { return e ? e->env->nranks : 0; }
int yk_env_get_rank_index(yk_env_h e) { return e ? e->env->rank : 0; }
int yk_env_global_barrier(yk_env_h e) {
    YK_TRY
    if (!e) YKH_THROW("null env handle");
    (void)e->env->sum_over_ranks(0);
    return 0;
    YK_CATCH(1)
}
yk_idx_t yk_env_sum_over_ranks(yk_env_h e, yk_idx_t v) {
    YK_TRY
    if (!e) YKH_THROW("null env handle");
    return e->env->sum_over_ranks(v);
    YK_CATCH(0)
}
void yk_env_set_trace_enabled(yk_env_h e, int en) { if (e) e->env->trace = en != 0; }
int yk_env_get_device_bus_id(yk_env_h e, char* out, int cap) {
    YK_TRY
    if (!e || !out || cap < 2) YKH_THROW("get_device_bus_id: bad arguments");
    int dev = 0;
    YKH_HIP(hipGetDevice(&dev));
    YKH_HIP(hipDeviceGetPCIBusId(out, cap, dev));
    out[cap - 1] = 0;
    return (int)std::strlen(out);
    YK_CATCH(-1)
}
int yk_env_set_ranks(yk_env_h e, int rank, int n) {
    YK_TRY
    if (!e) YKH_THROW("null env handle");
    e->env->set_ranks(rank, n);
    return 0;
    YK_CATCH(1)
}
// control-plane counters of the installed halo transport (ykh_ipc.cpp): out[0] registrations sent over the TCP mesh, out[1]
// their bytes, out[2] collective begin calls (one per run_solution() / exchange_halos()), out[3] collective resets, out[4]
// device operations enqueued (flag kernels + copies), out[5] kind of mailbox memory; returns how many it knows, 0 = none
int yk_env_get_transport_counters(yk_env_h e, long long* out, int cap) {
    YK_TRY
    if (!e) YKH_THROW("null env handle");
    if (!e->env->exch_counters || !out || cap <= 0) return 0;
    return e->env->exch_counters(e->env->user, out, cap);
    YK_CATCH(-1)
}
int yk_env_set_transport(yk_env_h e, yk_exchange_fn start, yk_exchange_fn wait, yk_allreduce_fn ar, void* user) {
    YK_TRY
    if (!e) YKH_THROW("null env handle");
    static_assert(sizeof(yk_halo_msg) == sizeof(HaloMsg), "yk_halo_msg must mirror ykh::HaloMsg");
    e->env->drop_transport();               // a built-in transport installed earlier goes, with every hook it had set
    e->env->exch_start = reinterpret_cast<ykh_exchange_fn>(start);
    e->env->exch_wait = reinterpret_cast<ykh_exchange_fn>(wait);
    e->env->allreduce = ar;
    e->env->user = user;
    e->env->user_free = nullptr;            // the host owns its own state
    return 0;
    YK_CATCH(1)
}

// ---- solution
const char* yk_solution_get_name(yk_soln_h s) { return s ? s->soln->meta->name : ""; }
const char* yk_solution_get_description(yk_soln_h s) { return s ? s->soln->meta->description : ""; }
const char* yk_solution_get_target(yk_soln_h s) { return s ? s->soln->meta->target : ""; }
int yk_solution_is_offloaded(yk_soln_h) { return 1; }
int yk_solution_get_element_bytes(yk_soln_h s) { return s ? s->soln->meta->elem_bytes : 0; }
const char* yk_solution_get_step_dim_name(yk_soln_h s) { return s ? s->soln->step_dim_name.c_str() : ""; }
int yk_solution_get_num_domain_dims(yk_soln_h s) { return s ? s->soln->ndd + (s->soln->has_outer ? 1 : 0) : 0; }
const char* yk_solution_get_domain_dim_name(yk_soln_h s, int i) {
    YK_TRY
    Solution& so = S(s);
    if (so.has_outer) { if (i == 0) return so.outer_dim_name.c_str(); i--; }      // outermost first
    if (i < 0 || i >= so.ndd) YKH_THROW("domain-dim index out of range");
    return so.domain_dim_names[i].c_str();
    YK_CATCH("")
}
int yk_solution_get_num_misc_dims(yk_soln_h s) { return s ? (int)s->soln->misc_dim_names.size() : 0; }
const char* yk_solution_get_misc_dim_name(yk_soln_h s, int i) {
    YK_TRY
    Solution& so = S(s);
    if (i < 0 || i >= (int)so.misc_dim_names.size()) YKH_THROW("misc-dim index out of range");
    return so.misc_dim_names[i].c_str();
    YK_CATCH("")
}

#define SOLN_SET(NAME, BODY)                                                   \
    int NAME(yk_soln_h s, const char* dim, yk_idx_t n) {                       \
        YK_TRY                                                                 \
        Solution& so = S(s);                                                   \
        BODY;                                                                  \
        return 0;                                                              \
        YK_CATCH(1)                                                            \
    }
#define SOLN_GET(NAME, EXPR)                                                   \
    yk_idx_t NAME(yk_soln_h s, const char* dim) {                              \
        YK_TRY                                                                 \
        Solution& so = S(s);                                                   \
        return (EXPR);                                                         \
        YK_CATCH(0)                                                            \
    }
// Setting a size after prepare_solution() clears "prepared" (soln_apis.cpp:47-57,86-103).
SOLN_SET(yk_solution_set_rank_domain_size, { int d = so.domain_dim_idx(dim, "set_rank_domain_size"); so.rank_size[d] = n; if (n) so.global_size[d] = 0; so.invalidate(); })
SOLN_GET(yk_solution_get_rank_domain_size, so.prepared ? so.local_size[so.domain_dim_idx(dim, "get_rank_domain_size")] : so.rank_size[so.domain_dim_idx(dim, "get_rank_domain_size")])
SOLN_SET(yk_solution_set_overall_domain_size, { int d = so.domain_dim_idx(dim, "set_overall_domain_size"); so.global_size[d] = n; if (n) so.rank_size[d] = 0; so.invalidate(); })
SOLN_GET(yk_solution_get_overall_domain_size, so.global_size[so.domain_dim_idx(dim, "get_overall_domain_size")])
SOLN_SET(yk_solution_set_num_ranks, { so.num_ranks[so.domain_dim_idx(dim, "set_num_ranks")] = n; so.invalidate(); })
SOLN_GET(yk_solution_get_num_ranks, so.num_ranks[so.domain_dim_idx(dim, "get_num_ranks")])
SOLN_SET(yk_solution_set_rank_index, { so.rank_index[so.domain_dim_idx(dim, "set_rank_index")] = n; so.rank_index_set = true; so.invalidate(); })
SOLN_GET(yk_solution_get_rank_index, so.rank_index[so.domain_dim_idx(dim, "get_rank_index")])

int yk_solution_set_block_size(yk_soln_h s, const char* dim, yk_idx_t n) {
    YK_TRY
    Solution& so = S(s);
    if (so.step_dim_name == dim) so.block_size[0] = n;
    else so.block_size[1 + so.domain_dim_idx(dim, "set_block_size")] = n;
    return 0;
    YK_CATCH(1)
}
yk_idx_t yk_solution_get_block_size(yk_soln_h s, const char* dim) {
    YK_TRY
    Solution& so = S(s);
    if (so.step_dim_name == dim) return so.block_size[0];
    return so.block_size[1 + so.domain_dim_idx(dim, "get_block_size")];
    YK_CATCH(0)
}

int yk_solution_apply_command_line_options(yk_soln_h s, const char* args, char* rem, size_t cap) {
    YK_TRY
    Solution& so = S(s);
    std::vector<std::string> toks;
    std::istringstream is(args ? args : "");
    std::string t;
    while (is >> t) toks.push_back(t);
    std::string r = so.apply_command_line_options(toks);
    if (rem && cap) { std::strncpy(rem, r.c_str(), cap - 1); rem[cap - 1] = 0; }
    return 0;
    YK_CATCH(1)
}
const char* yk_solution_get_command_line_help(yk_soln_h s) {
    YK_TRY
    s->help = S(s).get_command_line_help();
    return s->help.c_str();
    YK_CATCH("")
}
const char* yk_solution_get_command_line_values(yk_soln_h s) {
    YK_TRY
    s->values = S(s).get_command_line_values();
    return s->values.c_str();
    YK_CATCH("")
}
int yk_solution_get_num_vars(yk_soln_h s) { return s ? (int)s->soln->vars.size() : 0; }
yk_var_h yk_solution_get_var(yk_soln_h s, const char* name) {
    YK_TRY
    return reinterpret_cast<yk_var_h>(S(s).get_var(name ? name : "").get());
    YK_CATCH(nullptr)
}
yk_var_h yk_solution_get_var_by_index(yk_soln_h s, int i) {
    YK_TRY
    Solution& so = S(s);
    if (i < 0 || i >= (int)so.vars.size()) YKH_THROW("var index out of range");
    return reinterpret_cast<yk_var_h>(so.vars[i].get());
    YK_CATCH(nullptr)
}
int yk_solution_prepare(yk_soln_h s) { YK_TRY S(s).prepare(); return 0; YK_CATCH(1) }
yk_idx_t yk_solution_get_first_rank_domain_index(yk_soln_h s, const char* dim) {
    YK_TRY
    Solution& so = S(s);
    return so.rank_ofs[so.domain_dim_idx(dim, "get_first_rank_domain_index")];
    YK_CATCH(0)
}
yk_idx_t yk_solution_get_last_rank_domain_index(yk_soln_h s, const char* dim) {
    YK_TRY
    Solution& so = S(s);
    int d = so.domain_dim_idx(dim, "get_last_rank_domain_index");
    return so.rank_ofs[d] + so.local_size[d] - 1;
    YK_CATCH(0)
}
int yk_solution_run(yk_soln_h s, yk_idx_t a, yk_idx_t b) { YK_TRY S(s).run(a, b); return 0; YK_CATCH(1) }
int yk_solution_end(yk_soln_h s) { YK_TRY S(s).end(); return 0; YK_CATCH(1) }
int yk_solution_exchange_halos(yk_soln_h s) { YK_TRY S(s).exchange_halos_all(); S(s).synchronize(); S(s).check_async_errors("exchange_halos()"); return 0; YK_CATCH(1) }
int yk_solution_copy_vars_to_device(yk_soln_h s) { YK_TRY S(s).copy_vars_to_device(); return 0; YK_CATCH(1) }
int yk_solution_copy_vars_from_device(yk_soln_h s) { YK_TRY S(s).copy_vars_from_device(); return 0; YK_CATCH(1) }
int yk_solution_get_stats(yk_soln_h s, yk_stats_t* out) {
    YK_TRY
    if (!out) YKH_THROW("null stats pointer");
    Stats st = S(s).get_stats();
    out->num_elements = st.num_elements; out->num_steps_done = st.num_steps_done;
    out->num_writes_done = st.num_writes_done; out->est_fp_ops_done = st.est_fp_ops_done;
    out->elapsed_secs = st.elapsed_secs; out->num_reads_done = st.num_reads_done;
    out->halo_secs = st.halo_secs; out->points_per_sec = st.pts_per_sec;
    out->halo_pack_secs = st.halo_pack_secs; out->halo_xfer_secs = st.halo_xfer_secs; out->halo_unpack_secs = st.halo_unpack_secs;
    out->halo_wait_secs = st.halo_wait_secs; out->exterior_secs = st.exterior_secs; out->interior_secs = st.interior_secs;
    out->halo_bytes_sent = st.halo_bytes_sent; out->halo_bytes_recv = st.halo_bytes_recv; out->halo_msgs_sent = st.halo_msgs_sent;
    out->fused_passes = st.fused_passes;
    out->graph_replays = st.graph_replays;
    out->graph_steps = st.graph_steps;
    return 0;
    YK_CATCH(1)
}
int yk_solution_get_step_times(yk_soln_h s, float* ms, int cap) {
    YK_TRY
    Solution& so = S(s);
    const int n = (int)so.step_ms.size();
    for (int i = 0; i < n && i < cap && ms; i++) ms[i] = so.step_ms[i];
    return n;
    YK_CATCH(-1)
}
int yk_env_transport_loopback(yk_env_h e, size_t nbytes) {
    YK_TRY
    if (!e) YKH_THROW("null env handle");
    Env& env = *e->env;
    if (!env.exch_start) YKH_THROW("transport loop-back: no halo transport is installed in this env");
    if (nbytes == 0) nbytes = 1 << 20;
    struct Bufs {
        unsigned char *s = nullptr, *r = nullptr; hipStream_t st = nullptr;
        ~Bufs() { if (s) (void)hipFree(s); if (r) (void)hipFree(r); if (st) (void)hipStreamDestroy(st); }
    } b;
    YKH_HIP(hipMalloc(&b.s, nbytes));
    YKH_HIP(hipMalloc(&b.r, nbytes));
    YKH_HIP(hipStreamCreateWithFlags(&b.st, hipStreamNonBlocking));
    std::vector<unsigned char> h(nbytes), back(nbytes, 0);
    for (size_t i = 0; i < nbytes; i++) h[i] = (unsigned char)((i * 2654435761u) >> 13);
    YKH_HIP(hipMemcpyAsync(b.s, h.data(), nbytes, hipMemcpyHostToDevice, b.st));
    YKH_HIP(hipMemsetAsync(b.r, 0, nbytes, b.st));
    HaloMsg m;
    m.peer = env.rank; m.send_buf = b.s; m.recv_buf = b.r; m.send_bytes = m.recv_bytes = nbytes; m.tag = 13; m.key = 0;
    if (env.exch_start(env.user, 1, &m, (void*)b.st) != 0) YKH_THROW("transport loop-back: start failed");
    if (env.exch_wait && env.exch_wait(env.user, 1, &m, (void*)b.st) != 0) YKH_THROW("transport loop-back: wait failed");
    YKH_HIP(hipMemcpyAsync(back.data(), b.r, nbytes, hipMemcpyDeviceToHost, b.st));
    YKH_HIP(hipStreamSynchronize(b.st));
    if (std::memcmp(h.data(), back.data(), nbytes) != 0) YKH_THROW("transport loop-back: received bytes differ from the bytes sent");
    if (env.allreduce) {
        long long v = 7 + env.rank;
        if (env.allreduce(env.user, 0, &v) != 0) YKH_THROW("transport loop-back: all-reduce failed");
        long long want = 0;
        for (int r = 0; r < env.nranks; r++) want += 7 + r;
        if (v != want) YKH_THROW("transport loop-back: all-reduce returned a wrong sum");
    }
    return 0;
    YK_CATCH(1)
}
double yk_env_probe_bandwidth(yk_env_h e, int kind, size_t bytes, int reps) {
    YK_TRY
    if (!e) YKH_THROW("null env handle");
    return probe_bandwidth(kind, bytes, reps);
    YK_CATCH(-1.0)
}
int yk_solution_clear_stats(yk_soln_h s) { YK_TRY (void)S(s).get_stats(); return 0; YK_CATCH(1) }
int yk_solution_set_min_pad_size(yk_soln_h s, const char* dim, yk_idx_t n) {
    YK_TRY Solution& so = S(s); so.min_pad[so.domain_dim_idx(dim, "set_min_pad_size")] = n; so.invalidate(); return 0; YK_CATCH(1)
}
yk_idx_t yk_solution_get_min_pad_size(yk_soln_h s, const char* dim) {
    YK_TRY Solution& so = S(s); return so.min_pad[so.domain_dim_idx(dim, "get_min_pad_size")]; YK_CATCH(0)
}
int yk_solution_set_step_wrap(yk_soln_h s, int w) { YK_TRY S(s).step_wrap = (w != 0); return 0; YK_CATCH(1) }
int yk_solution_get_step_wrap(yk_soln_h s) { return s && s->soln->step_wrap ? 1 : 0; }
int yk_solution_reset_auto_tuner(yk_soln_h s, int enable, int) { YK_TRY S(s).reset_auto_tuner(enable != 0); return 0; YK_CATCH(1) }
int yk_solution_is_auto_tuner_enabled(yk_soln_h s) { return s && s->soln->auto_tune ? 1 : 0; }
int yk_solution_run_auto_tuner_now(yk_soln_h s, int verbose) {
    YK_TRY
    Solution& so = S(s);
    bool old = so.env->trace;
    if (verbose) so.env->trace = true;
    so.run_auto_tuner_now();
    so.env->trace = old;
    return 0;
    YK_CATCH(1)
}
yk_var_h yk_solution_new_var(yk_soln_h s, const char* name, int nd, const char* const* dims) {
    YK_TRY
    std::vector<std::string> d;
    for (int i = 0; i < nd; i++) d.push_back(dims[i]);
    return reinterpret_cast<yk_var_h>(S(s).new_var(name, d).get());
    YK_CATCH(nullptr)
}
yk_var_h yk_solution_new_fixed_size_var(yk_soln_h s, const char* name, int nd, const char* const* dims,
                                        const yk_idx_t* sizes) {
    YK_TRY
    std::vector<std::string> d;
    std::vector<idx_t> sz;
    for (int i = 0; i < nd; i++) { d.push_back(dims[i]); sz.push_back(sizes[i]); }
    return reinterpret_cast<yk_var_h>(S(s).new_fixed_size_var(name, d, sz).get());
    YK_CATCH(nullptr)
}
yk_idx_t yk_solution_compare_data(yk_soln_h s, yk_soln_h ref, double eps) {
    YK_TRY
    return S(s).compare_data(S(ref), eps);
    YK_CATCH(-1)
}
int yk_solution_set_streams(yk_soln_h s, void* c, void* m) {
    YK_TRY S(s).set_streams((hipStream_t)c, (hipStream_t)m); return 0; YK_CATCH(1)
}
const char* yk_solution_get_kernel_variant(yk_soln_h s, int part) {
    YK_TRY
    Solution& so = S(s);
    if (part < 0 || part >= (int)so.impl.parts.size()) YKH_THROW("part index out of range");
    int v = so.part_variant[part];
    if (v < 0) v = so.impl.parts[part].default_variant;
    std::ostringstream os;
    os << so.impl.parts[part].variants[v].name;
    s->variant = os.str();
    return s->variant.c_str();
    YK_CATCH("")
}
int yk_solution_get_num_kernel_variants(yk_soln_h s, int part) {
    YK_TRY
    Solution& so = S(s);
    if (part < 0 || part >= (int)so.impl.parts.size()) YKH_THROW("part index out of range");
    return (int)so.impl.parts[part].variants.size();
    YK_CATCH(0)
}
const char* yk_solution_get_kernel_variant_name(yk_soln_h s, int part, int i) {
    YK_TRY
    Solution& so = S(s);
    if (part < 0 || part >= (int)so.impl.parts.size()) YKH_THROW("part index out of range");
    if (i < 0 || i >= (int)so.impl.parts[part].variants.size()) YKH_THROW("variant index out of range");
    return so.impl.parts[part].variants[i].name;
    YK_CATCH("")
}
yk_idx_t yk_solution_get_kernel_variant_scratch_bytes(yk_soln_h s, int part, int i) {
    YK_TRY
    Solution& so = S(s);
    if (part < 0 || part >= (int)so.impl.parts.size()) YKH_THROW("part index out of range");
    if (i < 0 || i >= (int)so.impl.parts[part].variants.size()) YKH_THROW("variant index out of range");
    return (yk_idx_t)variant_scratch_bytes(so.impl.parts[part].variants[i]);
    YK_CATCH(-1)
}
int yk_solution_get_part_bounding_box(yk_soln_h s, int part, yk_idx_t* first, yk_idx_t* last) {
    YK_TRY
    Solution& so = S(s);
    if (part < 0 || part >= (int)so.impl.parts.size()) YKH_THROW("part index out of range");
    if (!so.prepared) YKH_THROW("get_part_bounding_box() called without calling prepare_solution() first");
    const bool has = (size_t)part < so.part_has_bb.size() && so.part_has_bb[part];
    const Box b = has ? so.part_bb[part] : so.rank_box();
    for (int d = 0; d < 3; d++) { first[d] = d < MAX_DOMAIN_DIMS ? b.lo[d] : 0; last[d] = d < MAX_DOMAIN_DIMS ? b.hi[d] - 1 : 0; }
    return !has ? 0 : (b.empty() ? 2 : 1);
    YK_CATCH(-1)
}
int yk_solution_get_part_full_boxes(yk_soln_h s, int part, int cap, yk_idx_t* first, yk_idx_t* last) {
    YK_TRY
    Solution& so = S(s);
    if (part < 0 || part >= (int)so.impl.parts.size()) YKH_THROW("part index out of range");
    if (!so.prepared) YKH_THROW("get_part_full_boxes() called without calling prepare_solution() first");
    if ((size_t)part >= so.part_boxes.size()) return 0;
    const auto& bl = so.part_boxes[part];
    for (int i = 0; i < (int)bl.size() && i < cap; i++)
        for (int d = 0; d < 3; d++) { first[3 * i + d] = bl[i].lo[d]; last[3 * i + d] = bl[i].hi[d] - 1; }
    return (int)bl.size();
    YK_CATCH(-1)
}
int yk_solution_get_fused_groups(yk_soln_h s, int g, long long* info) {
    YK_TRY
    Solution& so = S(s);
    if (!so.prepared) YKH_THROW("get_fused_groups() called without calling prepare_solution() first");
    if (!so.fused_on) return 0;
    // (impl.fused lists every tile shape of every group: a group is reported once, on the shape in use)
    std::vector<const FusedGroupImpl*> used;
    for (auto& f : so.impl.fused)
        if (so.fused_group_at(f.first_stage) == &f) used.push_back(&f);
    const int n = (int)used.size();
    if (info && g >= 0 && g < n) {
        const FusedGroupImpl& fg = *used[(size_t)g];
        info[0] = fg.n_parts; info[1] = fg.n_scratch_vars; info[2] = fg.n_slots; info[3] = (long long)fg.lds_bytes; info[4] = fg.ti; info[5] = fg.tj;
    }
    return n;
    YK_CATCH(-1)
}
int yk_solution_get_part_info(yk_soln_h s, int part, yk_part_info_t* out) {
    YK_TRY
    Solution& so = S(s);
    if (part < 0 || part >= (int)so.impl.parts.size() || !out) YKH_THROW("part index out of range");
    if (!so.prepared) YKH_THROW("get_part_info() called without calling prepare_solution() first");
    const PartMeta& pm = *so.impl.parts[part].meta;
    std::memset(out, 0, sizeof(*out));
    out->name = pm.name; out->stage = pm.stage; out->is_scratch = pm.is_scratch;
    out->has_condition = pm.has_domain_cond || pm.has_step_cond || pm.has_step_cond_dev;
    out->fp_ops = pm.fp_ops; out->points_read = pm.points_read; out->points_written = pm.points_written;
    std::vector<char> rd(pm.n_groups, 0), wr(pm.n_groups, 0);
    for (int i = 0; i < pm.n_reads; i++) rd[pm.reads[i].g] = 1;
    for (int i = 0; i < pm.n_writes; i++) wr[pm.writes[i]] = 1;
    const int ndom = so.ndd + (so.has_outer ? 1 : 0);
    for (int g = 0; g < pm.n_groups; g++) {
        const VarMeta& vm = so.meta->vars[pm.groups[g].var];
        int nd = 0;
        for (int d = 0; d < vm.ndims; d++) { const int ty = so.meta->dims[vm.dims[d]].type; nd += ty == DIM_DOMAIN || ty == DIM_OUTER; }
        if (nd < ndom) continue;                       // a line or plane of coefficients: cache-resident
        if (vm.is_scratch) { out->scratch_arrays_read += rd[g]; out->scratch_arrays_written += wr[g]; }
        else { out->arrays_read += rd[g]; out->arrays_written += wr[g]; }
    }
    const bool has = (size_t)part < so.part_has_bb.size() && so.part_has_bb[part];
    const Box b = has ? so.part_bb[part] : so.rank_box();
    yk_idx_t pts = b.empty() ? 0 : 1;
    for (int d = 0; d < so.ndd && pts; d++) pts *= b.hi[d] - b.lo[d];
    if ((size_t)part < so.part_boxes.size() && !so.part_boxes[part].empty()) {      // the valid points, not their bounding box
        pts = 0;
        for (const Box& fb : so.part_boxes[part]) { yk_idx_t v = 1; for (int d = 0; d < so.ndd; d++) v *= fb.hi[d] - fb.lo[d]; pts += v; }
    }
    if (so.has_outer) pts *= so.local_size[3];
    out->points = pts;
    out->compulsory_bytes_per_point = (double)(out->arrays_read + out->arrays_written) * so.elem_bytes();
    return 0;
    YK_CATCH(1)
}
int yk_solution_time_part(yk_soln_h s, int part, int variant, yk_idx_t xchunk, yk_idx_t t, int reps, float* ms) {
    YK_TRY
    Solution& so = S(s);
    if (!so.prepared) YKH_THROW("yk_solution_time_part() called without calling prepare_solution() first");
    if (part < 0 || part >= (int)so.impl.parts.size()) YKH_THROW("part index out of range");
    if (variant < 0) { variant = so.part_variant[part]; if (xchunk <= 0) xchunk = so.part_xchunk[part]; }
    if (variant >= (int)so.impl.parts[part].variants.size()) YKH_THROW("variant index out of range");
    hipEvent_t e0, e1;
    YKH_HIP(hipEventCreate(&e0));
    YKH_HIP(hipEventCreate(&e1));
    Box rb = so.rank_box();
    YKH_HIP(hipEventRecord(e0, so.compute_stream));
    for (int i = 0; i < reps; i++) so.launch_part_variant(part, variant, xchunk, t + i, rb, so.compute_stream);
    YKH_HIP(hipEventRecord(e1, so.compute_stream));
    YKH_HIP(hipEventSynchronize(e1));
    float m = 0;
    YKH_HIP(hipEventElapsedTime(&m, e0, e1));
    if (ms) *ms = m / (reps > 0 ? reps : 1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
    YK_CATCH(1)
}

int yk_solution_time_part_box(yk_soln_h s, int part, int variant, yk_idx_t xchunk, const yk_idx_t* first, const yk_idx_t* last,
                              yk_idx_t t, int reps, float* ms) {
    YK_TRY
    Solution& so = S(s);
    if (!so.prepared) YKH_THROW("yk_solution_time_part_box() called without calling prepare_solution() first");
    if (part < 0 || part >= (int)so.impl.parts.size()) YKH_THROW("part index out of range");
    if (variant < 0) { variant = so.part_variant[part]; if (xchunk <= 0) xchunk = so.part_xchunk[part]; }
    if (variant >= (int)so.impl.parts[part].variants.size()) YKH_THROW("variant index out of range");
    if (!first || !last) YKH_THROW("null box");
    Box b = so.rank_box();
    for (int d = 0; d < so.ndd; d++) {
        b.lo[d] = std::max<idx_t>(b.lo[d], first[d]);
        b.hi[d] = std::min<idx_t>(b.hi[d], last[d] + 1);
    }
    if (b.empty()) YKH_THROW("empty box");
    hipEvent_t e0, e1;
    YKH_HIP(hipEventCreate(&e0));
    YKH_HIP(hipEventCreate(&e1));
    so.launch_part_variant(part, variant, xchunk, t, b, so.compute_stream);       // warm-up
    YKH_HIP(hipEventRecord(e0, so.compute_stream));
    for (int i = 0; i < reps; i++) so.launch_part_variant(part, variant, xchunk, t + i, b, so.compute_stream);
    YKH_HIP(hipEventRecord(e1, so.compute_stream));
    YKH_HIP(hipEventSynchronize(e1));
    float m = 0;
    YKH_HIP(hipEventElapsedTime(&m, e0, e1));
    if (ms) *ms = m / (reps > 0 ? reps : 1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
    YK_CATCH(1)
}

int yk_solution_time_decomposed_step(yk_soln_h s, const int* has_lo3, const int* has_hi3, int reps, float* ms3) {
    YK_TRY
    if (!has_lo3 || !has_hi3 || !ms3) YKH_THROW("null argument");
    bool lo[MAX_DOMAIN_DIMS], hi[MAX_DOMAIN_DIMS];
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++) { lo[d] = has_lo3[d] != 0; hi[d] = has_hi3[d] != 0; }
    S(s).time_decomposed_step(lo, hi, reps, ms3);
    return 0;
    YK_CATCH(1)
}

// ---- device-free planning
int yk_plan_rank(int ndims, int num_ranks, int rank, yk_rank_plan_t* plan) {
    YK_TRY
    if (!plan || ndims < 1 || ndims > MAX_DOMAIN_DIMS) YKH_THROW("yk_plan_rank: bad arguments");
    if (num_ranks < 1 || rank < 0 || rank >= num_ranks) YKH_THROW("invalid rank " + std::to_string(rank) + " of " + std::to_string(num_ranks));
    RankPlan p;
    for (int d = 0; d < ndims; d++) {
        p.global_size[d] = plan->global_size[d];
        p.rank_size[d] = plan->global_size[d] > 0 ? 0 : plan->local_size[d];
        p.num_ranks[d] = plan->num_ranks[d];
    }
    try { plan_rank(p, ndims, num_ranks, rank, {}, false); } catch (const PlanError& e) { YKH_THROW(e.what()); }
    for (int d = 0; d < ndims; d++) {
        plan->global_size[d] = p.global_size[d]; plan->local_size[d] = p.local_size[d]; plan->num_ranks[d] = p.num_ranks[d];
        plan->rank_index[d] = p.rank_index[d]; plan->rank_offset[d] = p.rank_ofs[d];
    }
    plan->num_neighbors = (int)p.neighbors.size();
    for (size_t i = 0; i < p.neighbors.size() && i < 26; i++) {
        plan->neighbor_rank[i] = p.neighbors[i].rank;
        for (int d = 0; d < 3; d++) plan->neighbor_offset[i][d] = p.neighbors[i].ofs[d];
    }
    return 0;
    YK_CATCH(1)
}
int yk_plan_halo_slab(int ndims, const yk_rank_plan_t* plan, const int* nofs, const yk_idx_t* hl, const yk_idx_t* hr,
                      int l1_norm, int sending, yk_box_t* box) {
    YK_TRY
    if (!plan || !nofs || !hl || !hr || !box || ndims < 1 || ndims > MAX_DOMAIN_DIMS) YKH_THROW("yk_plan_halo_slab: bad arguments");
    VarGeom g;
    PlanNeighbor nb;
    nb.rank = -1; nb.l1 = 0;
    idx_t nr[3] = {1, 1, 1}, ri[3] = {0, 0, 0};
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++) {
        g.uses_domain[d] = d < ndims;
        g.dom_size[d] = d < ndims ? plan->local_size[d] : 1;
        g.halo_l[d] = d < ndims ? hl[d] : 0; g.halo_r[d] = d < ndims ? hr[d] : 0;
        nb.ofs[d] = d < ndims ? nofs[d] : 0;
        nb.l1 += nb.ofs[d] < 0 ? -nb.ofs[d] : nb.ofs[d];
        if (d < ndims) { nr[d] = plan->num_ranks[d]; ri[d] = plan->rank_index[d]; }
    }
    g.l1_norm = l1_norm;
    idx_t lo[3], n[3];
    if (!plan_halo_slab(ndims, nr, ri, g, nb, sending != 0, lo, n)) return 0;
    for (int d = 0; d < 3; d++) { box->first[d] = lo[d]; box->size[d] = n[d]; }
    return 1;
    YK_CATCH(-1)
}

int yk_plan_wavefront(yk_idx_t lo, yk_idx_t hi, yk_idx_t width, yk_idx_t angle, yk_idx_t nphases, yk_idx_t* out3, int cap) {
    YK_TRY
    auto v = plan_wavefront(lo, hi, width, angle, nphases);
    for (int i = 0; i < (int)v.size() && i < cap && out3; i++) { out3[3 * i] = v[i].phase; out3[3 * i + 1] = v[i].lo; out3[3 * i + 2] = v[i].hi; }
    return (int)v.size();
    YK_CATCH(-1)
}

int yk_plan_blocks(const yk_idx_t* n3, const int* has_lo3, const int* has_hi3, const yk_idx_t* width3, int tile_y, int tile_z,
                   int overhead, int num_cus, int shell_pct, int mode, yk_block_desc_t* out, int cap, yk_idx_t* info5) {
    YK_TRY
    if (!n3 || !has_lo3 || !has_hi3 || !width3) YKH_THROW("yk_plan_blocks: null argument");
    BlockPlanIn in;
    for (int d = 0; d < 3; d++) { in.n[d] = n3[d]; in.has_lo[d] = has_lo3[d] != 0; in.has_hi[d] = has_hi3[d] != 0; in.width[d] = width3[d]; }
    in.ty = tile_y; in.tz = tile_z; in.overhead = overhead; in.ncu = num_cus; in.shell_frac = shell_pct / 100.0; in.mode = mode;
    BlockPlan p;
    try { p = plan_blocks(in); } catch (const PlanError& e) { YKH_THROW(e.what()); }
    static_assert(sizeof(yk_block_desc_t) == sizeof(BlockDesc), "yk_block_desc_t mirrors ykh::BlockDesc");
    for (int i = 0; i < (int)p.blocks.size() && i < cap && out; i++) std::memcpy(&out[i], &p.blocks[i], sizeof(BlockDesc));
    if (info5) { info5[0] = p.mode_used == 4 ? p.cut : p.n_signal; info5[1] = p.shell_done; info5[2] = p.makespan; info5[3] = p.undivided; info5[4] = p.mode_used; }
    return (int)p.blocks.size();
    YK_CATCH(-1)
}
int yk_plan_halves(yk_idx_t nx, yk_idx_t xwidth, yk_idx_t* q1, yk_idx_t* q2) {
    YK_TRY
    if (!q1 || !q2) YKH_THROW("yk_plan_halves: null argument");
    idx_t a = 0, b = 0;
    const bool ok = halves_split(nx, xwidth, &a, &b);
    *q1 = a; *q2 = b;
    return ok ? 1 : 0;
    YK_CATCH(-1)
}
int yk_plan_halves_slab(int half, int x_neighbor, yk_idx_t lo, yk_idx_t n, yk_idx_t q1, yk_idx_t q2, yk_idx_t* out4) {
    YK_TRY
    if (!out4 || half < 0 || half > 1) YKH_THROW("yk_plan_halves_slab: bad argument");
    idx_t l[2] = {0, 0}, m[2] = {0, 0};
    const int k = halves_slab_ranges(half, x_neighbor != 0, lo, n, q1, q2, l, m);
    for (int i = 0; i < 2; i++) { out4[2 * i] = l[i]; out4[2 * i + 1] = m[i]; }
    return k;
    YK_CATCH(-1)
}

// ---- var
const char* yk_var_get_name(yk_var_h v) { return v ? reinterpret_cast<Var*>(v)->name.c_str() : ""; }
int yk_var_get_num_dims(yk_var_h v) { return v ? (int)reinterpret_cast<Var*>(v)->dims.size() : 0; }
const char* yk_var_get_dim_name(yk_var_h v, int i) {
    YK_TRY
    Var* x = V(v);
    if (i < 0 || i >= (int)x->dims.size()) YKH_THROW("dim index out of range");
    return x->dims[i].name.c_str();
    YK_CATCH("")
}
int yk_var_is_dim_used(yk_var_h v, const char* dim) { YK_TRY return V(v)->dim_posn(dim, false) >= 0 ? 1 : 0; YK_CATCH(0) }
int yk_var_is_fixed_size(yk_var_h v) { YK_TRY return V(v)->fixed_size ? 1 : 0; YK_CATCH(0) }

#define VAR_GET(NAME, EXPR)                                  \
    yk_idx_t NAME(yk_var_h v, const char* dim) {             \
        YK_TRY                                               \
        Var* x = V(v);                                       \
        (void)dim;                                           \
        return (EXPR);                                       \
        YK_CATCH(0)                                          \
    }
VAR_GET(yk_var_get_first_local_index, x->first_local_index(x->dim_posn(dim)))
VAR_GET(yk_var_get_last_local_index, x->last_local_index(x->dim_posn(dim)))
VAR_GET(yk_var_get_alloc_size, x->alloc_size(x->dim_posn(dim)))
yk_idx_t yk_var_get_first_valid_step_index(yk_var_h v) {
    YK_TRY
    Var* x = V(v);
    if (!x->has_step) YKH_THROW("'get_first_valid_step_index' called on var '" + x->name + "' that does not use the step dimension");
    return x->first_valid_step;
    YK_CATCH(0)
}
yk_idx_t yk_var_get_last_valid_step_index(yk_var_h v) {
    YK_TRY
    Var* x = V(v);
    if (!x->has_step) YKH_THROW("'get_last_valid_step_index' called on var '" + x->name + "' that does not use the step dimension");
    return x->last_valid_step();
    YK_CATCH(0)
}
// (the outer dim of a solution with 4 domain dims is a domain dim for the API: one rank, offset 0, pads = its halo
//  or min pad; stored like a misc dim whose index range includes the pads)
static const VarDim* outer_dim(const Var* x, const char* dim) {
    for (auto& d : x->dims) if (d.is_outer && d.name == dim) return &d;
    return nullptr;
}
#define VAR_GET_DOM(NAME, FN, EXPR, OUTER)                   \
    yk_idx_t NAME(yk_var_h v, const char* dim) {             \
        YK_TRY                                               \
        Var* x = V(v);                                       \
        if (const VarDim* od = outer_dim(x, dim ? dim : "")) { const idx_t n = x->soln->local_size[3]; (void)n; return (OUTER); } \
        const int d = dom(x, dim, FN);                       \
        return (EXPR);                                       \
        YK_CATCH(0)                                          \
    }
VAR_GET_DOM(yk_var_get_rank_domain_size, "get_rank_domain_size", x->dom_size[d], n)
VAR_GET_DOM(yk_var_get_first_rank_domain_index, "get_first_rank_domain_index", x->rank_ofs[d], (void(od), 0))
VAR_GET_DOM(yk_var_get_last_rank_domain_index, "get_last_rank_domain_index", x->rank_ofs[d] + x->dom_size[d] - 1, (void(od), n - 1))
VAR_GET_DOM(yk_var_get_left_halo_size, "get_left_halo_size", x->halo_l[d], od->outer_halo_l)
VAR_GET_DOM(yk_var_get_right_halo_size, "get_right_halo_size", x->halo_r[d], od->outer_halo_r)
VAR_GET_DOM(yk_var_get_first_rank_halo_index, "get_first_rank_halo_index", x->rank_ofs[d] - x->halo_l[d], -od->outer_halo_l)
VAR_GET_DOM(yk_var_get_last_rank_halo_index, "get_last_rank_halo_index", x->rank_ofs[d] + x->dom_size[d] + x->halo_r[d] - 1, n - 1 + od->outer_halo_r)
VAR_GET_DOM(yk_var_get_left_pad_size, "get_left_pad_size", x->pad_l[d], -od->first_misc)
VAR_GET_DOM(yk_var_get_right_pad_size, "get_right_pad_size", x->pad_r[d], od->last_misc - (n - 1))
VAR_GET_DOM(yk_var_get_left_extra_pad_size, "get_left_extra_pad_size", x->pad_l[d] - x->halo_l[d], -od->first_misc - od->outer_halo_l)
VAR_GET_DOM(yk_var_get_right_extra_pad_size, "get_right_extra_pad_size", x->pad_r[d] - x->halo_r[d], od->last_misc - (n - 1) - od->outer_halo_r)
yk_idx_t yk_var_get_first_misc_index(yk_var_h v, const char* dim) {
    YK_TRY
    Var* x = V(v);
    int p = x->dim_posn(dim);
    if (x->dims[p].type != DIM_MISC) YKH_THROW("get_first_misc_index: '" + std::string(dim) + "' is not a misc dimension of var '" + x->name + "'");
    return x->dims[p].first_misc;
    YK_CATCH(0)
}
yk_idx_t yk_var_get_last_misc_index(yk_var_h v, const char* dim) {
    YK_TRY
    Var* x = V(v);
    int p = x->dim_posn(dim);
    if (x->dims[p].type != DIM_MISC) YKH_THROW("get_last_misc_index: '" + std::string(dim) + "' is not a misc dimension of var '" + x->name + "'");
    return x->dims[p].last_misc;
    YK_CATCH(0)
}

#define VAR_SET(NAME, BODY)                                         \
    int NAME(yk_var_h v, const char* dim, yk_idx_t n) {             \
        YK_TRY                                                      \
        Var* x = V(v);                                              \
        BODY;                                                       \
        return 0;                                                   \
        YK_CATCH(1)                                                 \
    }
// changing geometry of a var with storage requires re-preparing the solution
static void regeom(Var* x) {
    if (x->fixed_size) { x->compute_geometry(); if (x->is_allocated()) x->allocate(); }
    else x->soln->invalidate();
}
VAR_SET(yk_var_set_left_min_pad_size, { x->min_pad_l[dom(x, dim, "set_left_min_pad_size")] = n; regeom(x); })
VAR_SET(yk_var_set_right_min_pad_size, { x->min_pad_r[dom(x, dim, "set_right_min_pad_size")] = n; regeom(x); })
VAR_SET(yk_var_set_min_pad_size, { int d = dom(x, dim, "set_min_pad_size"); x->min_pad_l[d] = x->min_pad_r[d] = n; regeom(x); })
VAR_SET(yk_var_set_left_halo_size, { x->halo_l[dom(x, dim, "set_left_halo_size")] = n; regeom(x); })
VAR_SET(yk_var_set_right_halo_size, { x->halo_r[dom(x, dim, "set_right_halo_size")] = n; regeom(x); })
VAR_SET(yk_var_set_halo_size, { int d = dom(x, dim, "set_halo_size"); x->halo_l[d] = x->halo_r[d] = n; regeom(x); })
VAR_SET(yk_var_set_first_misc_index, {
    int p = x->dim_posn(dim);
    if (x->dims[p].type != DIM_MISC) YKH_THROW("set_first_misc_index: '" + std::string(dim) + "' is not a misc dimension");
    idx_t sz = x->dims[p].last_misc - x->dims[p].first_misc;
    x->dims[p].first_misc = n; x->dims[p].last_misc = n + sz;
})
VAR_SET(yk_var_set_alloc_size, {
    int p = x->dim_posn(dim);
    if (n < 1) YKH_THROW("set_alloc_size: size must be positive");
    if (x->dims[p].type == DIM_DOMAIN) YKH_THROW("set_alloc_size: cannot set the allocation of domain dimension '" + std::string(dim) + "'; use the pad-size calls");
    if (x->dims[p].type == DIM_STEP) {
        if (!x->dynamic_step_alloc) YKH_THROW("set_alloc_size: var '" + x->name + "' does not allow dynamic step allocation");
        x->nslots = (int)n; x->dirty.assign(n, 1);
    } else x->dims[p].last_misc = x->dims[p].first_misc + n - 1;
    regeom(x);
})

int yk_var_are_indices_local(yk_var_h v, const yk_idx_t* idx) { YK_TRY Var* x = V(v); return x->indices_local(vec(x, idx)) ? 1 : 0; YK_CATCH(0) }
double yk_var_get_element(yk_var_h v, const yk_idx_t* idx) { YK_TRY Var* x = V(v); return x->get_element(vec(x, idx)); YK_CATCH(0.0) }
yk_idx_t yk_var_set_element(yk_var_h v, double val, const yk_idx_t* idx, int strict) {
    YK_TRY Var* x = V(v); return x->set_element(val, vec(x, idx), strict != 0); YK_CATCH(0)
}
yk_idx_t yk_var_add_to_element(yk_var_h v, double val, const yk_idx_t* idx, int strict) {
    YK_TRY Var* x = V(v); return x->add_to_element(val, vec(x, idx), strict != 0); YK_CATCH(0)
}
yk_idx_t yk_var_get_elements_in_slice_f32(yk_var_h v, float* buf, size_t n, const yk_idx_t* f, const yk_idx_t* l) {
    YK_TRY Var* x = V(v); return x->get_elements_in_slice(buf, n, 4, vec(x, f), vec(x, l)); YK_CATCH(0)
}
yk_idx_t yk_var_get_elements_in_slice_f64(yk_var_h v, double* buf, size_t n, const yk_idx_t* f, const yk_idx_t* l) {
    YK_TRY Var* x = V(v); return x->get_elements_in_slice(buf, n, 8, vec(x, f), vec(x, l)); YK_CATCH(0)
}
yk_idx_t yk_var_set_elements_in_slice_f32(yk_var_h v, const float* buf, size_t n, const yk_idx_t* f, const yk_idx_t* l) {
    YK_TRY Var* x = V(v); return x->set_elements_in_slice(buf, n, 4, vec(x, f), vec(x, l)); YK_CATCH(0)
}
yk_idx_t yk_var_set_elements_in_slice_f64(yk_var_h v, const double* buf, size_t n, const yk_idx_t* f, const yk_idx_t* l) {
    YK_TRY Var* x = V(v); return x->set_elements_in_slice(buf, n, 8, vec(x, f), vec(x, l)); YK_CATCH(0)
}
yk_idx_t yk_var_set_elements_in_slice_same(yk_var_h v, double val, const yk_idx_t* f, const yk_idx_t* l, int strict) {
    YK_TRY Var* x = V(v); return x->set_elements_in_slice_same(val, vec(x, f), vec(x, l), strict != 0); YK_CATCH(0)
}
yk_idx_t yk_var_set_elements_in_slice_from_var(yk_var_h v, yk_var_h src, const yk_idx_t* fs, const yk_idx_t* ft, const yk_idx_t* lt) {
    YK_TRY
    Var* x = V(v); Var* y = V(src);
    if (x->dims.size() != y->dims.size()) YKH_THROW("set_elements_in_slice(): source and target vars have different numbers of dims");
    std::vector<idx_t> f = vec(x, ft), l = vec(x, lt), sf = vec(y, fs), sl(sf);
    size_t n = 1;
    for (size_t p = 0; p < f.size(); p++) { if (l[p] < f[p]) return 0; sl[p] = sf[p] + (l[p] - f[p]); n *= (size_t)(l[p] - f[p] + 1); }
    // staged through a host buffer in the target's precision (API convenience path, not a hot path)
    const int eb = x->elem_bytes();
    std::vector<char> buf(n * eb);
    y->get_elements_in_slice(buf.data(), n, eb, sf, sl);
    return x->set_elements_in_slice(buf.data(), n, eb, f, l);
    YK_CATCH(0)
}
int yk_var_set_all_elements_same(yk_var_h v, double val) { YK_TRY V(v)->set_all_elements_same(val); return 0; YK_CATCH(1) }
int yk_var_reduce_elements_in_slice(yk_var_h v, int mask, const yk_idx_t* f, const yk_idx_t* l, int strict, yk_reduction_t* out) {
    YK_TRY
    Var* x = V(v);
    if (!out) YKH_THROW("null reduction pointer");
    Var::Reduction r = x->reduce_elements_in_slice(mask, vec(x, f), vec(x, l), strict != 0);
    out->reduction_mask = mask; out->num_elements_reduced = r.n;
    out->sum = r.sum; out->sum_squares = r.sum_sq; out->product = r.prod; out->max = r.vmax; out->min = r.vmin;
    return 0;
    YK_CATCH(1)
}
int yk_var_get_halo_exchange_l1_norm(yk_var_h v) { YK_TRY return V(v)->l1_norm; YK_CATCH(0) }
int yk_var_set_halo_exchange_l1_norm(yk_var_h v, int n) { YK_TRY V(v)->l1_norm = n; V(v)->soln->invalidate(); return 0; YK_CATCH(1) }
int yk_var_is_dynamic_step_alloc(yk_var_h v) { YK_TRY return V(v)->dynamic_step_alloc ? 1 : 0; YK_CATCH(0) }
int yk_var_is_storage_allocated(yk_var_h v) { YK_TRY return V(v)->is_allocated() ? 1 : 0; YK_CATCH(0) }
yk_idx_t yk_var_get_num_storage_bytes(yk_var_h v) { YK_TRY return (yk_idx_t)V(v)->bytes(); YK_CATCH(0) }
yk_idx_t yk_var_get_num_storage_elements(yk_var_h v) { YK_TRY Var* x = V(v); return x->slot_elems * x->nslots; YK_CATCH(0) }
int yk_var_alloc_storage(yk_var_h v) {
    YK_TRY
    Var* x = V(v);
    if (!x->is_allocated()) x->compute_geometry();
    if (!x->storage_fits()) x->allocate();       // also when the step/misc allocation changed since
    return 0;
    YK_CATCH(1)
}
int yk_var_release_storage(yk_var_h v) { YK_TRY V(v)->release(); return 0; YK_CATCH(1) }
int yk_var_is_storage_layout_identical(yk_var_h v, yk_var_h o) {
    YK_TRY
    Var *a = V(v), *b = V(o);
    if (a->dims.size() != b->dims.size() || a->slot_elems != b->slot_elems || a->nslots != b->nslots) return 0;
    for (size_t i = 0; i < a->dims.size(); i++)
        if (a->dims[i].name != b->dims[i].name) return 0;
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++)
        if (a->stride[d] != b->stride[d] || a->pad_l[d] != b->pad_l[d] || a->dom_size[d] != b->dom_size[d]) return 0;
    return 1;
    YK_CATCH(0)
}
// fuse_vars (yk_var_api.hpp:1370-1396, yk_var_apis.cpp:334-367): `v` becomes another reference to `source` -- one
// allocation, shared by reference count, under two names (possibly in two solutions): what is written through one is read
// through the other, release_storage() on either applies to both, the storage is freed when the last holder goes.
int yk_var_fuse_vars(yk_var_h v, yk_var_h src) {
    YK_TRY
    Var *a = V(v), *b = V(src);
    if (a->dims.size() != b->dims.size()) YKH_THROW("fuse_vars: '" + a->name + "' and '" + b->name + "' have different numbers of dims");
    for (size_t i = 0; i < a->dims.size(); i++)
        if (a->dims[i].name != b->dims[i].name) YKH_THROW("fuse_vars: dims of '" + a->name + "' and '" + b->name + "' differ");
    // (a var used by its solution's kernels must keep the geometry the kernels were given: layouts must agree)
    if (a->nslots != b->nslots) YKH_THROW("fuse_vars: storage layouts of '" + a->name + "' and '" + b->name + "' differ (step allocations)");
    if ((b->is_allocated() || a->soln->prepared || b->soln->prepared) && !yk_var_is_storage_layout_identical(v, src))
        YKH_THROW("fuse_vars: storage layouts of '" + a->name + "' and '" + b->name + "' differ");
    b->before_device_use();
    // everything queued on either solution's stream so far sees the storage it was queued with
    YKH_HIP(hipStreamSynchronize(a->soln->compute_stream));
    YKH_HIP(hipStreamSynchronize(b->soln->compute_stream));
    a->fuse_with(*b);
    a->soln->drop_step_graphs();        // (captured launches hold the old addresses)
    return 0;
    YK_CATCH(1)
}
void* yk_var_get_raw_storage_buffer(yk_var_h v) { YK_TRY return V(v)->host_mirror(); YK_CATCH(nullptr) }
int yk_var_sync_raw_storage_to_device(yk_var_h v) { YK_TRY V(v)->sync_mirror_to_device(); return 0; YK_CATCH(1) }
int yk_var_release_raw_storage_buffer(yk_var_h v) { YK_TRY V(v)->release_raw_storage(); return 0; YK_CATCH(1) }
int yk_solution_get_placement_trials(yk_soln_h s, int* chosen, float* ms, int cap) {
    YK_TRY
    Solution& so = *s->soln;
    if (chosen) *chosen = so.placement_chosen;
    const int n = (int)so.placement_ms.size();
    for (int i = 0; i < n && i < cap; i++)
        if (ms) ms[i] = so.placement_ms[i];
    return n;
    YK_CATCH(-1)
}
void* yk_var_get_device_storage(yk_var_h v) { YK_TRY return V(v)->dptr; YK_CATCH(nullptr) }
int yk_var_set_elements_hash(yk_var_h v, double offset, double scale, int id) {
    YK_TRY V(v)->set_elements_hash(offset, scale, id); return 0; YK_CATCH(1)
}

}  // extern "C"
