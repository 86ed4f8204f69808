// ykh_solution.cpp -- the HIP launch-grid scheduler behind yk_solution.
//
// GPU re-design of the reference's L3 runtime on the run_solution() path:
//   StencilContext::run_solution / calc_mega_block / calc_block / calc_micro_block
//       (src/kernel/lib/context.cpp:220-1174)        -> Solution::run(): per step, per stage, per part
//       one kernel launch over the rank box (or exterior slabs + interior when halos are exchanged);
//   prepare_solution chain (src/kernel/lib/soln_apis.cpp:137-249; setup.cpp:169-524,666-805;
//       alloc.cpp:343-452)                            -> Solution::prepare(): rank grid, sizes, device allocation;
//   exchange_halos (src/kernel/lib/halo.cpp:80-491), buffer geometry alloc_mpi_data
//       (src/kernel/lib/alloc.cpp:456-859)            -> pack kernel -> transport (RCCL) -> unpack kernel on a
//       side stream, overlapped with the interior launch (exterior-first order of context.cpp:377-478);
//   get_stats (src/kernel/lib/soln_apis.cpp:349-562)  -> Solution::get_stats();
//   AutoTuner (src/kernel/lib/auto_tuner.cpp:206-586) -> run_auto_tuner_now(): times the compiled tile shapes;
//   KernelSettings::add_options (src/kernel/lib/settings.cpp:328-524) -> apply_command_line_options().
// The CPU block/mega-block/micro-block/nano-block/pico-block nest has no GPU meaning: its size
// options are accepted and remembered but only the HIP tile shape (tuner / -hip_variant) matters.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "ykh_boxes.hpp"
#include "ykh_runtime.hpp"
#include "ykh_solution_internal.hpp"

namespace ykh {

std::string version_string() { return "4.05.04-cdna4_hip"; }

bool Box::empty() const {
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++)
        if (hi[d] <= lo[d]) return true;
    return false;
}

// ------------------------------------------------------------------ Env
Env::Env() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        YKH_THROW("no HIP device is visible: the cdna4_hip kernel library needs an AMD GPU (there is no CPU fallback)");
    // The env lives on the device that is current when it is created -- the one the host (torch, the application) chose.  Binding
    // a launcher-started rank to "its" GPU is yk_env_init_from_launcher()'s job (ykh_launch.cpp), i.e. yk_factory::new_env()'s.
    (void)hipGetDevice(&device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) num_cus = prop.multiProcessorCount;
}
void Env::set_ranks(int r, int n) {
    if (n < 1 || r < 0 || r >= n) YKH_THROW("invalid rank " + std::to_string(r) + " of " + std::to_string(n));
    rank = r;
    nranks = n;
}
static long long env_reduce(const Env& e, int op, long long v) {
    if (e.nranks <= 1) return v;
    if (!e.allreduce) YKH_THROW("multi-rank env has no all-reduce transport installed");
    if (e.allreduce(e.user, op, &v) != 0) YKH_THROW("all-reduce transport failed");
    return v;
}
long long Env::sum_over_ranks(long long v) const { return env_reduce(*this, 0, v); }
long long Env::min_over_ranks(long long v) const { return env_reduce(*this, 1, v); }
long long Env::max_over_ranks(long long v) const { return env_reduce(*this, 2, v); }

size_t variant_scratch_bytes(const KernelVariant& kv) {
    size_t worst = 0;
    auto one = [&](const void* f) {
        hipFuncAttributes at;
        if (!f) return;
        if (hipFuncGetAttributes(&at, f) != hipSuccess) { (void)hipGetLastError(); return; }
        worst = std::max(worst, (size_t)at.localSizeBytes);
    };
    one(kv.func);
    for (int i = 0; i < kv.n_more_funcs; i++) one(kv.more_funcs[i]);      // (a cluster variant is K kernels, not its first)
    return worst;
}

// ------------------------------------------------------------------ Solution basics
Solution::Solution(std::shared_ptr<Env> e, const SolnImpl& im) : env(e), impl(im), meta(im.meta) {
    ordinal = env->solutions_made++;
    if (const char* e = getenv("YASK_HIP_FUSE_SCRATCH")) fuse_scratch_mode = atoi(e);
    ndd = 0;
    for (int i = 0; i < meta->ndims; i++) {
        const DimMeta& d = meta->dims[i];
        if (d.type == DIM_STEP) step_dim_name = d.name;
        else if (d.type == DIM_OUTER) { has_outer = true; outer_dim_name = d.name; }
        else if (d.type == DIM_DOMAIN) { domain_dim_names.push_back(d.name); ndd++; }
        else misc_dim_names.push_back(d.name);
    }
    if (ndd > MAX_DOMAIN_DIMS) YKH_THROW("cdna4_hip runtime supports at most 3 domain dims");
    for (int v = 0; v < meta->n_vars; v++) {
        auto var = std::make_shared<Var>(this, &meta->vars[v], v);
        if (meta->vars[v].is_scratch) { scratch_vars.push_back(var); continue; }   // (reference: one per thread, per block)
        vars.push_back(var);
        var_map[var->name] = var;
    }
    part_variant.assign(impl.parts.size(), -1);
    part_xchunk.assign(impl.parts.size(), 0);
    YKH_HIP(hipStreamCreateWithFlags(&compute_stream, hipStreamNonBlocking));
    {
        // halo traffic gets the highest stream priority: its small pack/RCCL/unpack kernels are dispatched
        // ahead of queued stencil workgroups whenever a CU frees up
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
        if (hipStreamCreateWithPriority(&comm_stream, hipStreamNonBlocking, hi) != hipSuccess)
            YKH_HIP(hipStreamCreateWithFlags(&comm_stream, hipStreamNonBlocking));
    }
    own_streams = true;
    YKH_HIP(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
    YKH_HIP(hipEventCreateWithFlags(&ev_shell, hipEventDisableTiming));
    YKH_HIP(hipEventCreateWithFlags(&ev_b, hipEventDisableTiming));
}

Solution::~Solution() {
    drop_step_graphs();
    drop_launch_plans();
    drop_fused_args();
    if (lockstep_dev) (void)hipFree(lockstep_dev);
    free_halo_buffers();
    vars.clear();
    scratch_vars.clear();
    var_map.clear();
    if (ev_a) (void)hipEventDestroy(ev_a);
    if (ev_shell) (void)hipEventDestroy(ev_shell);
    if (ev_b) (void)hipEventDestroy(ev_b);
    for (auto& ph : phase_pool)
        for (auto e : ph.e) if (e) (void)hipEventDestroy(e);
    for (auto e : step_events) if (e) (void)hipEventDestroy(e);
    if (own_streams) {
        if (compute_stream) (void)hipStreamDestroy(compute_stream);
        if (comm_stream) (void)hipStreamDestroy(comm_stream);
    }
}

void Solution::set_streams(hipStream_t c, hipStream_t m) {
    drop_step_graphs();
    if (own_streams) {
        (void)hipStreamDestroy(compute_stream);
        (void)hipStreamDestroy(comm_stream);
        own_streams = false;
    }
    compute_stream = c;
    comm_stream = m;
}

void Solution::synchronize() {
    YKH_HIP(hipStreamSynchronize(compute_stream));
    YKH_HIP(hipStreamSynchronize(comm_stream));
}

// Errors raised on the device while a call's work ran (a waiter that gave up): turned into an exception by the call that queued the
// work, once its streams have drained -- the error words are always consumed by the call that raised them, so that the next call
// does not start with every wait returning at once (exchange_halos() used to return success with stale halos).
void Solution::check_async_errors(const char* who) {
    if (env->nranks > 1 && env->exch_check && env->exch_check(env->user) != 0)
        YKH_THROW(std::string(who) + ": the halo transport reports a failed exchange");
}

int Solution::domain_dim_idx(const std::string& dim, const char* fn) const {
    for (int d = 0; d < ndd; d++)
        if (domain_dim_names[d] == dim) return d;
    if (has_outer && dim == outer_dim_name) return 3;        // (the per-dim setting arrays have a 4th entry for it)
    YKH_THROW(std::string(fn) + ": '" + dim + "' is not a domain dimension of solution '" + meta->name + "'");
}

std::shared_ptr<Var> Solution::get_var(const std::string& name) const {
    auto it = var_map.find(name);
    if (it == var_map.end()) YKH_THROW("var '" + name + "' not found");
    return it->second;
}

std::shared_ptr<Var> Solution::new_var(const std::string& name, const std::vector<std::string>& dims) {
    if (var_map.count(name)) YKH_THROW("var '" + name + "' already exists");
    auto v = std::make_shared<Var>(this, name, dims, (int)vars.size(), nullptr);
    vars.push_back(v);
    var_map[name] = v;
    if (prepared) { v->compute_geometry(); }
    return v;
}

std::shared_ptr<Var> Solution::new_fixed_size_var(const std::string& name, const std::vector<std::string>& dims,
                                                  const std::vector<idx_t>& sizes) {
    if (var_map.count(name)) YKH_THROW("var '" + name + "' already exists");
    auto v = std::make_shared<Var>(this, name, dims, (int)vars.size(), &sizes);
    vars.push_back(v);
    var_map[name] = v;
    v->compute_geometry();
    v->allocate();   // fixed-size vars get storage immediately (new_var.cpp)
    return v;
}

Box Solution::rank_box() const {
    Box b;
    for (int d = 0; d < MAX_DOMAIN_DIMS; d++) { b.lo[d] = 0; b.hi[d] = d < ndd ? local_size[d] : 1; }
    return b;
}

// ------------------------------------------------------------------ command-line options
// Same option names as KernelSettings::add_options (src/kernel/lib/settings.cpp:328-524); options
// that only steer the CPU loop nest / OpenMP are accepted and recorded. Unknown tokens are returned
// (soln_apis.cpp:297-313).
namespace {
struct OptSpec { const char* name; int kind; };   // kind: 0 bool, 1 int (per-dim family), 2 int, 3 double, 4 string
bool parse_idx(const std::string& s, idx_t& v) {
    if (s.empty()) return false;
    char* end = nullptr;
    long long x = strtoll(s.c_str(), &end, 10);
    if (*end) return false;
    v = x;
    return true;
}
}  // namespace

std::string Solution::apply_command_line_options(const std::vector<std::string>& args) {
    std::vector<std::string> rem;
    const char* dimfam_set[] = {"g", "l", "d", "b", "Mb", "mb", "nb", "pb", "mp", "ep", "nr", "ri", "r", "B", "sb"};
    const char* bool_opts[] = {"overlap_comms", "use_shm", "use_device_mpi", "force_scalar_exchange", "force_scalar",
                               "bind_inner_threads", "bundle_allocs", "init_scratch_vars", "auto_tune",
                               "allow_addl_padding", "round_up_temporal_angles", "print_suffixes", "verbose",
                               "exchange_halos", "auto_tune_each_stage", "trace", "hip_direct_halo", "hip_thin_slab_point_kernel",
                               "hip_step_timers", "hip_phase_timers", "hip_fast_div", "hip_planned_launch", "hip_halves"};
    const char* int_opts[] = {"hip_placement_trials", "hip_step_graphs", "hip_fuse_steps", "min_exterior", "max_threads", "outer_threads", "inner_threads", "numa_pref",
                              "auto_tune_radius", "thread_divisor", "block_threads", "hip_xchunk", "device_thread_limit"};
    const char* dbl_opts[] = {"auto_tune_trial_secs"};
    const char* str_opts[] = {"auto_tune_targets", "hip_variant"};
    for (size_t i = 0; i < args.size(); i++) {
        const std::string& tok = args[i];
        if (tok.size() < 2 || tok[0] != '-') { rem.push_back(tok); continue; }
        std::string opt = tok.substr(1);
        bool handled = false;
        // booleans, with -no- prefix
        {
            bool val = true;
            std::string b = opt;
            if (b.rfind("no-", 0) == 0) { val = false; b = b.substr(3); }
            for (auto bo : bool_opts)
                if (b == bo) {
                    handled = true;
                    if (b == "overlap_comms") overlap_comms = val;
                    else if (b == "exchange_halos") do_halo_exchange = val;
                    else if (b == "force_scalar") { force_scalar = val; }
                    else if (b == "auto_tune") { auto_tune = val; tune_at_prepare = val; }   // -no-auto_tune: no timing pass at all
                    else if (b == "hip_step_timers") step_timers = val;
                    else if (b == "hip_phase_timers") phase_timers = val;
                    else if (b == "hip_planned_launch") planned_launch = val;
                    else if (b == "hip_halves") halves = val;
                    else if (b == "trace") env->trace = val;
                    else if (b == "hip_direct_halo") { direct_halo = val; invalidate(); }
                    else if (b == "hip_thin_slab_point_kernel") thin_slab_point_kernel = val;
                    else if (b == "hip_fast_div") { fast_div = val; invalidate(); }
                    else ignored_opts[b] = val ? "true" : "false";
                }
        }
        if (handled) continue;
        auto next_val = [&](std::string& out) -> bool {
            if (i + 1 >= args.size()) return false;
            out = args[++i];
            return true;
        };
        for (auto io : int_opts)
            if (opt == io) {
                std::string v; idx_t n;
                if (!next_val(v) || !parse_idx(v, n)) YKH_THROW("option '-" + opt + "' requires an integer value");
                handled = true;
                if (opt == "min_exterior") { min_exterior = n; drop_launch_plans(); }
                else if (opt == "hip_xchunk") xchunk_override = n;
                else if (opt == "hip_fuse_steps") fuse_steps = n;
                else if (opt == "hip_step_graphs") step_graphs = n;
                else if (opt == "hip_placement_trials") { placement_trials = std::max<idx_t>(1, n); invalidate(); }
                else ignored_opts[opt] = v;
            }
        if (handled) continue;
        for (auto d : dbl_opts)
            if (opt == d) {
                std::string v;
                if (!next_val(v)) YKH_THROW("option '-" + opt + "' requires a value");
                auto_tune_trial_secs = atof(v.c_str());
                handled = true;
            }
        if (handled) continue;
        for (auto so : str_opts)
            if (opt == so) {
                std::string v;
                if (!next_val(v)) YKH_THROW("option '-" + opt + "' requires a value");
                if (opt == "hip_variant") variant_override = v;
                else ignored_opts[opt] = v;
                handled = true;
            }
        if (handled) continue;
        // per-dim families: -g 64 (all dims), -gx 64 (one dim); -bt for the step dim of block sizes
        for (auto fam : dimfam_set) {
            std::string f(fam);
            if (opt.rfind(f, 0) != 0) continue;
            std::string dim = opt.substr(f.size());
            int didx = -2;   // -2: no match, -1: all dims, >=0 domain dim, 100: step
            if (dim.empty()) didx = -1;
            else if (dim == step_dim_name && (f == "b" || f == "Mb")) didx = 100;
            else {
                for (int d = 0; d < ndd; d++)
                    if (dim == domain_dim_names[d]) didx = d;
                if (has_outer && dim == outer_dim_name) didx = 3;
            }
            if (didx == -2) continue;
            std::string v; idx_t n;
            if (!next_val(v) || !parse_idx(v, n)) YKH_THROW("option '-" + opt + "' requires an integer value");
            handled = true;
            auto apply = [&](int d) {
                if (f == "g") { global_size[d] = n; rank_size[d] = 0; invalidate(); }
                else if (f == "l" || f == "d") { rank_size[d] = n; global_size[d] = 0; invalidate(); }
                else if (f == "b") block_size[d + 1] = n;
                else if (f == "Mb") mega_block_size[d + 1] = n;
                else if (f == "mp") { min_pad[d] = n; invalidate(); }
                else if (f == "ep") { extra_pad[d] = n; invalidate(); }
                else if (f == "nr") { num_ranks[d] = n; invalidate(); }
                else if (f == "ri") { rank_index[d] = n; rank_index_set = true; invalidate(); }
                else ignored_opts[f + (d < ndd ? domain_dim_names[d] : outer_dim_name)] = v;
            };
            if (didx == 100) {
                // (with neighbours the number of wave-front steps sets the width of halos and pads: prepare_solution() again)
                if (f == "b") { if (block_size[0] != n && env->nranks > 1) invalidate(); block_size[0] = n; }
                else if (f == "Mb") { if (mega_block_size[0] != n && env->nranks > 1) invalidate(); mega_block_size[0] = n; }
                else ignored_opts[opt] = v;
            }
            else if (didx == -1) { for (int d = 0; d < ndd; d++) apply(d); if (has_outer) apply(3); }
            else apply(didx);
            break;
        }
        if (!handled) rem.push_back(tok);
    }
    std::string out;
    for (size_t i = 0; i < rem.size(); i++) out += (i ? " " : "") + rem[i];
    return out;
}

std::string Solution::get_command_line_help() const {
    std::ostringstream os;
    os << "Options of the cdna4_hip kernel library (names follow the YASK kernel options):\n"
          " -g<dim> <n>   global-domain size      -l<dim> <n>  local-domain (per-rank) size\n"
          " -nr<dim> <n>  number of ranks          -ri<dim> <n> this rank's index\n"
          " -mp<dim> <n>  minimum padding          -ep<dim> <n> extra padding\n"
          " -b<dim> <n>   block size (advisory; the HIP tile shape is what matters on the GPU)\n"
          " -Mbt <n> | -bt <n>  wave-front temporal tiling.  One rank: n steps are applied to one x-slab after the other, each step\n"
          "               shifted by the stencil's x-halo (the reference's mega-block wave-fronts); -Mbx <n> = slab width\n"
          "               (default 128); exact; slower than plain sweeps on this GPU (DESIGN.md 3.7).  Several ranks: halos grow by\n"
          "               (n x stages - 1) x the stencil halo, every rank evaluates the steps of a group on boxes that shrink towards\n"
          "               its own (redundantly with its neighbours) and halos are exchanged ONCE per n steps.  Exact.  Off by default.\n"
          " -[no-]overlap_comms   overlap halo exchange with interior computation\n"
          " -min_exterior <n>     minimum width of the exterior slabs\n"
          " -[no-]exchange_halos  perform halo exchanges\n"
          " -[no-]auto_tune       time the compiled HIP tile shapes at prepare_solution() (-no-auto_tune also disables the\n"
          "                       one-off timing of small grids / generic stencils: static default shapes, reproducible)\n"
          " -hip_fuse_steps <n>   2: two time steps per pass, fused on chip, for solutions that have such a kernel (3axis family;\n"
          "                       one rank); the in-between step stays on chip, an odd last step runs the plain kernel.\n"
          "                       0: never (default: a call over several steps is then bit-identical to one call per step;\n"
          "                       the fused pass differs in the last bits).  Measured 1.2-1.3x at radius 1, 0.6-0.8x at radius 4\n"
          " -[no-]hip_step_timers record one HIP event per step (per-step times of the last run)\n"
          " -[no-]hip_phase_timers multi-rank runs: HIP events around exterior / interior / pack / transport / unpack / wait of\n"
          "                       every stage (yk_stats time breakdown; a ring of 64 event sets; default on)\n"
          " -hip_step_graphs <n>  1: one-rank runs of several steps are captured once into a hipGraph (whole slot periods, up to\n"
          "                       ~256 launches) and replayed, one host call per replay; 0: plain launches.  Default: on for rank\n"
          "                       boxes of up to 2^20 points (measured: 64^3 +13 %, 128^3 and larger +-0.5 %).  Bit-identical.\n"
          " -auto_tune_trial_secs <s>\n"
          " -[no-]force_scalar    use the generic one-thread-per-point kernel\n"
          " -hip_variant <name>   force a kernel variant     -hip_xchunk <n>  x-march chunk length\n"
          " -hip_placement_trials <n>         prepare_solution() allocates the vars n times (from 256 MiB in total, memory permitting),\n"
          "                                   times a step on each set and keeps the fastest: where the arrays happen to lie in\n"
          "                                   memory is worth 3-4 % of a step (default 1: take the first allocation; the search\n"
          "                                   runs trial kernels and briefly holds two sets of arrays -- bench.py and the harnesses pass 6)\n"
          " -[no-]hip_fast_div                fp32 divisions as a * v_rcp_f32(b), <= 1.5 ulp, in the kernel shapes that have such a\n"
          "                                   form (ssg's defaults: 8 divisions per point were a third of the instructions); off: the\n"
          "                                   correctly rounded shapes, the reference's own arithmetic (default on)\n"
          " -[no-]hip_direct_halo             x-face halos of full-dim vars are sent/received in place (default on)\n"
          " -[no-]hip_thin_slab_point_kernel  thin y/z exterior slabs run on the point kernel (default on)\n"
          " -[no-]hip_planned_launch          decomposed runs (y or z neighbours): a stage whose one part runs on a marching kernel goes out\n"
          "                                   as whole-box launches of equal blocks instead of exterior slabs + interior (default on).\n"
          " -[no-]hip_halves                  ... as TWO launches in regular order, the outer and the inner half of the x range, each\n"
          "                                   followed by the exchange of its own part of the faces, which travels while the other half is\n"
          "                                   computed (default on; needs face-only reads of the written vars).  -no-hip_halves: ONE plan of\n"
          "                                   equal blocks, the blocks a neighbour needs first, the exchange released by an event behind\n"
          "                                   their rounds.  -no-overlap_comms: the whole box, then the exchange.\n"
          " CPU-only options (-Mb -mb -nb -pb -max_threads -outer_threads -inner_threads -numa_pref\n"
          "  -bind_inner_threads -bundle_allocs -use_shm -use_device_mpi ...) are accepted and ignored.\n";
    return os.str();
}

std::string Solution::get_command_line_values() const {
    std::ostringstream os;
    for (int d = 0; d < ndd; d++) os << " -g" << domain_dim_names[d] << " " << global_size[d];
    for (int d = 0; d < ndd; d++) os << " -l" << domain_dim_names[d] << " " << (prepared ? local_size[d] : rank_size[d]);
    for (int d = 0; d < ndd; d++) os << " -nr" << domain_dim_names[d] << " " << num_ranks[d];
    for (int d = 0; d < ndd; d++) os << " -ri" << domain_dim_names[d] << " " << rank_index[d];
    for (int d = 0; d < ndd; d++) os << " -b" << domain_dim_names[d] << " " << block_size[d + 1];
    if (fuse_steps > 0) os << " -hip_fuse_steps " << fuse_steps;
    if (step_graphs >= 0) os << " -hip_step_graphs " << step_graphs;
    if (!fast_div) os << " -no-hip_fast_div";
    os << (overlap_comms ? " -overlap_comms" : " -no-overlap_comms") << " -min_exterior " << min_exterior
       << (auto_tune ? " -auto_tune" : " -no-auto_tune") << (force_scalar ? " -force_scalar" : " -no-force_scalar");
    for (size_t p = 0; p < impl.parts.size(); p++)
        if (part_variant[p] >= 0) os << " -hip_variant[" << impl.parts[p].meta->name << "] " << impl.parts[p].variants[part_variant[p]].name;
    return os.str();
}

// ------------------------------------------------------------------ rank grid (setup.cpp:169-524)
// The arithmetic lives in ykh_plan.cpp (shared with the device-free yk_plan_* C entry points).
void Solution::setup_rank() {
    RankPlan p;
    for (int d = 0; d < ndd; d++) {
        p.global_size[d] = global_size[d]; p.rank_size[d] = rank_size[d];
        p.num_ranks[d] = num_ranks[d]; p.rank_index[d] = rank_index[d];
    }
    try {
        plan_rank(p, ndd, env->nranks, env->rank, domain_dim_names, rank_index_set);
    } catch (const PlanError& e) { YKH_THROW(e.what()); }
    neighbors.clear();
    for (int d = 0; d < ndd; d++) {
        global_size[d] = p.global_size[d]; num_ranks[d] = p.num_ranks[d]; rank_index[d] = p.rank_index[d];
        local_size[d] = p.local_size[d]; rank_ofs[d] = p.rank_ofs[d];
    }
    if (has_outer) {
        // the outer (4th) domain dim is never decomposed: one rank, whole extent
        if (env->nranks > 1) YKH_THROW("solution '" + std::string(meta->name) + "' has 4 domain dims: it runs on one rank");
        if (num_ranks[3] > 1) YKH_THROW("the '" + outer_dim_name + "' dim cannot be decomposed");
        local_size[3] = rank_size[3] > 0 ? rank_size[3] : global_size[3];
        if (local_size[3] < 1) YKH_THROW("domain size in the '" + outer_dim_name + "' dim is not set");
        global_size[3] = local_size[3];
        num_ranks[3] = 1; rank_index[3] = 0; rank_ofs[3] = 0;
    }
    for (auto& pn : p.neighbors) {
        Neighbor nb;
        nb.rank = pn.rank; nb.l1 = pn.l1;
        for (int d = 0; d < MAX_DOMAIN_DIMS; d++) nb.ofs[d] = pn.ofs[d];
        neighbors.push_back(nb);
    }
}

// ------------------------------------------------------------------ prepare / end
void Solution::prepare() {
    for (auto& h : before_prepare) h(*this);
    in_outer_loop = launching_exterior = launching_interior = false;
    cur_outer = 0;
    cur_phase = nullptr;
    setup_rank();
    // solution-wide pads = max halo over all vars, scratch vars included (see Var::compute_geometry): every var
    // over all domain dims then has the same strides, which the kernels rely on (P::group_full)
    for (int d = 0; d < ndd; d++) {
        shared_pad_l_[d] = shared_pad_r_[d] = 0;
        for (auto* list : {&vars, &scratch_vars})
            for (auto& v : *list) {
                if (v->fixed_size || !v->uses_domain[d]) continue;
                shared_pad_l_[d] = std::max({shared_pad_l_[d], v->halo_l[d], v->min_pad_l[d]});
                shared_pad_r_[d] = std::max({shared_pad_r_[d], v->halo_r[d], v->min_pad_r[d]});
            }
    }
    // Wave-front tiling across ranks (-Mbt / -bt n > 1 with neighbours): halos and pads grow by angle x (phases - 1) in
    // every decomposed dim (the reference's wf_shift_pts / left_wf_exts / right_wf_exts, setup.cpp:717-805).
    {
        const idx_t wf_steps = std::max<idx_t>(mega_block_size[0], block_size[0]);
        const idx_t shifts = wf_steps > 1 ? wf_steps * meta->n_stages - 1 : 0;
        for (int d = 0; d < MAX_DOMAIN_DIMS; d++) {
            wf_angle_[d] = 0;
            if (d < ndd)
                for (auto* list : {&vars, &scratch_vars})
                    for (auto& v : *list)
                        if (!v->fixed_size && v->uses_domain[d]) wf_angle_[d] = std::max({wf_angle_[d], v->halo_l[d], v->halo_r[d]});
            const bool split = d < ndd && !has_outer && num_ranks[d] > 1;
            wf_ext_[d] = (split && shifts > 0) ? wf_angle_[d] * shifts : 0;
        }
    }
    for (int d = 0; d < ndd; d++) {
        idx_t need = std::max(shared_pad_l_[d], shared_pad_r_[d]) + wf_ext_[d];
        if (num_ranks[d] > 1 && local_size[d] < need)
            YKH_THROW("local-domain size of " + std::to_string(local_size[d]) + " in '" + domain_dim_names[d] +
                      "' dim is less than the required halo size of " + std::to_string(need) +
                      (wf_ext_[d] > 0 ? " (stencil halo + temporal wave-front extension)" : ""));
    }
    std::vector<Var*> need_alloc;
    for (auto& v : vars) {
        if (v->fixed_size) continue;
        // keep existing storage only if the geometry is unchanged
        idx_t old_slot = v->slot_elems, old_ofs = v->origin_elems;
        idx_t old_stride[3] = {v->stride[0], v->stride[1], v->stride[2]};
        v->compute_geometry();
        bool same = v->storage_fits() && old_slot == v->slot_elems && old_ofs == v->origin_elems &&
                    old_stride[0] == v->stride[0] && old_stride[1] == v->stride[1] && old_stride[2] == v->stride[2];
        if (!same) need_alloc.push_back(v.get());
        v->set_dirty_all(true);
    }
    for (auto& v : scratch_vars) { v->compute_geometry(); need_alloc.push_back(v.get()); v->l1_norm = 0; }
    // (the placement search runs trial steps: only when no var of the solution holds data from before this call)
    bool placed = false, all_fresh = false;
    {
        size_t total = 0, movable = scratch_vars.size();
        for (auto& v : vars) movable += v->fixed_size ? 0 : 1;
        for (auto* v : need_alloc) { v->allocate(); total += v->bytes(); }
        all_fresh = !need_alloc.empty() && need_alloc.size() == movable;
        placed = all_fresh && total >= ((size_t)256 << 20);
        for (auto& v : vars) if (v->fuse_group) placed = false;      // storage shared with another solution's var stays where it is
    }
    free_halo_buffers();
    alloc_halo_buffers();
    drop_step_graphs();
    drop_launch_plans();
    // Sub-domain parts: bounding box of the condition inside this rank's domain (the reference's
    // find_bounding_box, src/kernel/lib/setup.cpp:1082-1169); the part is then launched over box ∩ bb only --
    // a free-surface condition `z == last_domain_index(z)` costs one plane instead of a sweep of the grid.
    part_bb.assign(impl.parts.size(), rank_box());
    part_has_bb.assign(impl.parts.size(), 0);
    part_bb_solid.assign(impl.parts.size(), 0);
    part_boxes.assign(impl.parts.size(), std::vector<Box>());
    part_hole.assign(impl.parts.size(), Box{{0, 0, 0}, {0, 0, 0}});
    part_box_variant.assign(impl.parts.size(), std::vector<int>());
    {
        int* dbb = nullptr;
        for (size_t p = 0; p < impl.parts.size(); p++) {
            const PartImpl& pi = impl.parts[p];
            // (scratch parts too, since round 6: swe2d / wave2d are 65 / 15 parts per step, all but four of them scratch parts under
            //  a condition -- without a box each swept its whole grown box on the scalar point kernel, predicate per point, 31 of
            //  swe2d's to write a boundary strip.  Their box is found over the rank box GROWN by the halos of the scratch vars they write,
            //  which is what launch_part() evaluates them over.)
            if (!pi.cond_bb || !pi.meta->has_domain_cond) continue;
            if (!dbb) YKH_HIP(hipMalloc(&dbb, 8 * sizeof(int)));
            const int init[8] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0, 0};
            YKH_HIP(hipMemcpyAsync(dbb, init, sizeof(init), hipMemcpyHostToDevice, compute_stream));
            // Wave-front tiling across ranks evaluates the early phases of a group on boxes grown INTO the neighbours' domains
            // (run_wavefront_multi): the condition's box must cover those points too (the reference finds its bounding boxes over
            // the extended box, ext_bb, setup.cpp:1000-1075) -- clipped to the rank box, a sub-domain part would never be
            // evaluated there and the next phase would read stale values near the rank boundary.
            Box rb = rank_box();
            if (wf_multi()) {
                bool lo[MAX_DOMAIN_DIMS], hi[MAX_DOMAIN_DIMS];
                neighbor_sides(lo, hi);
                for (int d = 0; d < ndd; d++) {
                    if (lo[d]) rb.lo[d] -= wf_ext_[d];
                    if (hi[d]) rb.hi[d] += wf_ext_[d];
                }
            }
            if (pi.meta->is_scratch) rb = scratch_grown_box((int)p, rb);
            PartArgs a;
            fill_part_args((int)p, 0, rb, a);
            a.nxc = 0;
            pi.cond_bb(a, point_grid(rb, a.lane_dim), dbb, compute_stream);
            YKH_HIP(hipGetLastError());
            int out[8];
            YKH_HIP(hipMemcpyAsync(out, dbb, sizeof(out), hipMemcpyDeviceToHost, compute_stream));
            YKH_HIP(hipStreamSynchronize(compute_stream));
            unsigned long long count;
            std::memcpy(&count, &out[6], sizeof(count));      // 64-bit counter in out[6..7]
            Box bb = rb;
            if (count == 0) { for (int d = 0; d < MAX_DOMAIN_DIMS; d++) bb.hi[d] = bb.lo[d]; }      // never true here
            else for (int d = 0; d < 3 && d < MAX_DOMAIN_DIMS; d++) { bb.lo[d] = out[d]; bb.hi[d] = out[3 + d] + 1; }
            part_bb[p] = bb;
            part_has_bb[p] = 1;
            unsigned long long vol = 1;
            for (int d = 0; d < MAX_DOMAIN_DIMS; d++) vol *= (unsigned long long)std::max<idx_t>(0, bb.hi[d] - bb.lo[d]);
            part_bb_solid[p] = (count == vol);     // the condition holds at every point of the box
            // not solid: the list of full boxes, when the region is a handful of them (awp's free-surface planes minus their
            // sponge margins, fsg_abc's 20-point shell = 6 boxes); the fast kernels then run box by box instead of the point
            // kernel sweeping the bounding box with a predicate (fsg_abc 512^3: 30 ms for the stress part's shell)
            if (!part_bb_solid[p] && count > 0 && ndd == 3 && !has_outer && !wf_multi() && pi.cond_profile && !pi.meta->is_scratch) {
                std::vector<Box> boxes;
                if (find_part_boxes((int)p, bb, count, boxes)) part_boxes[p] = std::move(boxes);
            }
            // two domain dims: a condition that is a box minus a box -- the boundary ring around an interior, which is what the
            // complement of an interior condition looks like (swe2d: 31 of its 65 parts define a scratch var on such a ring) -- is found
            // by ONE more reduction: the bounding box and count of the points where the condition does NOT hold.  If that hole is
            // itself solid, the valid points are at most four strips: the part runs its unpredicated kernels strip by strip, and the
            // fused scratch kernel tests two boxes instead of evaluating a 64-bit predicate per point.
            part_hole[p] = Box{{0, 0, 0}, {0, 0, 0}};
            if (!part_bb_solid[p] && count > 0 && ndd <= 2 && !has_outer && !wf_multi()) {      // (one domain dim: the hole leaves two intervals)
                YKH_HIP(hipMemcpyAsync(dbb, init, sizeof(init), hipMemcpyHostToDevice, compute_stream));
                fill_part_args((int)p, 0, bb, a);
                a.nxc = -1;
                pi.cond_bb(a, point_grid(bb, a.lane_dim), dbb, compute_stream);
                YKH_HIP(hipGetLastError());
                YKH_HIP(hipMemcpyAsync(out, dbb, sizeof(out), hipMemcpyDeviceToHost, compute_stream));
                YKH_HIP(hipStreamSynchronize(compute_stream));
                unsigned long long nhole;
                std::memcpy(&nhole, &out[6], sizeof(nhole));
                Box hole = bb;
                for (int d = 0; d < 2; d++) { hole.lo[d] = out[d]; hole.hi[d] = out[3 + d] + 1; }
                std::vector<Box> strips;
                if (ring_strips(bb, (unsigned long long)count, hole, nhole, strips)) {       // (csrc/ykh_boxes.hpp; tests/test_part_boxes_cpu.py)
                    part_hole[p] = hole;
                    part_boxes[p] = std::move(strips);
                }
            }
        }
        if (dbb) YKH_HIP(hipFree(dbb));
    }
    // kernel variants
    bool small_grid = false;
    for (size_t p = 0; p < impl.parts.size(); p++) {
        const PartImpl& pi = impl.parts[p];
        int v = pi.default_variant;
        if (!fast_div && pi.exact_div_variant >= 0) v = pi.exact_div_variant;
        if (force_scalar) v = 0;
        if (!variant_override.empty()) {
            bool found = false;
            for (size_t k = 0; k < pi.variants.size(); k++)
                if (variant_override == pi.variants[k].name) { v = (int)k; found = true; }
            if (!found && impl.parts.size() == 1) {
                std::string names;
                for (auto& kv : pi.variants) names += std::string(" ") + kv.name;
                YKH_THROW("unknown -hip_variant '" + variant_override + "'; available:" + names);
            }
        }
        if (variant_override.empty() && !force_scalar) {
            // a default whose kernel spilled registers into scratch is never worth it (measured 1.4-6x slower):
            // fall back to the next less specialised shape that did not
            while (v > 0 && variant_scratch_bytes(pi.variants[v]) > 0) v--;
        }
        if (variant_override.empty() && !force_scalar && ndd == 3 && pi.large_grid_variant >= 0 &&
            variant_scratch_bytes(pi.variants[pi.large_grid_variant]) == 0) {
            // Large grids: a bigger tile (less halo re-read, longer rows) once its (y,z) tiles alone keep the chip busy
            // (3axis fp64: 128x32 vs 64x32 -- 512^3 0.97x, 768^3 1.05x, 1024^3 1.05x; gpurun_out/r02s)
            const KernelVariant& kv = pi.variants[pi.large_grid_variant];
            if (ceil_div(local_size[2], (idx_t)kv.tz) * ceil_div(local_size[1], (idx_t)kv.ty) * 2 >= std::max(1, env->num_cus))
                v = pi.large_grid_variant;
        }
        if (variant_override.empty() && !force_scalar && ndd == 3 && pi.variants[v].star && pi.variants[v].rx == 0) {
            // Small grids: the default (largest) tile can leave most CUs without a workgroup.  Among the
            // compiled tile shapes of the same kernel family pick the one that fills the most CUs (ties: the
            // larger tile); x-chunks shorter than 32 planes are not worth their halo planes.
            auto blocks_of = [&](const KernelVariant& kv) -> idx_t {
                return ceil_div(local_size[2], (idx_t)kv.tz) * ceil_div(local_size[1], (idx_t)kv.ty) *
                       std::max<idx_t>(1, local_size[0] / 32);
            };
            const idx_t cus = std::max(1, env->num_cus);
            if (blocks_of(pi.variants[v]) < cus) {
                small_grid = true;       // ... and the shapes are timed below; this score is the fallback
                const std::string dn = pi.variants[v].name;
                const std::string family = dn.substr(0, dn.find("_z"));
                // score = share of the CUs that get a workgroup x share of a tile's lanes that do useful work
                auto score_of = [&](const KernelVariant& kv) -> double {
                    double fill = (double)std::min(blocks_of(kv), cus) / (double)cus;
                    double useful = (double)(local_size[2] * local_size[1]) /
                                    (double)(ceil_div(local_size[2], (idx_t)kv.tz) * kv.tz * ceil_div(local_size[1], (idx_t)kv.ty) * kv.ty);
                    return fill * useful;
                };
                double best = score_of(pi.variants[v]);
                idx_t best_area = (idx_t)pi.variants[v].tz * pi.variants[v].ty;
                for (size_t k = 0; k < pi.variants.size(); k++) {
                    const KernelVariant& kv = pi.variants[k];
                    const std::string kn = kv.name;
                    if (!kv.star || kv.rx > 0 || kn.compare(0, 3, "abl") == 0 || kn.compare(0, family.size(), family) != 0) continue;
                    if (kn.find("_pd2") != std::string::npos || kn.find("_cd2") != std::string::npos || kn.find("_hl") != std::string::npos) continue;
                    if (!fast_div && kn.find("_fd") != std::string::npos) continue;
                    if (variant_scratch_bytes(kv) > 0) continue;
                    double sc = score_of(kv);
                    idx_t area = (idx_t)kv.tz * kv.ty;
                    if (sc > best * 1.001 || (sc > best * 0.999 && area > best_area)) { best = sc; best_area = area; v = (int)k; }
                }
            }
        }
        if (part_needs_predicate((int)p)) v = 0;
        part_variant[p] = v;
        part_xchunk[p] = xchunk_override;
    }
    // interior box for comm/compute overlap (alloc.cpp:686-723 `mpi_interior`).  Exterior width per dim = the halo (or
    // -min_exterior).  In z, the unit-stride dim, a slab of 8 points costs as much as one of 64 (every row touches the
    // same 128-byte lines: iso3dfd 512^3, z-face slab of width 8 / 16 / 32 / 64: 0.106 / 0.107 / 0.107 / 0.109 ms,
    // tools/slab_kernels.py) -- so by default the z exterior is one tile of the marching kernel wide and runs on it at
    // full speed, and the interior shrinks by the same points.
    interior_box = rank_box();
    have_interior = false;
    if (env->nranks > 1) {
        bool lo[MAX_DOMAIN_DIMS], hi[MAX_DOMAIN_DIMS];
        for (int d = 0; d < MAX_DOMAIN_DIMS; d++) { lo[d] = d < ndd && rank_index[d] > 0; hi[d] = d < ndd && rank_index[d] < num_ranks[d] - 1; }
        interior_box = interior_for(lo, hi);
        have_interior = !interior_box.empty();
    }
    // pipelined half-exchanges: legal on this rank?  (every written var read face-only, the box long enough in x, something left to
    // overlap with) -- and on every other rank: the halves change what the messages hold, so all ranks take the schedule or none
    halves_geom_ok_ = false;
    halves_in_flight_ = false;
    exch_half_ = -1;
    if (env->nranks > 1) {
        bool ok = halves_geometry(&halves_q1_, &halves_q2_) && have_interior && (num_ranks[1] > 1 || num_ranks[2] > 1);
        for (auto& v : vars) if (v->is_written && v->l1_norm > 1) ok = false;
        halves_geom_ok_ = env->min_over_ranks(ok ? 1 : 0) != 0;
    }
    stats = Stats();
    prepared = true;
    // scratch stages + their consumer as one kernel (ykh_fused.hpp): on where legal unless switched off; the timing pass below
    // (tune_variants) runs a step both ways and keeps the faster when the choice was left open
    fused_on = fused_usable() && fuse_scratch_mode != 0;
    fused_pick_.clear();
    if (placed && placement_trials > 1) tune_placement();
    else { placement_ms.clear(); placement_chosen = 0; }
    if (env->nranks > 1) small_grid = env->max_over_ranks(small_grid ? 1 : 0) != 0;     // (local sizes may differ by rank)
    if (auto_tune) run_auto_tuner_now();
    // Grids too small to give every CU a default tile: which family wins depends on the size (iso3dfd 64^3: point
    // kernel 39 Gpoints/s vs 6.5 for the default marching shape; 256^3: star25d 290 vs 209), so time them once.
    // (-no-auto_tune switches this off too: the static defaults are then reproducible run to run)
    else if (tune_at_prepare && (impl.select_by_timing || small_grid) && variant_override.empty() && !force_scalar) tune_variants(true, all_fresh);
    if (env->exch_reset && env->nranks > 1) env->exch_reset(env->user);      // (var storage may have moved: tune_placement())
    for (auto& h : after_prepare) h(*this);
}

void Solution::end() {
    synchronize();
    drop_step_graphs();
    drop_fused_args();
    drop_launch_plans();
    free_halo_buffers();
    for (auto& v : vars) v->release();
    for (auto& v : scratch_vars) v->release();
    prepared = false;
}

// ------------------------------------------------------------------ full boxes of a sub-domain condition
// The reference's list of FULL bounding boxes of a condition (StencilPartBase::find_bounding_boxes, setup.cpp:1235-1500), found from
// two device reductions instead of a scan of the points: the algorithm is decompose_full_boxes() (ykh_boxes.hpp, host logic, tested
// on the CPU); here its two questions are put to the GPU -- cond_bb_kernel (bounding box + number of valid points of a query box)
// and cond_profile_kernel (valid points per index along x, y, z).
bool Solution::find_part_boxes(int part, const Box& bb0, unsigned long long total, std::vector<Box>& out) {
    const PartImpl& pi = impl.parts[part];
    {
        PartArgs a;
        fill_part_args(part, 0, bb0, a);
        if (a.lane_dim != 2) return false;        // (the profile kernel's histogram layout: 3-D solutions, lanes along z)
    }
    int* dbb = nullptr;
    unsigned* dhist = nullptr;
    YKH_HIP(hipMalloc(&dbb, 8 * sizeof(int)));
    struct Free { int* a; unsigned*& b; ~Free() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } } guard{dbb, dhist};
    size_t hist_cap = 0;
    auto query = [&](const Box& q, Box& bb) -> unsigned long long {
        const int init[8] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0, 0};
        YKH_HIP(hipMemcpyAsync(dbb, init, sizeof(init), hipMemcpyHostToDevice, compute_stream));
        PartArgs a;
        fill_part_args(part, 0, q, a);
        a.nxc = 0;
        pi.cond_bb(a, point_grid(q, a.lane_dim), dbb, compute_stream);
        YKH_HIP(hipGetLastError());
        int o[8];
        YKH_HIP(hipMemcpyAsync(o, dbb, sizeof(o), hipMemcpyDeviceToHost, compute_stream));
        YKH_HIP(hipStreamSynchronize(compute_stream));
        unsigned long long c;
        std::memcpy(&c, &o[6], sizeof(c));
        bb = q;
        if (c) for (int d = 0; d < 3; d++) { bb.lo[d] = o[d]; bb.hi[d] = o[3 + d] + 1; }
        return c;
    };
    auto profile = [&](const Box& bb, std::vector<unsigned>& hist) {
        PartArgs a;
        fill_part_args(part, 0, bb, a);
        const size_t need = hist.size() + 64 + 4;      // (blocks cover 64 z / 4 y: the counts of the overhang are zero)
        if (need > hist_cap) {
            if (dhist) (void)hipFree(dhist);
            dhist = nullptr;
            YKH_HIP(hipMalloc(&dhist, need * sizeof(unsigned)));
            hist_cap = need;
        }
        YKH_HIP(hipMemsetAsync(dhist, 0, need * sizeof(unsigned), compute_stream));
        pi.cond_profile(a, point_grid(bb, a.lane_dim), dhist, compute_stream);
        YKH_HIP(hipGetLastError());
        YKH_HIP(hipMemcpyAsync(hist.data(), dhist, hist.size() * sizeof(unsigned), hipMemcpyDeviceToHost, compute_stream));
        YKH_HIP(hipStreamSynchronize(compute_stream));
    };
    return decompose_full_boxes<Box>(bb0, total, query, profile, out);
}

// ------------------------------------------------------------------ launches
void Solution::fill_part_args(int part, idx_t t, const Box& box, PartArgs& a) const {
    const PartMeta& pm = *impl.parts[part].meta;
    std::memset(&a, 0, sizeof(a));
    if (pm.n_groups > MAX_GROUPS) YKH_THROW("part has too many access groups");
    const Var* full = nullptr;
    for (int g = 0; g < pm.n_groups; g++) {
        const AccessGroup& ag = pm.groups[g];
        const Var* v = nullptr;
        for (auto& vv : vars) if (vv->meta == &meta->vars[ag.var]) v = vv.get();
        for (auto& vv : scratch_vars) if (vv->meta == &meta->vars[ag.var]) v = vv.get();
        if (!v || !v->is_allocated()) YKH_THROW("var used by part '" + std::string(pm.name) + "' has no storage");
        char* base = (char*)v->slot_base(ag.has_step ? t + ag.dt : 0);
        // constant misc indices of this access group
        int mi = 0;
        for (size_t p = 0; p < v->dims.size(); p++)
            if (v->dims[p].is_outer) {        // outer domain dim: the plane this launch works on, plus the group's offset
                const idx_t w = cur_outer + ag.dw;
                if (w < v->dims[p].first_misc || w > v->dims[p].last_misc)
                    YKH_THROW("outer-dim index " + std::to_string(w) + " of var '" + v->name + "' is outside its allocation");
                base += (size_t)((w - v->dims[p].first_misc) * v->misc_stride[p]) * elem_bytes();
            } else if (v->dims[p].type == DIM_MISC) {
                idx_t mv = mi < ag.nmisc ? ag.misc[mi] : v->dims[p].first_misc;
                if (mv < v->dims[p].first_misc || mv > v->dims[p].last_misc)
                    YKH_THROW("misc index " + std::to_string(mv) + " of var '" + v->name + "' is outside its allocation");
                base += (size_t)((mv - v->dims[p].first_misc) * v->misc_stride[p]) * elem_bytes();
                mi++;
            }
        a.ptr[g] = base;
        a.gsx[g] = v->stride[0];
        a.gsy[g] = v->stride[1];
        a.gsz[g] = (int)v->stride[2];
        bool is_full = true;
        for (int d = 0; d < ndd; d++) is_full &= v->uses_domain[d];
        if (is_full && !v->fixed_size && !full) full = v;
    }
    if (full) {      // (solutions with fewer than 3 domain dims: missing dims have stride 0 and extent 1)
        a.sx = full->stride[0];
        a.sy = full->stride[1];
        a.ax0 = (int)-full->pad_l[0]; a.ax1 = (int)(full->dom_size[0] + full->pad_r[0]);
        a.ay0 = (int)-full->pad_l[1]; a.ay1 = (int)(full->dom_size[1] + full->pad_r[1]);
        a.az0 = (int)-full->pad_l[2]; a.az1 = (int)(full->dom_size[2] + full->pad_r[2]);
    }
    // boxes are expressed in (x,y,z) = domain_idx (0,1,2); solutions with fewer dims use size-1 boxes
    a.x0 = (int)box.lo[0]; a.x1 = (int)box.hi[0];
    a.y0 = (int)box.lo[1]; a.y1 = (int)box.hi[1];
    a.z0 = (int)box.lo[2]; a.z1 = (int)box.hi[2];
    a.ofs_x = (int)rank_ofs[0]; a.ofs_y = (int)rank_ofs[1]; a.ofs_z = (int)rank_ofs[2];
    a.glast_x = (int)(ndd > 0 ? global_size[0] - 1 : 0);
    a.glast_y = (int)(ndd > 1 ? global_size[1] - 1 : 0);
    a.glast_z = (int)(ndd > 2 ? global_size[2] - 1 : 0);
    a.t = t;
    a.dom_x1 = (int)(ndd > 0 ? local_size[0] : 1);
    a.dom_y1 = (int)(ndd > 1 ? local_size[1] : 1);
    a.dom_z1 = (int)(ndd > 2 ? local_size[2] : 1);
    a.lane_dim = std::max(0, std::min(2, ndd - 1));
    // (experiment knob of -DYKH_PROFILING kernel builds, profiles/r6_iso3dfd_fetch; shipped kernels never read it)
    static const int xcd_map_env = [] { const char* e = getenv("YASK_HIP_XCD_MAP"); return e ? atoi(e) : 0; }();
    a.xcd_map = xcd_map_env;
}

// workgroups of a kernel shape that one CU holds at a time (registers, LDS, waves), asked of the runtime once per shape
int Solution::resident_blocks(const KernelVariant& kv) {
    // (keyed by kernel AND launch shape; a failing query is not remembered -- ADVICE r05: the question used to be asked before the
    //  shape's first launch had raised the kernel's dynamic-LDS limit, answered with an error, and the lock-step shape then ran as
    //  its plain sibling for the life of the Solution, without a trace)
    const auto key = std::make_tuple(kv.func, kv.threads, kv.lds_bytes);
    auto it = resident_cache_.find(key);
    if (it != resident_cache_.end()) return it->second;
    int n = 0;
    if (kv.func) {
        if (kv.lds_bytes > 48 * 1024)     // what the shape's launch() does on first use (ykh_stencil_tu.hpp)
            (void)hipFuncSetAttribute(kv.func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kv.lds_bytes);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kv.func, kv.threads, kv.lds_bytes) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    }
    if (n > 0) resident_cache_[key] = n;
    else if (env->trace) fprintf(stderr, "yask-hip: occupancy query of kernel shape '%s' failed: a lock-step shape runs as its plain sibling\n", kv.name);
    return n;
}

void Solution::launch_part_variant(int part, int variant, idx_t xchunk, idx_t t, const Box& box_in, hipStream_t s) {
    // a sub-domain part with a list of full boxes: one launch per box that meets the request (every kernel family is legal there)
    if (!in_part_boxes_ && (size_t)part < part_boxes.size() && !part_boxes[part].empty()) {
        ScopedSet<bool> walking(in_part_boxes_, true);
        const bool per_box = (size_t)part < part_box_variant.size() && part_box_variant[part].size() == part_boxes[part].size() &&
                             variant == part_variant[part];
        for (size_t bi = 0; bi < part_boxes[part].size(); bi++) {
            const Box& fb = part_boxes[part][bi];
            Box b = box_in;
            for (int d = 0; d < MAX_DOMAIN_DIMS; d++) { b.lo[d] = std::max(b.lo[d], fb.lo[d]); b.hi[d] = std::min(b.hi[d], fb.hi[d]); }
            if (!b.empty()) launch_part_variant(part, per_box ? part_box_variant[part][bi] : variant, per_box ? 0 : xchunk, t, b, s);
        }
        return;
    }
    // sub-domain parts run inside the bounding box of their condition only; where the condition does not hold
    // everywhere in that box, only the point kernel (which evaluates it per point) is legal
    Box box = box_in;
    if ((size_t)part < part_has_bb.size() && part_has_bb[part]) {
        for (int d = 0; d < MAX_DOMAIN_DIMS; d++) {
            box.lo[d] = std::max(box.lo[d], part_bb[part].lo[d]);
            box.hi[d] = std::min(box.hi[d], part_bb[part].hi[d]);
        }
    }
    if (part_needs_predicate(part)) variant = 0;
    if (box.empty()) return;
    const KernelVariant& kv = impl.parts[part].variants[variant];
    PartArgs a;
    fill_part_args(part, t, box, a);
    if (kv.lift2d) {
        // a 2-D part on a 3-D kernel family (ykh_lift2d.hpp): (d0, d1) -> (y, z) of ONE x plane.  Everything that is indexed by
        // domain dim moves up one place; base pointers stay (they address local element (0, 0) either way).
        for (int g = 0; g < impl.parts[part].meta->n_groups; g++) { a.gsz[g] = (int)a.gsy[g]; a.gsy[g] = a.gsx[g]; a.gsx[g] = 0; }
        a.sy = a.sx; a.sx = 0;
        a.z0 = a.y0; a.z1 = a.y1; a.y0 = a.x0; a.y1 = a.x1; a.x0 = 0; a.x1 = 1;
        a.az0 = a.ay0; a.az1 = a.ay1; a.ay0 = a.ax0; a.ay1 = a.ax1; a.ax0 = 0; a.ax1 = 1;
        a.ofs_z = a.ofs_y; a.ofs_y = a.ofs_x; a.ofs_x = 0;
        a.glast_z = a.glast_y; a.glast_y = a.glast_x; a.glast_x = 0;
        a.dom_z1 = a.dom_y1; a.dom_y1 = a.dom_x1; a.dom_x1 = 1;
        a.lane_dim = 2;
        box.lo[2] = box.lo[1]; box.hi[2] = box.hi[1]; box.lo[1] = box.lo[0]; box.hi[1] = box.hi[0]; box.lo[0] = 0; box.hi[0] = 1;
    }
    if (kv.lift1d) {
        // a 1-D part on a 3-D kernel family: d0 -> z of ONE row of ONE plane (everything indexed by domain dim moves up two places)
        for (int g = 0; g < impl.parts[part].meta->n_groups; g++) { a.gsz[g] = (int)a.gsx[g]; a.gsy[g] = 0; a.gsx[g] = 0; }
        a.sy = 0; a.sx = 0;
        a.z0 = a.x0; a.z1 = a.x1; a.y0 = 0; a.y1 = 1; a.x0 = 0; a.x1 = 1;
        a.az0 = a.ax0; a.az1 = a.ax1; a.ay0 = 0; a.ay1 = 1; a.ax0 = 0; a.ax1 = 1;
        a.ofs_z = a.ofs_x; a.ofs_y = 0; a.ofs_x = 0;
        a.glast_z = a.glast_x; a.glast_y = 0; a.glast_x = 0;
        a.dom_z1 = a.dom_x1; a.dom_y1 = 1; a.dom_x1 = 1;
        a.lane_dim = 2;
        box.lo[2] = box.lo[0]; box.hi[2] = box.hi[0]; box.lo[1] = 0; box.hi[1] = 1; box.lo[0] = 0; box.hi[0] = 1;
    }
    if (kv.star) {
        const int vz = kv.vz > 0 ? kv.vz : 16 / elem_bytes();
        idx_t zt0 = box.lo[2] & ~(idx_t)(vz - 1);
        a.ntz = (int)ceil_div(box.hi[2] - zt0, kv.tz);
        a.nty = (int)ceil_div(box.hi[1] - box.lo[1], kv.ty);
        idx_t nx = box.hi[0] - box.lo[0];
        if (kv.rx > 0) {
            // point kernel: grid = z tiles * y tiles * x blocks
            a.nxc = (int)ceil_div(nx, kv.rx);
            a.xchunk = kv.rx;
            dim3 grid((unsigned)((idx_t)a.ntz * a.nty * a.nxc), 1, 1);
            kv.launch(a, grid, s);
            YKH_HIP(hipGetLastError());
            return;
        }
        idx_t xc = xchunk;
        if (xc <= 0) {
            // default x-chunking: the (y,z) tiles times the number of x-chunks should fill the CUs in whole
            // rounds (a 576-tile plane on 256 CUs would otherwise run 3 rounds for 2.25 rounds of work),
            // while every chunk re-loads the x-halo planes of its neighbours (cost ~ xhalo / chunk length).
            const idx_t tiles = (idx_t)a.ntz * a.nty;
            const idx_t cus = std::max<idx_t>(1, env->num_cus);
            idx_t xhalo = shared_pad_l_[0] + shared_pad_r_[0];
            double best_eff = -1;
            idx_t best_n = 1;
            for (idx_t n = 1; n <= 32; n++) {
                idx_t len = ceil_div(nx, n);
                if (n > 1 && len < 32) break;
                idx_t blocks = tiles * ceil_div(nx, len);
                double fill = (double)blocks / (double)(ceil_div(blocks, cus) * cus);
                double eff = fill * (double)len / (double)(len + xhalo);
                if (eff > best_eff * 1.02) { best_eff = eff; best_n = n; }   // prefer fewer chunks on near-ties
            }
            xc = ceil_div(nx, best_n);
        }
        xc = std::max<idx_t>(1, std::min(xc, nx));
        a.xchunk = (int)xc;
        a.nxc = (int)ceil_div(nx, xc);
        {
            // More tiles than CUs: one launch per round of whole tile rows, so that the tiles of a round start
            // together and march in step (neighbouring tiles then touch the same DRAM pages and L2 lines at
            // about the same time; tiles of a second round started one by one drift apart).
            const idx_t per_row = (idx_t)a.ntz * a.nxc, cus = std::max(1, env->num_cus);
            const idx_t rows = cus / per_row;
            // (only when that does not add rounds: 100 tiles per row would give 200-block launches)
            if (per_row * a.nty > cus && rows >= 1 && ceil_div((idx_t)a.nty, rows) <= ceil_div(per_row * a.nty, cus)) {
                for (idx_t y0 = box.lo[1]; y0 < box.hi[1]; y0 += rows * kv.ty) {
                    Box sub = box;
                    sub.lo[1] = y0;
                    sub.hi[1] = std::min(y0 + rows * kv.ty, box.hi[1]);
                    if (kv.lift2d) {          // (the callee lifts again: hand it the sub-box in the solution's own two dims)
                        sub.lo[0] = sub.lo[1]; sub.hi[0] = sub.hi[1]; sub.lo[1] = sub.lo[2]; sub.hi[1] = sub.hi[2]; sub.lo[2] = 0; sub.hi[2] = 1;
                    }
                    if (kv.lift1d) { sub.lo[0] = sub.lo[2]; sub.hi[0] = sub.hi[2]; sub.lo[1] = 0; sub.hi[1] = 1; sub.lo[2] = 0; sub.hi[2] = 1; }
                    launch_part_variant(part, variant, xc, t, sub, s);
                }
                return;
            }
        }
        dim3 grid((unsigned)((idx_t)a.ntz * a.nty * a.nxc), 1, 1);
        // "_ls<K>" shapes: the workgroups of an XCD keep within K planes of each other (starlin_kernel, xcd_sync).  Only where the
        // hand-shake can complete: block i on XCD i % 8 with equal shares (grid a multiple of 8), every block of the launch resident at
        // once, and every block with the same number of planes to march (equal x-chunks) -- else the shape runs as its plain sibling.
        if (kv.lockstep) {
            if ((grid.x & 7) == 0 && nx % a.nxc == 0 && (idx_t)a.xchunk * a.nxc == nx &&
                (idx_t)grid.x <= (idx_t)resident_blocks(kv) * std::max(1, env->num_cus)) {
                if (!lockstep_dev) YKH_HIP(hipMalloc(&lockstep_dev, 8 * 32 * sizeof(unsigned)));
                YKH_HIP(hipMemsetAsync(lockstep_dev, 0, 8 * 32 * sizeof(unsigned), s));
                a.sig = lockstep_dev;
            } else if (env->trace)
                fprintf(stderr, "yask-hip: lock-step shape '%s' runs without its hand-shake on this launch (grid %u, %lld x-chunks of %d over %lld planes)\n",
                        kv.name, grid.x, (long long)a.nxc, a.xchunk, (long long)nx);
        }
        kv.launch(a, grid, s);
        if (a.sig && kv.lockstep && env->trace) {
            // -trace: did any workgroup give up waiting for its XCD (400 polls)?  Such a launch is slower AND its A/B numbers are polluted
            unsigned h[8 * 32];
            YKH_HIP(hipMemcpyAsync(h, lockstep_dev, sizeof(h), hipMemcpyDeviceToHost, s));
            YKH_HIP(hipStreamSynchronize(s));
            unsigned gave_up = 0;
            for (int i = 0; i < 8; i++) gave_up += h[i * 32 + 1];
            if (gave_up) fprintf(stderr, "yask-hip: lock-step of '%s': %u of %u workgroups timed out waiting for their XCD\n", kv.name, gave_up, grid.x);
        }
    } else {
        // a box that is (nearly) a plane of constant z -- awp's free-surface parts: 512 x 512 x 1 -- would keep ONE lane of each wave
        // busy with the lanes along z (0.4-0.6 ms for 262 144 points): there the lanes run along y.  Every lane then touches a cache
        // line of its own, but 64 of them do.
        if (a.lane_dim == 2 && ndd == 3 && box.hi[2] - box.lo[2] <= 8 && box.hi[1] - box.lo[1] >= 32) a.lane_dim = 1;
        kv.launch(a, point_grid(box, a.lane_dim), s);
    }
    YKH_HIP(hipGetLastError());
}

void Solution::launch_part(int part, idx_t t, const Box& box_in, hipStream_t s) {
    if (has_outer && !in_outer_loop) {
        // 4 domain dims: the kernels sweep (x, y, z); the outermost dim is a loop of launches, every access group's base
        // pointer following it (fill_part_args).  Planes of one step are independent (reads and writes are in different
        // step slots).
        ScopedSet<bool> in_loop(in_outer_loop, true);
        ScopedSet<idx_t> outer(cur_outer, 0);
        for (cur_outer = 0; cur_outer < local_size[3]; cur_outer++) launch_part(part, t, box_in, s);
        return;
    }
    const PartMeta& pm = *impl.parts[part].meta;
    if (pm.step_cond && !pm.step_cond(t)) return;            // IF_STEP: the part is idle this step
    if (!pm.is_scratch) {
        int v = part_variant[part];
        const KernelVariant& kv = impl.parts[part].variants[v];
        Box box = box_in;                     // (sub-domain parts: what launch_part_variant will really cover)
        if ((size_t)part < part_has_bb.size() && part_has_bb[part])
            for (int d = 0; d < MAX_DOMAIN_DIMS; d++) {
                box.lo[d] = std::max(box.lo[d], part_bb[part].lo[d]);
                box.hi[d] = std::min(box.hi[d], part_bb[part].hi[d]);
            }
        // Thin exterior slabs of a y/z decomposition (8 points wide against a 128 x 32 tile) would keep 1/16 of a
        // marching tile's lanes busy: such boxes go to the point kernel (always variant 0).
        // The same for thin x slabs (x-face exteriors): a marching tile runs a 16-plane prologue for 8 planes of output
        // (iso3dfd 512^2 x 8: default marching shape 0.040 ms, point kernel 0.019 ms).
        // (only the exterior slabs of a decomposed run: slabs of the wave-front schedule keep the part's kernel)
        // (... where a plane of tiles leaves most CUs idle; with 256 tiles -- a 1024^2 face -- the marching kernel's 24
        // plane-iterations, 0.07 ms, beat the point kernel's 0.10 ms)
        if (launching_exterior && thin_slab_point_kernel && kv.star && kv.rx == 0 && ndd == 3 && !box.empty()) {
            const idx_t tiles = ceil_div(box.hi[2] - box.lo[2], (idx_t)kv.tz) * ceil_div(box.hi[1] - box.lo[1], (idx_t)kv.ty);
            if ((box.hi[2] - box.lo[2]) * 4 <= kv.tz || (box.hi[1] - box.lo[1]) * 4 <= kv.ty ||
                ((box.hi[0] - box.lo[0]) <= shared_pad_l_[0] + shared_pad_r_[0] && tiles * 2 <= std::max(1, env->num_cus)))
                v = 0;
        }
        launch_part_variant(part, v, part_xchunk[part], t, box, s);
        return;
    }
    // scratch part: evaluate over the box grown by the halo of the scratch var(s) it writes, so that the
    // parts reading them at offsets find every value (the reference does this per micro-block,
    // src/kernel/lib/stencil_calc.cpp:40-289; here the scratch var is a whole device array)
    launch_part_variant(part, part_variant[part], part_xchunk[part], t, scratch_grown_box(part, box_in), s);
}
// the box a scratch part is evaluated over when its consumers run over `box`: grown by the halos of the scratch var(s) it writes
Box Solution::scratch_grown_box(int part, const Box& box) const {
    const PartMeta& pm = *impl.parts[part].meta;
    Box b = box;
    for (int w = 0; w < pm.n_writes; w++) {
        const AccessGroup& ag = pm.groups[pm.writes[w]];
        for (auto& v : scratch_vars)
            if (v->meta == &meta->vars[ag.var])
                for (int d = 0; d < ndd; d++)
                    if (v->uses_domain[d]) {
                        b.lo[d] = std::min(b.lo[d], box.lo[d] - v->halo_l[d]);
                        b.hi[d] = std::max(b.hi[d], box.hi[d] + v->halo_r[d]);
                    }
    }
    return b;
}

// ------------------------------------------------------------------ run
// Coherency of the host copies behind get_raw_storage_buffer() around a call that uses and changes var data (ykh_var.cpp).
struct RawStorageGuard {
    Solution& s;
    explicit RawStorageGuard(Solution& s_) : s(s_) { for (auto& v : s.vars) v->before_device_use(); }
    ~RawStorageGuard() {
        try { for (auto& v : s.vars) v->after_device_write(); } catch (...) {}
    }
};

// bookkeeping: written vars become valid at the output step and dirty for neighbours (yk_var.cpp:122-152,559-575)
void Solution::note_stage_written(const StageMeta& sm, idx_t t) {
    for (int k = 0; k < sm.n_parts; k++) {
        const PartMeta& pm = *impl.parts[sm.parts[k]].meta;
        if (pm.is_scratch || (pm.step_cond && !pm.step_cond(t))) continue;
        for (int w = 0; w < pm.n_writes; w++) {
            const AccessGroup& ag = pm.groups[pm.writes[w]];
            for (auto& v : vars)
                if (v->meta == &meta->vars[ag.var]) {
                    if (ag.has_step) { v->update_valid_step(t + ag.dt); v->set_dirty(true, t + ag.dt); }
                    else v->set_dirty_all(true);
                }
        }
    }
}
void Solution::note_step_written(idx_t t) {
    for (int st = 0; st < meta->n_stages; st++) note_stage_written(meta->stages[st], t);
}

void Solution::run(idx_t first_step, idx_t last_step) {
    for (auto& h : before_run) h(*this, first_step, last_step);
    if (!prepared) YKH_THROW("run_solution() called without calling prepare_solution() first");
    RawStorageGuard raw_guard(*this);     // vars whose raw buffer has been handed out: host copy in, host copy out
    // Direction comes from the order of the indices only; each step index is evaluated as the stencil defines it
    // (context.cpp:236-246: a single index is one step whatever the stencil's own direction).
    const idx_t dir = (last_step >= first_step) ? 1 : -1;
    auto t0 = std::chrono::steady_clock::now();
    const bool multi = env->nranks > 1 && do_halo_exchange && !neighbors.empty();
    double halo_secs = 0;
    if (multi) {
        // other ranks may have changed data through the API: treat every var as possibly dirty
        // (set_all_neighbor_vars_dirty, context.cpp:234) so that all ranks agree on message sizes
        auto h0 = std::chrono::steady_clock::now();
        exchange_halos_all();   // initial exchange of everything marked dirty (context.cpp:346)
        halo_secs += std::chrono::duration<double>(std::chrono::steady_clock::now() - h0).count();
    }
    const Box rb = rank_box();
    idx_t nsteps = 0;
    const idx_t nsteps_total = (dir > 0 ? last_step - first_step : first_step - last_step) + 1;
    if (step_timers) {
        while ((idx_t)step_events.size() < nsteps_total + 1) {
            hipEvent_t e;
            YKH_HIP(hipEventCreate(&e));
            step_events.push_back(e);
        }
        YKH_HIP(hipEventRecord(step_events[0], compute_stream));
    }
    phase_used = 0;
    cur_phase = nullptr;
    shell_event_pending = false;
    // Wave-front temporal tiling (-Mbt / -bt > 1): groups of steps go slab by slab (run_wavefront below).  Single rank only:
    // with neighbours the halos would have to be wf_steps x wider (the reference extends them, setup.cpp:863-1020).
    const idx_t wf_steps = std::max<idx_t>(mega_block_size[0], block_size[0]);
    idx_t first_plain = first_step;           // steps before this one were done two at a time (run_fused)
    if (can_fuse() && !multi) {
        const idx_t total = (dir > 0 ? last_step - first_step : first_step - last_step) + 1, npairs = total / 2;
        if (npairs > 0) {
            run_fused(first_step, npairs, dir);
            for (idx_t k = 0; k < 2 * npairs; k++) {
                nsteps++;
                if (step_timers) YKH_HIP(hipEventRecord(step_events[nsteps], compute_stream));
            }
            first_plain = first_step + dir * 2 * npairs;
        }
    }
    if (wf_steps > 1 && multi && wf_multi()) {
        // groups of wf_steps steps, one halo exchange per group (run_wavefront_multi above)
        bool lo[MAX_DOMAIN_DIMS], hi[MAX_DOMAIN_DIMS];
        neighbor_sides(lo, hi);
        for (idx_t t = first_step; dir > 0 ? t <= last_step : t >= last_step;) {
            const idx_t left = (dir > 0 ? last_step - t : t - last_step) + 1, g = std::min(wf_steps, left);
            run_wavefront_multi(t, g, dir, lo, hi, /*exchange=*/true);
            for (idx_t k = 0; k < g; k++) {
                nsteps++;
                if (step_timers) YKH_HIP(hipEventRecord(step_events[nsteps], compute_stream));
            }
            t += dir * g;
        }
        first_plain = last_step + dir;         // nothing left for the plain loop below
    }
    const bool wavefront = wf_steps > 1 && !multi && env->nranks == 1 && ndd >= 1;
    for (idx_t t = first_plain; wavefront && (dir > 0 ? t <= last_step : t >= last_step);) {
        const idx_t left = (dir > 0 ? last_step - t : t - last_step) + 1, g = std::min(wf_steps, left);
        run_wavefront(t, g, dir);
        for (idx_t k = 0; k < g; k++) {
            nsteps++;
            if (step_timers) YKH_HIP(hipEventRecord(step_events[nsteps], compute_stream));   // (all g steps end together)
        }
        t += dir * g;
    }
    if (fused_on) ensure_fused_args();                    // (device tables of the fused scratch groups: never built inside a capture)
    // Launch-bound runs: the launches of a whole number of slot periods are captured once and replayed (step graphs, below).
    idx_t t_plain = first_plain;
    if (!wavefront && !multi && !step_timers && step_graph_wanted()) {
        idx_t left = (dir > 0 ? last_step - t_plain : t_plain - last_step) + 1, done = 0;
        const idx_t P = slot_period();
        // steps per graph: whole periods, at most ~256 kernel nodes (the chain is linear: more nodes only cost host memory)
        idx_t per_step = 0;
        for (int st = 0; st < meta->n_stages; st++) per_step += meta->stages[st].n_parts;
        if (has_outer) per_step *= std::max<idx_t>(1, local_size[3]);
        const idx_t cap = std::max<idx_t>(P, (256 / std::max<idx_t>(1, per_step)) / P * P);
        while (left >= 2 * P && left >= 2) {        // (what is left in the end, fewer than two periods, is issued as plain launches)
            const idx_t G = std::min(cap, left / P * P);
            StepGraph* sg = get_step_graph(t_plain, dir, G);
            if (!sg) break;
            YKH_HIP(hipGraphLaunch(sg->exec, compute_stream));
            stats.graph_replays++;
            stats.graph_steps += G;
            nsteps += G;
            done += G;
            left -= G;
            t_plain += dir * G;
        }
        // bookkeeping of the replayed steps: what the plain loop does per step; the last two periods decide the final state
        for (idx_t k = std::min<idx_t>(done, 2 * P); k >= 1; k--) note_step_written(t_plain - dir * k);
    }
    const bool halves_on = multi && halves_active();
    halves_in_flight_ = false;
    for (idx_t t = t_plain; !wavefront && (dir > 0 ? t <= last_step : t >= last_step); t += dir) {
        for (int st = 0; st < meta->n_stages; st++) {
            if (fused_on) {
                const FusedGroupImpl* fg = fused_group_at(st);
                if (fg && fused_ok_at(*fg, t)) {
                    // a run of scratch stages and the stage they feed: ONE launch, scratch vars in the LDS (ykh_fused.hpp).  A decomposed
                    // rank runs it over its whole box and exchanges afterwards (the scratch vars themselves never travel: every rank
                    // computes them on its box grown by their halos, from inputs whose halos the previous exchange filled)
                    cur_phase = multi ? phase_next() : nullptr;
                    phase_mark(PH_EXT1, compute_stream);
                    launch_fused(*fg, t, compute_stream);
                    note_stage_written(meta->stages[fg->last_stage], t);
                    if (multi) {
                        exchange_halos(t, fg->last_stage, /*start_only=*/true, false);
                        phase_mark(PH_INT1, compute_stream);
                        exchange_halos(t, fg->last_stage, false, /*finish_only=*/true);
                        phase_mark(PH_WAIT1, compute_stream);
                    }
                    cur_phase = nullptr;
                    st = fg->last_stage;
                    continue;
                }
            }
            const StageMeta& sm = meta->stages[st];
            bool lo[MAX_DOMAIN_DIMS], hi[MAX_DOMAIN_DIMS];
            neighbor_sides(lo, hi);
            if (halves_on) {
                // two launches in regular order, each followed by the exchange of its part of the faces (run_stage_halves)
                run_stage_halves(sm, st, t, lo, hi);
                continue;
            }
            const bool overlap = multi && overlap_comms && have_interior;
            cur_phase = multi ? phase_next() : nullptr;
            // (x-only decompositions keep the slab schedule: their faces are whole planes, the two thin slabs cost 2-14 % where
            //  cutting every tile into rounds costs 4-40 %, tools/decomp_cost.py)
            const int pl_part = (overlap && (lo[1] || hi[1] || lo[2] || hi[2])) ? planned_part(sm) : -1;
            if (pl_part >= 0) {
                // ONE launch over the rank box: shell blocks first, the exchange released from the device when they are done
                // (the reference's exterior-first order, context.cpp:377-478, without separate launches)
                LaunchPlan* lp = get_launch_plan(pl_part, lo, hi);
                phase_mark(PH_EXT0, compute_stream);
                note_stage_written(sm, t);
                launch_planned(pl_part, t, *lp, /*signal=*/true, compute_stream);
                exchange_halos(t, st, /*start_only=*/true, false);       // (marks PH_EXT1 on the comm stream, after its wait)
                phase_mark(PH_INT1, compute_stream);
                exchange_halos(t, st, false, /*finish_only=*/true);
                phase_mark(PH_WAIT1, compute_stream);
                cur_phase = nullptr;
                continue;
            }
            if (overlap) {
                // exterior slabs first (context.cpp:377-444), then start the exchange, then the interior
                phase_mark(PH_EXT0, compute_stream);
                launch_exterior(sm, t, interior_box);
                phase_mark(PH_EXT1, compute_stream);
            } else {
                phase_mark(PH_EXT1, compute_stream);      // (no split: the whole box counts as interior time)
                for (int k = 0; k < sm.n_parts; k++) launch_part(sm.parts[k], t, rb, compute_stream);
            }
            note_stage_written(sm, t);
            if (multi) {
                exchange_halos(t, st, /*start_only=*/true, false);
                if (overlap) launch_interior(sm, t, interior_box);
                phase_mark(PH_INT1, compute_stream);
                exchange_halos(t, st, false, /*finish_only=*/true);
                phase_mark(PH_WAIT1, compute_stream);     // completes when the halos have landed (stream waits on ev_b)
            }
            cur_phase = nullptr;
        }
        nsteps++;
        if (step_timers) YKH_HIP(hipEventRecord(step_events[nsteps], compute_stream));
    }
    if (halves_in_flight_) {
        // the last half-exchange of the run: nothing left to hide it behind
        cur_phase = phase_next();
        phase_mark(PH_INT1, compute_stream);
        halves_finish();
        phase_mark(PH_WAIT1, compute_stream);
        cur_phase = nullptr;
    }
    YKH_HIP(hipStreamSynchronize(compute_stream));
    if (multi) YKH_HIP(hipStreamSynchronize(comm_stream));
    check_async_errors("run_solution()");
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stats.elapsed_secs += secs;
    stats.halo_secs += halo_secs;
    stats.num_steps_done += nsteps;
    steps_done_total += nsteps;
    if (multi) phase_collect();
    if (step_timers) {
        step_ms.assign((size_t)nsteps, 0.f);
        for (idx_t i = 0; i < nsteps; i++)
            if (hipEventElapsedTime(&step_ms[i], step_events[i], step_events[i + 1]) != hipSuccess) { (void)hipGetLastError(); step_ms[i] = 0.f; }
    }
    for (auto& h : after_run) h(*this, first_step, last_step);
}

// ------------------------------------------------------------------ stats (soln_apis.cpp:349-562)
Stats Solution::get_stats() {
    Stats s = stats;
    idx_t pts = 1;
    for (int d = 0; d < ndd; d++) pts *= global_size[d];
    if (has_outer) pts *= global_size[3];
    s.num_elements = pts;
    idx_t reads = 0, writes = 0, fpops = 0;
    for (auto& p : impl.parts) {
        if (p.meta->is_scratch) continue;      // the reference's work stats cover non-scratch parts
        reads += p.meta->points_read; writes += p.meta->points_written; fpops += p.meta->fp_ops;
    }
    s.num_writes_done = writes * pts * s.num_steps_done;
    s.num_reads_done = reads * pts * s.num_steps_done;
    s.est_fp_ops_done = fpops * pts * s.num_steps_done;
    s.pts_per_sec = s.elapsed_secs > 0 ? (double)pts * (double)s.num_steps_done / s.elapsed_secs : 0.0;
    stats = Stats();   // cleared on read, like the reference
    return s;
}

}  // namespace ykh
