// ykh_runtime.hpp -- host side of the MI355X stencil runtime ("ykh" = YASK kernel, HIP).
//
// Mirrors, for the GPU, the pieces of the reference's StencilContext that sit on the
// run_solution() hot path:
//   Var       <- YkVarBase/YkVarImpl + GenericVar   (src/kernel/lib/yk_var.{hpp,cpp}, generic_var.*)
//   Solution  <- StencilContext + KernelSettings     (src/kernel/lib/context.*, settings.*, setup.cpp,
//                                                      soln_apis.cpp, alloc.cpp, halo.cpp, auto_tuner.*)
//   Env       <- KernelEnv                           (src/kernel/lib/settings.hpp:60-140, setup.cpp:38-137)
// Storage is device-resident, plain row-major [step][misc..][x][y][z] with z unit-stride (no vector
// folding: the 64 lanes of a wavefront span z).  The public faces are the C ABI in
// include/yask_hip_c_api.h and, on top of it, the C++ yk_* classes (yask_amd/cxxapi) and the Python
// package yask_amd.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "ykh_device.hpp"
#include "ykh_meta.hpp"
#include "ykh_plan.hpp"

namespace ykh {

// Same message prefix as yask::yask_exception (include/yask_common_api.hpp:125-179).
struct Error : public std::runtime_error {
    explicit Error(const std::string& m) : std::runtime_error("YASK error: " + m) {}
};
#define YKH_THROW(msg) throw ::ykh::Error(std::string(msg))
#define YKH_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            throw ::ykh::Error(std::string(#call) + " failed: " + hipGetErrorString(e_));          \
    } while (0)

// ------------------------------------------------------------------ kernel registry (per stencil lib)
struct KernelVariant {
    const char* name;      // e.g. "star25d_z32_y8_r1_u"
    bool star;             // false: naive
    int tz, ty;            // tile extent in elements (star only)
    size_t lds_bytes;
    int threads;
    void (*launch)(const PartArgs& a, dim3 grid, hipStream_t s);
    int vz = 0;            // elements per thread along z (0: one 16-byte vector)
    int rx = 0;            // >0: not a marching kernel; a block handles rx consecutive x planes (vecpt)
    const void* func = nullptr;   // the __global__ symbol (for hipFuncGetAttributes: scratch use = register spills)
    // planned launches (ykh_plan.cpp): the shape's twin that reads per-workgroup BlockDescs (PartArgs::blk) and signals
    void (*launch_desc)(const PartArgs& a, dim3 grid, hipStream_t s) = nullptr;
    const void* func_desc = nullptr;
    int xover = 0;                // plane-iterations of overhead per block (prologue; the planner's cost model)
    bool lockstep = false;        // "_ls<K>" shapes: PartArgs::sig = 8 zeroed per-XCD counters, see starlin_kernel
    // cluster variants ("c<K>_*": K kernels behind one launch function): the kernels of clusters 1 .. K-1 (func is cluster 0's)
    const void* more_funcs[7] = {};
    int n_more_funcs = 0;
    // a 3-D kernel family instantiated on Lift2D<part> (ykh_lift2d.hpp): the part has two domain dims (d0, d1), the kernel sees them
    // as (y, z) of one x plane; Solution::launch_part_variant() moves the box and the PartArgs up one dim before the launch
    bool lift2d = false;
    bool lift1d = false;          // ... one domain dim d0, seen as z of one row of one plane
};
// bytes of scratch (private segment) per thread of a variant's kernel(s) -- the maximum over the kernels of a cluster variant;
// > 0 means hipcc spilled registers
size_t variant_scratch_bytes(const KernelVariant& kv);
// Two-steps-per-pass kernel of a part (ykh_starlin2.hpp), when the part has one: not a variant of the one-step
// kernels (it reads ptr[0] = S(t), the pads of ptr[1] = the t+1 slot, and writes S(t+2) to ptr[2]).
struct Fused2Variant {
    const char* name = nullptr;
    int tz = 0, ty = 0;        // outer tile (threads cover it)
    int tzi = 0, tyi = 0;      // inner tile = tile stride
    int vz = 0, xr = 0;        // elements per thread along z; stencil radius along x (extra planes per x-chunk = 4 * xr + 1)
    size_t lds_bytes = 0;
    int threads = 0;
    void (*launch)(const PartArgs& a, dim3 grid, hipStream_t s, bool store_b) = nullptr;
    const void* func = nullptr;
};
struct PartImpl {
    const PartMeta* meta;
    Fused2Variant fused2;
    std::vector<KernelVariant> variants;   // variants[0] is the always-legal naive kernel
    int default_variant = 0;
    int large_grid_variant = -1;       // preferred over the default when its (y,z) tiles alone cover at least half of the CUs
    // sub-domain (IF_DOMAIN) parts: launches cond_bb_kernel over the box of `a` (grid as for the naive kernel)
    void (*cond_bb)(const PartArgs& a, dim3 grid, int* dev_out, hipStream_t s) = nullptr;
    // ... and cond_profile_kernel: per-index counts along x, y, z of the box (3-D solutions)
    void (*cond_profile)(const PartArgs& a, dim3 grid, unsigned* dev_hist, hipStream_t s) = nullptr;
    void set_default(const char* name) {
        for (size_t i = 0; i < variants.size(); i++)
            if (std::string(variants[i].name) == name) { default_variant = (int)i; return; }
        throw std::runtime_error(std::string("no kernel variant named ") + name);
    }
    // Shape whose twin runs the planned launches of a decomposed rank when the part runs on its static default (and, with
    // -no-hip_fast_div, on the exact-division default): it must have the same arithmetic as that default (queue renaming, operand
    // refill and halo rings change none) -- e.g. ssg stage 2's default sits at 256 VGPRs and its twin would spill.
    int planned_variant = -1, planned_exact_variant = -1;
    void set_planned(const char* name, const char* exact_name = nullptr) {
        for (size_t i = 0; i < variants.size(); i++) {
            if (std::string(variants[i].name) == name) planned_variant = (int)i;
            if (exact_name && std::string(variants[i].name) == exact_name) planned_exact_variant = (int)i;
        }
        if (planned_variant < 0 || (exact_name && planned_exact_variant < 0)) throw std::runtime_error(std::string("no kernel variant named ") + name);
    }
    int exact_div_variant = -1;        // what -no-hip_fast_div selects instead of a default whose divisions are a * rcp(b)
    void set_exact_div(const char* name) {
        for (size_t i = 0; i < variants.size(); i++)
            if (std::string(variants[i].name) == name) { exact_div_variant = (int)i; return; }
        throw std::runtime_error(std::string("no kernel variant named ") + name);
    }
    void set_large_grid(const char* name) {
        for (size_t i = 0; i < variants.size(); i++)
            if (std::string(variants[i].name) == name) { large_grid_variant = (int)i; return; }
        throw std::runtime_error(std::string("no kernel variant named ") + name);
    }
};
// A run of scratch stages and the stage they feed as ONE kernel with the scratch vars in the LDS (ykh_fused.hpp; 2-D solutions)
struct FusedGeom;
struct FusedGroupImpl {
    int first_stage = 0, last_stage = 0;      // indices into SolnMeta::stages
    int n_parts = 0;                          // parts of those stages, in order = the kernel's PartArgs array
    size_t lds_bytes = 0;
    int threads = 0, ti = 0, tj = 0, n_slots = 0, n_scratch_vars = 0;
    void (*launch)(const PartArgs* dev_args, const FusedGeom& g, unsigned grid, hipStream_t s) = nullptr;
    const void* func = nullptr;
};
struct SolnImpl {
    const SolnMeta* meta;
    std::vector<PartImpl> parts;
    bool select_by_timing = false;     // prepare_solution() times the legal shapes of each part once and keeps the fastest
    std::vector<FusedGroupImpl> fused;
};
// Defined once per stencil library (stencil_<name>.hip).
const SolnImpl& ykh_solution_impl();

// ------------------------------------------------------------------ Env
// Halo-exchange transport. The runtime packs every outgoing halo into a contiguous device buffer and
// hands the transport a list of (peer, send buffer, recv buffer, bytes, tag) messages; the transport
// moves the bytes (RCCL send/recv over xGMI, or a host-provided callback -- e.g. torch.distributed).
struct HaloMsg {
    int peer;           // neighbour rank
    void* send_buf;     // device pointer (contiguous), may be null if nbytes_send == 0
    void* recv_buf;     // device pointer (contiguous)
    size_t send_bytes;
    size_t recv_bytes;
    int tag;
    // Which of the receiver's buffers this message lands in, named the SAME way on both ends of the link: 0 = the neighbour's
    // packed receive buffer for this direction, (var ordinal * 16 + step slot) + 1 = the planes of that var slot (in-place x
    // faces), plus the solution's ordinal within its env x 4096 (solutions are created in the same order on every rank; two
    // solutions of one env have different buffers).  A transport that maps the peer's buffers (ykh_ipc.cpp) learns the address of
    // a (peer, tag, key) once.
    int key;
};
typedef int (*ykh_exchange_fn)(void* user, int nmsgs, const HaloMsg* msgs, void* stream);
typedef int (*ykh_allreduce_fn)(void* user, int op /*0 sum,1 min,2 max*/, long long* val);

class Env {
public:
    int rank = 0, nranks = 1;
    int device = 0;
    int num_cus = 256;                      // compute units of the device (MI355X: 256)
    ykh_exchange_fn exch_start = nullptr;   // begin moving all msgs (async on `stream`)
    ykh_exchange_fn exch_wait = nullptr;    // make `stream` wait until they have landed
    ykh_allreduce_fn allreduce = nullptr;
    void (*exch_reset)(void* user) = nullptr;   // optional: buffers a transport may have cached addresses of are being freed / re-allocated
    int (*exch_begin)(void* user) = nullptr;    // optional, COLLECTIVE: called by every rank before the first exchange of a run_solution() / exchange_halos() call
    int (*exch_check)(void* user) = nullptr;    // optional: non-zero if an exchange failed asynchronously (called when the streams have drained)
    int (*exch_counters)(void* user, long long* out, int cap) = nullptr;   // optional: control-plane counters (yk_env_get_transport_counters)
    void* user = nullptr;
    void (*user_free)(void*) = nullptr;     // releases `user` (built-in transports own their state; host callbacks do not)
    bool trace = false;
    int solutions_made = 0;                 // ordinal of the next Solution of this env (HaloMsg::key)
    // false: the installed transport reaches a peer's receive buffer by MAPPING the allocation it lies in (ykh_ipc.cpp) -- x faces then
    // travel through the packed buffers like y / z faces instead of straight into the var's planes: a peer never maps var storage
    // (round 4: four bench.py ranks at the headline size hung in hipIpcOpenMemHandle on each other's 2.6 GB vars, gpurun_out/r4t;
    // with packed x faces the same job runs, r4v; the two extra copies of contiguous planes cost < 1 % of a step)
    bool direct_halo_ok = true;
    // a new transport is about to be installed: the old one's state goes, and so does every hook it had set
    void drop_transport() {
        if (user && user_free) user_free(user);
        user = nullptr; user_free = nullptr;
        exch_start = exch_wait = nullptr; allreduce = nullptr;
        exch_reset = nullptr; exch_begin = nullptr; exch_check = nullptr; exch_counters = nullptr;
        direct_halo_ok = true;
    }
    Env();
    ~Env() { if (user && user_free) user_free(user); }
    Env(const Env&) = delete;
    Env& operator=(const Env&) = delete;
    void set_ranks(int rank_, int nranks_);
    long long sum_over_ranks(long long v) const;
    long long min_over_ranks(long long v) const;
    long long max_over_ranks(long long v) const;
};

// ------------------------------------------------------------------ Var
class Solution;

struct VarDim {
    std::string name;
    int type;          // DimType
    int domain_idx;    // for domain dims
    idx_t first_misc = 0, last_misc = 0;   // misc dims (and the outer domain dim: its allocated index range, pads included)
    bool is_outer = false;                 // the 4th (outermost) domain dim of the solution: stored like a misc dim
    idx_t outer_halo_l = 0, outer_halo_r = 0;
};

class Var {
public:
    Var(Solution* soln, const VarMeta* meta, int ordinal);
    Var(Solution* soln, const std::string& name, const std::vector<std::string>& dims, int ordinal,
        const std::vector<idx_t>* fixed_sizes);
    ~Var();

    Solution* soln;
    const VarMeta* meta;       // null for user-created vars
    std::string name;
    int ordinal;
    std::vector<VarDim> dims;
    bool fixed_size = false;
    std::vector<idx_t> fixed_sizes;     // by dim position (fixed-size vars)
    bool has_step = false;
    int step_posn = -1;
    int nslots = 1;
    bool dynamic_step_alloc = false;
    int l1_norm = 0;
    bool is_written = false;

    // per domain_idx geometry (valid after Solution::prepare or set for fixed-size vars)
    bool uses_domain[MAX_DOMAIN_DIMS] = {false, false, false};
    idx_t dom_size[MAX_DOMAIN_DIMS] = {1, 1, 1};      // rank-domain size in this dim
    idx_t rank_ofs[MAX_DOMAIN_DIMS] = {0, 0, 0};      // global index of local 0
    idx_t halo_l[MAX_DOMAIN_DIMS] = {0, 0, 0}, halo_r[MAX_DOMAIN_DIMS] = {0, 0, 0};
    idx_t min_pad_l[MAX_DOMAIN_DIMS] = {0, 0, 0}, min_pad_r[MAX_DOMAIN_DIMS] = {0, 0, 0};
    idx_t pad_l[MAX_DOMAIN_DIMS] = {0, 0, 0}, pad_r[MAX_DOMAIN_DIMS] = {0, 0, 0};
    idx_t stride[MAX_DOMAIN_DIMS] = {0, 0, 0};        // element stride (0 if dim unused)
    idx_t misc_elems = 1;                              // product of misc-dim sizes
    std::vector<idx_t> misc_stride;                    // by dim position
    idx_t slot_elems = 0;                              // elements per step slot
    idx_t first_valid_step = 0;

    int elem_bytes() const;
    bool is_allocated() const { return dptr != nullptr; }
    void compute_geometry();          // sizes/strides from soln settings (no allocation)
    void allocate();                  // hipMalloc + zero
    void release();

    size_t bytes() const { return (size_t)slot_elems * nslots * elem_bytes(); }

    // device address of local element (0,..,0) of the slot holding step t
    void* slot_base(idx_t t) const;
    int slot_of(idx_t t) const;
    idx_t last_valid_step() const { return first_valid_step + nslots - 1; }
    void update_valid_step(idx_t t);

    // index helpers (by dim position, global indices as in the reference API)
    idx_t first_local_index(int posn) const;
    idx_t last_local_index(int posn) const;
    idx_t alloc_size(int posn) const;
    int dim_posn(const std::string& dim, bool must_exist = true) const;
    bool indices_local(const std::vector<idx_t>& idx) const;
    std::string format_indices(const std::vector<idx_t>& idx) const;
    void check_indices(const std::vector<idx_t>& idx, const char* fn, bool strict, bool check_step,
                       bool* clipped = nullptr) const;

    // element / slice access (global indices, inclusive bounds)
    double get_element(const std::vector<idx_t>& idx) const;
    idx_t set_element(double v, const std::vector<idx_t>& idx, bool strict);
    idx_t add_to_element(double v, const std::vector<idx_t>& idx, bool strict);
    idx_t get_elements_in_slice(void* buf, size_t buf_elems, int buf_elem_bytes,
                                const std::vector<idx_t>& first, const std::vector<idx_t>& last) const;
    idx_t set_elements_in_slice(const void* buf, size_t buf_elems, int buf_elem_bytes,
                                const std::vector<idx_t>& first, const std::vector<idx_t>& last);
    idx_t set_elements_in_slice_same(double v, const std::vector<idx_t>& first,
                                     const std::vector<idx_t>& last, bool strict);
    void set_all_elements_same(double v);
    // extension: logical-index hash init (offset + scale*H(ordinal, slot, gx, gy, gz)) over
    // domain+halo of every slot; same function as oracle/stencil_oracle.c:yo_hash_unit.
    void set_elements_hash(double offset, double scale, int hash_id);
    struct Reduction { idx_t n = 0; double sum = 0, sum_sq = 0, prod = 1, vmax = 0, vmin = 0; int mask = 0; };
    Reduction reduce_elements_in_slice(int mask, const std::vector<idx_t>& first,
                                       const std::vector<idx_t>& last, bool strict) const;
    // count of mismatching in-domain elements vs another var (compare_data, yk_var.cpp:401-477)
    idx_t compare(const Var& ref, double epsilon) const;

    // dirty flags per slot (halo needs exchange), src/kernel/lib/yk_var.cpp:122-152
    std::vector<char> dirty;
    void set_dirty(bool d, idx_t t);
    void set_dirty_all(bool d);
    bool is_dirty(idx_t t) const;

    // host mirror for get_raw_storage_buffer(), kept coherent around every API call while exposed (ykh_var.cpp)
    void* host_mirror();
    void sync_mirror_to_device();
    void before_device_use() const;     // host copy -> device, if the raw buffer has been handed out
    void after_device_write();          // device -> host copy, if the raw buffer has been handed out
    void release_raw_storage();         // extension: stop keeping the host copy coherent (the pointer becomes invalid)
    bool raw_exposed() const { return raw_exposed_; }
    void* scratch = nullptr;     // one extra slot (same geometry), used by Solution::run_fused()
    void* dptr = nullptr;        // first byte of the var's storage ...
    void* alloc_ptr = nullptr;   // ... inside this allocation (dptr = alloc_ptr + the var's skew, Var::allocate)
    size_t alloc_bytes = 0;      // size of that allocation (a changed step/misc allocation must re-allocate)
    // Shared ownership of the allocation (generic_var.hpp:136-140: storage is a shared_ptr): hipFree() runs when the last
    // var holding it lets go.  fuse_vars() (yk_var_api.hpp:1370-1396) makes this var "another reference to the source var":
    // the vars of a fuse group share storage, valid-step window and dirty flags, and release() on one applies to all.
    std::shared_ptr<void> alloc_owner;
    std::shared_ptr<std::vector<Var*>> fuse_group;      // every var fused with this one, this one included (null: not fused)
    void fuse_with(Var& src);
    void adopt_storage(std::shared_ptr<void> owner, void* alloc, void* data, size_t nbytes);   // ... for the whole fuse group
    static std::shared_ptr<void> own_allocation(void* alloc);                                   // owner whose deleter is hipFree
    bool storage_fits() const { return dptr && alloc_bytes == std::max<size_t>(bytes(), 256); }
    void drop_storage_refs();    // this var only: forget the storage (freed when no var of the fuse group holds it any more)
    idx_t origin_elems = 0;      // element offset of local (0,0,0), misc first, within a slot
private:
    void init_dims_from_meta();
    mutable std::vector<char> mirror_;
    bool raw_exposed_ = false;
    // iterate the (slot, misc) combinations of a slice and call fn(slot_base_ptr_at_misc, buffer offset)
    template <class F> idx_t for_boxes(const std::vector<idx_t>& first, const std::vector<idx_t>& last,
                                       bool strict, bool update_step, F&& fn) const;
};

// ------------------------------------------------------------------ Stats (yk_stats)
struct Stats {
    idx_t num_elements = 0;
    idx_t num_steps_done = 0;
    idx_t num_writes_done = 0;
    idx_t est_fp_ops_done = 0;
    idx_t num_reads_done = 0;
    double elapsed_secs = 0.0;
    double halo_secs = 0.0;         // device time of pack + transport + unpack (+ host time of the initial exchange)
    double pts_per_sec = 0.0;
    // Per-phase breakdown, from HIP events on the two streams (the reference's halo_pack/unpack/wait and
    // ext/int timers, src/kernel/lib/context.hpp:319-328, reported by soln_apis.cpp:349-562)
    double halo_pack_secs = 0.0, halo_xfer_secs = 0.0, halo_unpack_secs = 0.0;
    double halo_wait_secs = 0.0;    // compute stream idle until the halos have landed = communication NOT hidden
    double exterior_secs = 0.0, interior_secs = 0.0;
    idx_t halo_bytes_sent = 0, halo_bytes_recv = 0, halo_msgs_sent = 0;
    idx_t fused_passes = 0;         // two-steps-per-pass launches (each counts as 2 of num_steps_done)
    idx_t graph_replays = 0, graph_steps = 0;   // replays of captured step chains and the steps they advanced
};

// ------------------------------------------------------------------ Solution
struct Box { idx_t lo[MAX_DOMAIN_DIMS], hi[MAX_DOMAIN_DIMS]; bool empty() const; };

struct NeighborXfer;   // halo buffers per neighbour (below)

class Solution {
public:
    Solution(std::shared_ptr<Env> env, const SolnImpl& impl);
    ~Solution();

    std::shared_ptr<Env> env;
    int ordinal = 0;                        // n-th solution made in this env (the same on every rank): part of HaloMsg::key
    const SolnImpl& impl;
    const SolnMeta* meta;
    int ndd;                                   // number of domain dims
    std::vector<std::string> domain_dim_names;
    std::string step_dim_name;
    std::vector<std::string> misc_dim_names;

    // ---- settings (KernelSettings, src/kernel/lib/settings.hpp:140-330)
    // (per-dim arrays: index 0..2 = the kernels' x, y, z; index 3 = the outer dim of a solution with 4 domain dims)
    idx_t global_size[MAX_API_DOMAIN_DIMS] = {0, 0, 0, 0};
    idx_t rank_size[MAX_API_DOMAIN_DIMS] = {0, 0, 0, 0};     // requested (0 = derive)
    idx_t num_ranks[MAX_API_DOMAIN_DIMS] = {0, 0, 0, 0};
    idx_t rank_index[MAX_API_DOMAIN_DIMS] = {0, 0, 0, 0};
    bool rank_index_set = false;
    idx_t block_size[MAX_API_DOMAIN_DIMS + 1] = {0, 0, 0, 0, 0};   // [0]=step, then domain dims; accepted, advisory
    idx_t mega_block_size[MAX_API_DOMAIN_DIMS + 1] = {0, 0, 0, 0, 0};   // -Mbt (wave-front steps) / -Mbx (slab width); y, z unused
    idx_t min_pad[MAX_API_DOMAIN_DIMS] = {0, 0, 0, 0};
    idx_t extra_pad[MAX_API_DOMAIN_DIMS] = {0, 0, 0, 0};
    bool has_outer = false;                   // solution with 4 domain dims: the outermost one (DIM_OUTER, ykh_meta.hpp)
    std::string outer_dim_name;
    idx_t cur_outer = 0;                      // index in the outer dim of the launches being issued
    bool in_outer_loop = false;
    std::vector<std::string> api_domain_dim_names() const {
        std::vector<std::string> v;
        if (has_outer) v.push_back(outer_dim_name);
        v.insert(v.end(), domain_dim_names.begin(), domain_dim_names.end());
        return v;
    }
    bool overlap_comms = true;
    bool step_wrap = false;        // set_step_wrap(): any step index is accepted and wrapped onto the slots
    idx_t min_exterior = 0;
    bool thin_slab_point_kernel = true;   // -[no-]hip_thin_slab_point_kernel: thin y/z exterior slabs use the point kernel
    bool direct_halo = true;       // -[no-]hip_direct_halo: in-place transfer of contiguous x-face halos
    idx_t placement_trials = 1;    // -hip_placement_trials <n>: sets of var allocations prepare_solution() draws and times (tune_placement());
                                   // 1 (default) = take the first allocation: the search runs real kernels, holds two sets of arrays for
                                   // a moment and costs ~1-2 s, so it is the caller's decision (bench.py and the harnesses ask for 6)
    std::vector<float> placement_ms;   // ms per step measured on each set drawn by the last prepare(), and the one kept
    int placement_chosen = 0;
    void tune_placement();
    bool fast_div = true;          // -[no-]hip_fast_div: shapes whose fp32 divisions are a * v_rcp_f32(b) (<= 1.5 ulp; ssg's defaults, see
                                   // MarchAcc) may be the default; off = the correctly rounded siblings (PartImpl::exact_div_variant)
    idx_t step_graphs = -1;        // -hip_step_graphs: 1 = single-rank runs of several steps are captured once into a hipGraph (the
                                   // launches of a whole number of slot periods) and replayed: one host call per replay instead
                                   // of one per launch, and the command processor sees the whole chain; 0 = plain launches;
                                   // -1 (default) = on for rank boxes of up to 2^20 points (see step_graph_wanted(): measured)
    bool do_halo_exchange = true;
    bool auto_tune = false;        // tuned at prepare() when true
    double auto_tune_trial_secs = 0.05;
    bool force_scalar = false;     // use the naive kernel everywhere
    std::string variant_override;  // -hip_variant <name>
    idx_t xchunk_override = 0;     // -hip_xchunk <n>
    std::map<std::string, std::string> ignored_opts;   // accepted reference options with no GPU meaning
    std::string apply_command_line_options(const std::vector<std::string>& args);
    std::string get_command_line_help() const;
    std::string get_command_line_values() const;

    // ---- state
    bool prepared = false;
    idx_t rank_ofs[MAX_API_DOMAIN_DIMS] = {0, 0, 0, 0};
    idx_t local_size[MAX_API_DOMAIN_DIMS] = {0, 0, 0, 0};   // computed rank-domain size
    std::vector<std::shared_ptr<Var>> vars;
    std::map<std::string, std::shared_ptr<Var>> var_map;
    std::vector<std::shared_ptr<Var>> scratch_vars;   // compiler-declared scratch vars: device arrays, not in the API
    std::vector<HaloMsg> pending_msgs;               // halo messages in flight (exchange_halos start -> finish)
    std::vector<int> part_variant;                   // chosen variant per part
    std::vector<Box> part_bb;                        // sub-domain parts: bounding box of the condition (local indices)
    std::vector<char> part_has_bb, part_bb_solid;    // solid: the condition holds everywhere in the box
    // sub-domain parts whose condition does not fill its bounding box: the reference's list of FULL bounding boxes (non-overlapping,
    // valid points only: StencilPartBase::_bb_list, setup.cpp:1235-1500) where prepare_solution() found one (find_part_boxes);
    // the part then runs its unpredicated kernels box by box.  Empty: the point kernel evaluates the condition per point.
    std::vector<std::vector<Box>> part_boxes;
    bool in_part_boxes_ = false;                     // launch_part_variant is walking a part's box list
    // per BOX of a part's list: the kernel shape the timing pass kept for that box (a shell's z-slabs are 20 points thin, its x-slabs
    // are whole planes: one shape does not fit both); empty = the part's shape for every box.  Used when the part runs on its tuned shape.
    std::vector<std::vector<int>> part_box_variant;
    bool find_part_boxes(int part, const Box& bb, unsigned long long count, std::vector<Box>& out);
    std::vector<Box> part_hole;            // 2-D: the solid box of points where a ring-shaped condition does not hold (empty: none)
    bool part_needs_predicate(int part) const {      // only the point kernel evaluates the condition per point
        const PartMeta& pm = *impl.parts[part].meta;
        if (pm.has_step_cond_dev) return true;
        if (!pm.has_domain_cond) return false;
        if ((size_t)part < part_boxes.size() && !part_boxes[part].empty()) return false;
        return !((size_t)part < part_has_bb.size() && part_has_bb[part] && part_bb_solid[part]);
    }
    std::vector<idx_t> part_xchunk;
    hipStream_t compute_stream = nullptr, comm_stream = nullptr;
    bool own_streams = false;
    Stats stats;
    idx_t steps_done_total = 0;

    // neighbours (MPIInfo, src/kernel/lib/settings.hpp:331-433)
    struct Neighbor { int rank; int ofs[MAX_DOMAIN_DIMS]; int l1; };
    std::vector<Neighbor> neighbors;

    // hooks (soln_apis.cpp:116-133)
    typedef std::function<void(Solution&)> hook_fn;
    typedef std::function<void(Solution&, idx_t, idx_t)> hook_t_fn;
    std::vector<hook_fn> before_prepare, after_prepare;
    std::vector<hook_t_fn> before_run, after_run;

    // ---- API
    void set_streams(hipStream_t compute, hipStream_t comm);
    std::shared_ptr<Var> get_var(const std::string& name) const;
    std::shared_ptr<Var> new_var(const std::string& name, const std::vector<std::string>& dims);
    std::shared_ptr<Var> new_fixed_size_var(const std::string& name, const std::vector<std::string>& dims,
                                            const std::vector<idx_t>& sizes);
    int domain_dim_idx(const std::string& dim, const char* fn) const;
    void invalidate() { prepared = false; }
    void prepare();
    void end();
    void run(idx_t first_step, idx_t last_step);
    void run_wavefront(idx_t t0, idx_t nsteps, idx_t dir);
    // ---- captured step graphs (run(): plain single-rank loop)
    struct StepGraph { std::string key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; idx_t steps = 0; idx_t nodes = 0; };
    std::vector<StepGraph> step_graph_cache;           // most recently used last; dropped by prepare() / end() / the tuner
    void drop_step_graphs();
    void note_storage_changed();                       // a var's allocation was made, freed or shared: transports forget cached addresses
    idx_t slot_period() const;                         // steps after which every var is back in the same slots (lcm of the slot counts)
    bool step_graph_eligible() const;                  // every launch of a step depends on t through the base pointers only
    bool step_graph_wanted() const;
    std::string step_graph_key(idx_t t, idx_t dir, idx_t steps) const;
    void issue_step(idx_t t);                          // one plain single-rank step: every stage's parts over the rank box
    void note_stage_written(const StageMeta& sm, idx_t t);   // valid-step / dirty bookkeeping of a stage's written vars
    void note_step_written(idx_t t);
    StepGraph* get_step_graph(idx_t t, idx_t dir, idx_t steps);
    Box interior_for(const bool* has_lo, const bool* has_hi) const;
    void launch_exterior(const StageMeta& sm, idx_t t, const Box& ib);
    void launch_interior(const StageMeta& sm, idx_t t, const Box& ib);
    // ---- planned launches of a decomposed rank: the whole rank box as ONE launch of the marching kernel, shell blocks first,
    // the halo exchange released from the device when the shell is done (ykh_plan.cpp plan_blocks; replaces exterior slabs +
    // interior pieces wherever the stage's kernel takes block descriptors)
    bool planned_launch = true;        // -[no-]hip_planned_launch
    struct LaunchPlan { std::string key; BlockPlan plan; BlockDesc* dev = nullptr; size_t cut = 0; };      // cut: end of the round that holds the last shell block
    std::vector<std::unique_ptr<LaunchPlan>> launch_plans;
    unsigned* lockstep_dev = nullptr;  // per-XCD arrival counters of the "_ls<K>" shapes (8 x 32 words), zeroed before each such launch
    std::map<std::tuple<const void*, int, size_t>, int> resident_cache_;
    int resident_blocks(const KernelVariant& kv);
    bool shell_event_pending = false;  // ev_shell was recorded behind the shell part of the launch just issued
    hipEvent_t ev_shell = nullptr;
    void exchange_build_and_pack(hipStream_t st);
    int planned_part(const StageMeta& sm) const;        // the stage's one part if it can run as a planned launch, else -1
    int planned_variant_of(int part) const;             // the kernel shape whose descriptor-reading twin runs the part's planned launches, or -1
    mutable std::vector<int> planned_cache_;             // ... remembered per part (-2: not looked up yet); cleared with the launch plans
    LaunchPlan* get_launch_plan(int part, const bool* has_lo, const bool* has_hi, bool wide_shell = false, int mode = -1);   // mode < 0: 0
    void launch_planned(int part, idx_t t, LaunchPlan& lp, bool signal, hipStream_t s);
    // ---- pipelined half-exchanges (-[no-]hip_halves): a stage of a decomposed rank is TWO launches in regular
    // order, the outer x-half A = [0, q1) u [q2, nx) and the inner half B = [q1, q2) (plan mode 4), each followed by the exchange of
    // its own part of the faces.  The exchange started behind one half is finished -- unpacked, the compute stream made to wait for
    // it -- only after the NEXT half has been launched: A's halos are needed again by the next stage's A launch, a whole B launch
    // later.  No shell-first order (what planned launches pay for, section 4.1), and every transfer has one launch to hide behind.
    // Legal when what a half reads of a neighbour's data lies in the planes of that half: every written var is read face-only
    // (l1_norm <= 1; iso3dfd, 3axis, ssg), every stage one planned part, no wave-front extension; agreed across ranks in prepare().
    struct PhaseEvents;
    bool halves = true;                // -[no-]hip_halves (default since round 5: regular launch order costs 1.01-1.02x the undivided sweep where the
                                       // shell-first plan costs 1.05-1.12x, and every transfer has a whole launch to hide behind)
    bool halves_geom_ok_ = false;      // prepare(): legal here AND on every other rank
    idx_t halves_q1_ = 0, halves_q2_ = 0;
    int exch_half_ = -1;               // which slab lists exchange_halos() works on: -1 whole faces, 0 / 1 the halves
    bool halves_in_flight_ = false;    // a half-exchange has been started and not finished
    int halves_flight_half_ = 0;
    PhaseEvents* halves_flight_phase_ = nullptr;
    bool halves_geometry(idx_t* q1, idx_t* q2) const;   // this rank's box can be cut into the two halves (local; no collective)
    bool halves_active() const;        // this run() uses the schedule (options + halves_geom_ok_ + every stage planned)
    void launch_planned_half(int part, idx_t t, LaunchPlan& lp, int half, hipStream_t s);
    void halves_start(int half);       // behind the launch of `half`: pack + send its faces (comm stream, after ev_shell)
    void halves_finish();              // the exchange in flight: wait, unpack, compute stream waits for it
    void run_stage_halves(const StageMeta& sm, int st, idx_t t, const bool* has_lo, const bool* has_hi);
    void drop_launch_plans();
    void neighbor_sides(bool* has_lo, bool* has_hi) const;
    void time_decomposed_step(const bool* has_lo, const bool* has_hi, int reps, float* ms3);
    // on-chip fusion of two steps per pass (-hip_fuse_steps 2; ykh_starlin2.hpp)
    idx_t fuse_steps = 0;      // -hip_fuse_steps: 2 = on where the solution has such a kernel, 0/1 = off (default: run_solution(a, b) is then
                               // bit-identical to one call per step; the fused pass contracts its FMAs differently, <= 1e-13 apart in fp64)
    bool can_fuse() const;
    void run_fused(idx_t t0, idx_t npairs, idx_t dir);
    void launch_fused(idx_t t, const void* src, void* slot_b, void* dst, bool store_b);
    void exchange_halos_all();
    void check_async_errors(const char* who);      // after the streams have drained
    Stats get_stats();       // returns and clears, like soln_apis.cpp:349-562
    void reset_auto_tuner(bool enable);
    void run_auto_tuner_now();
    void tune_variants(bool quick, bool fresh_storage = false);      // fresh_storage: no var holds data yet (prepare_solution() allocated them all)
    idx_t compare_data(const Solution& ref, double epsilon) const;
    void copy_vars_to_device() {}
    void copy_vars_from_device() {}
    void synchronize();

    // internals used by ykh_halo.cpp / tuner
    void setup_rank();
    void launch_part(int part, idx_t t, const Box& box, hipStream_t s);
    Box scratch_grown_box(int part, const Box& box) const;
    // fused scratch groups (ykh_fused.hpp): usable on this solution / on for the coming steps; per group and step slot phase the
    // device array of the parts' PartArgs (built by ensure_fused_args(), outside any stream capture)
    int fuse_scratch_mode = -1;            // YASK_HIP_FUSE_SCRATCH: 0 off, 1 on wherever legal, -1 (default) decided by timing at prepare_solution()
    bool fused_on = false;
    bool fused_usable() const;
    bool fused_ok_at(const FusedGroupImpl& fg, idx_t t) const;
    const FusedGroupImpl* fused_group_at(int stage) const;
    void ensure_fused_args();
    void drop_fused_args();
    void launch_fused(const FusedGroupImpl& fg, idx_t t, hipStream_t s);
    std::map<int, int> fused_pick_;                       // first stage of a group -> index into impl.fused of the tile shape in use
    std::vector<std::vector<PartArgs*>> fused_args_;      // [entry of impl.fused][phase]
    std::string fused_args_key_;
    bool launching_interior = false;      // set by launch_interior() of an overlapped exchange
    bool launching_exterior = false;      // set by run() around the exterior slabs of a decomposed run (thin-slab kernel choice)
    void launch_part_variant(int part, int variant, idx_t xchunk, idx_t t, const Box& box_in, hipStream_t s);
    void fill_part_args(int part, idx_t t, const Box& box, PartArgs& a) const;
    bool halo_built_direct_ok = true;  // env->direct_halo_ok when alloc_halo_buffers() last ran
    void alloc_halo_buffers();
    void free_halo_buffers();
    void exchange_halos(idx_t t_written, int stage, bool start_only, bool finish_only);
    Box rank_box() const;
    Box interior_box;                 // rank box shrunk where a neighbour exists (overlap_comms)
    bool have_interior = false;
    std::vector<std::unique_ptr<NeighborXfer>> xfers;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    // phase timers: one set of events per (step, stage) of a multi-rank run, read back when run() has drained
    enum { PH_EXT0, PH_EXT1, PH_INT1, PH_WAIT1, PH_PACK0, PH_PACK1, PH_XFER1, PH_UNPACK1, PH_N };
    struct PhaseEvents { hipEvent_t e[PH_N]; bool rec[PH_N]; };
    std::vector<PhaseEvents> phase_pool;       // a ring of at most PHASE_RING sets: the oldest is folded into `stats` before it is reused
    enum { PHASE_RING = 64 };
    size_t phase_used = 0;                     // sets handed out since the last phase_collect()
    bool phase_timers = true;                  // -[no-]hip_phase_timers: the per-phase events of multi-rank runs (8 records per stage)
    void phase_fold(const PhaseEvents& ph);
    PhaseEvents* cur_phase = nullptr;          // set by run() around a stage; exchange_halos() records into it
    void phase_mark(int which, hipStream_t st);
    PhaseEvents* phase_next();
    void phase_collect();
    bool step_timers = false;                  // -[no-]hip_step_timers: one event per step, read by get_step_times()
    std::vector<hipEvent_t> step_events;
    std::vector<float> step_ms;                // per-step durations of the last run() (ms)
    bool tune_at_prepare = true;               // -no-auto_tune also switches off prepare()'s one-off shape timing
    int elem_bytes() const { return meta->elem_bytes; }
    idx_t shared_pad_l(int d) const { return shared_pad_l_[d]; }
    idx_t shared_pad_r(int d) const { return shared_pad_r_[d]; }
    idx_t shared_pad_l_[MAX_DOMAIN_DIMS] = {0, 0, 0}, shared_pad_r_[MAX_DOMAIN_DIMS] = {0, 0, 0};
    // ---- wave-front tiling across ranks (-Mbt n with neighbours; the reference's wave-front extensions, setup.cpp:717-805,
    // context.cpp:286-346): a group of n steps = P = n x stages phases; phase p is evaluated on the rank box EXTENDED by
    // angle x (P-1-p) towards every neighbour -- redundantly with that neighbour -- so that no exchange is needed inside the
    // group; halos (and pads) are wf_ext = angle x (P-1) wider and are exchanged ONCE per group, with all 26 neighbours.
    idx_t wf_angle_[MAX_DOMAIN_DIMS] = {0, 0, 0};     // widest halo per dim = the shift per phase
    idx_t wf_ext_[MAX_DOMAIN_DIMS] = {0, 0, 0};       // extra halo / pad per dim (0: plain sweeps)
    idx_t wf_ext(int d) const { return wf_ext_[d]; }
    bool wf_multi() const { return wf_ext_[0] + wf_ext_[1] + wf_ext_[2] > 0; }
    void run_wavefront_multi(idx_t t0, idx_t nsteps, idx_t dir, const bool* has_lo, const bool* has_hi, bool exchange);
};

// one var's region for one neighbour, and the per-neighbour buffers (see ykh_halo.cpp)
struct Slab {
    int var;            // index into Solution::vars
    idx_t lo[3], n[3];  // local box
    idx_t elems;        // per step slot (all misc indices)
};
struct NeighborXfer {
    Solution::Neighbor nb;
    std::vector<Slab> send, recv;
    // the same slabs cut at the x planes of the two halves of the pipelined schedule (Solution::halves_*; plan halves_slab_ranges):
    // [0] what travels after the outer half's launch, [1] after the inner half's
    std::vector<Slab> send_h[2], recv_h[2];
    void* send_buf = nullptr;
    void* recv_buf = nullptr;
    size_t send_cap = 0, recv_cap = 0;   // bytes
    size_t send_now = 0, recv_now = 0;   // bytes in the current exchange
    // x-only neighbour and every slab belongs to a full-dim var read face-only: the slab's planes, taken with
    // their y/z pads, are ONE contiguous range of the var -> send/receive in place, no pack/unpack kernels
    bool direct = false;
};

std::string version_string();

// pads (everything outside the domain box) of one dense [alloc0][alloc1][alloc2] step slot -> another (ykh_util_kernels.hip)
void launch_copy_pads(const void* src, void* dst, int elem_bytes, const idx_t alloc[3], const idx_t pad_l[3], const idx_t dom[3],
                      hipStream_t s);
// streaming-bandwidth probe (ykh_util_kernels.hip): kind 0 copy, 1 three reads + one write, 2 read only; GB/s
double probe_bandwidth(int kind, size_t bytes, int reps);

// stream-ordered waits / stores on words in device memory (ykh_util_kernels.hip): the stream continues when every word i has
// reached vals[i]; after timeout_s the waiter gives up and raises *err
void launch_wait_words(int n, const unsigned* const* ptrs, const unsigned* vals, unsigned* err, double timeout_s, hipStream_t s);
void launch_set_words(int n, unsigned* const* ptrs, const unsigned* vals, hipStream_t s);

// utility kernels (ykh_util_kernels.hip)
struct BoxCopyArgs {
    void* var_base;       // address of local element (0,0,0) of the slot (+misc offset)
    void* buf;            // contiguous buffer
    idx_t sx, sy, sz;     // var strides
    idx_t lo[3], n[3];    // box origin (local) and extent
    idx_t bs[3];          // buffer strides (elements) for box dims x,y,z
    int var_elem_bytes, buf_elem_bytes;
};
// Halo pack / unpack (ykh_util_kernels.hip): ALL slabs of an exchange -- every neighbour, var, step slot, misc index -- move
// in one launch per ~40 segments; a thread moves one 16-byte vector (or one element where a slab's z extent / alignment does not
// allow vectors) of the packed buffer, so every lane is busy whatever the slab's shape (round 2's per-slab kernels put 256
// lanes on the 8 points of a z-face row: 0.25 ms to pack 29 MB beside a running stencil, tools/overlap_probe.py).
struct HaloSeg {
    void* var_base;       // address of local element (0,0,0) of the slot (+misc offset)
    void* buf;            // where the segment starts in the packed message
    idx_t sx, sy, sz;     // var strides (elements)
    int lo[3], n[3];      // slab origin (local) and extent
};
void launch_halo_move(const std::vector<HaloSeg>& segs, bool pack, int elem_bytes, hipStream_t s);
void launch_linear_copy(void* dst, const void* src, size_t bytes, unsigned long long* t0, hipStream_t st);
void launch_hold_until(unsigned long long* t0, double seconds, hipStream_t st);
void launch_box_gather(const BoxCopyArgs& a, hipStream_t s);    // var -> buf
void launch_box_scatter(const BoxCopyArgs& a, hipStream_t s);   // buf -> var
void launch_box_fill(const BoxCopyArgs& a, double v, hipStream_t s);
void launch_box_add(const BoxCopyArgs& a, double v, hipStream_t s);
void launch_box_hash(const BoxCopyArgs& a, double offset, double scale, idx_t vid, idx_t slot,
                     idx_t gx, idx_t gy, idx_t gz, hipStream_t s);
// reduce over a box: out[0]=sum out[1]=sumsq out[2]=prod out[3]=max out[4]=min (doubles, device memory)
void launch_box_reduce(const BoxCopyArgs& a, double* out5, hipStream_t s);
// count elements of box a differing from box b (same extents) beyond epsilon (reference rule)
void launch_box_compare(const BoxCopyArgs& a, const BoxCopyArgs& b, double eps, unsigned long long* count,
                        hipStream_t s);

}  // namespace ykh
