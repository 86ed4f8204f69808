// ykh_meta.hpp -- compile-time description of a stencil solution, as emitted by the
// `cdna4_hip` compiler target (yask_amd/compiler/YaskHipPrinter.cpp) and consumed by the HIP
// runtime. It carries the same facts the reference bakes into its generated context
// (src/compiler/lib/YaskKernel.cpp:730-1181 `print_context`): dims, vars, per-var halos,
// step-slot counts, L1 norms, parts, stages and their dependencies.
#pragma once
#include <type_traits>

namespace ykh {

using idx_t = long long;

constexpr int MAX_DOMAIN_DIMS = 3;   // the kernels are specialised for <=3 spatial dims (x, y, z) ...
constexpr int MAX_API_DOMAIN_DIMS = 4;   // ... a 4th, outermost domain dim is an outer loop of launches (DIM_OUTER)
constexpr int MAX_VAR_DIMS = 6;

// DIM_OUTER: the outermost domain dim of a solution with 4 domain dims (`test_4d`: t, w, x, y, z).  For the API it is a
// domain dim (sizes, halos, indices); for storage and kernels it behaves like a misc dim whose index range includes the
// pads: an access group carries its offset in that dim (`dw`) and the runtime launches the 3-D kernels once per index.
enum DimType { DIM_STEP = 0, DIM_DOMAIN = 1, DIM_MISC = 2, DIM_OUTER = 3 };

struct DimMeta {
    const char* name;
    int type;        // DimType
    int domain_idx;  // 0..ndomain-1 for domain dims (0 = outermost = largest stride), else -1
};

struct VarMeta {
    const char* name;
    int ndims;
    int dims[MAX_VAR_DIMS];           // indices into SolnMeta::dims, in declaration order
    int step_alloc;                   // number of step slots (0 if var has no step dim)
    int halo_l[MAX_DOMAIN_DIMS];      // by domain_idx
    int halo_r[MAX_DOMAIN_DIMS];
    int misc_first[MAX_VAR_DIMS];     // by position in dims[] (only misc dims meaningful)
    int misc_last[MAX_VAR_DIMS];
    int l1_norm;                      // max L1 distance of any read (prunes halo-exchange neighbours)
    bool is_scratch;
    bool is_written;                  // updated by some equation
    int outer_halo_l = 0, outer_halo_r = 0;   // halo in the DIM_OUTER dim (solutions with 4 domain dims)
};

// One distinct (var, step offset) pair touched by a part: kernels receive one base pointer each.
struct AccessGroup {
    int var;        // index into SolnMeta::vars
    int dt;         // step offset relative to the evaluation step t (0 if var has no step dim)
    bool has_step;
    int nmisc = 0;                    // constant indices of the var's misc dims, in var-dim order
    int misc[MAX_VAR_DIMS] = {};
    int dw = 0;                       // offset in the DIM_OUTER dim (0 unless the solution has 4 domain dims)
};

// Which of the kernels' three domain dims (bit 0 = x, the marching dim ... bit 2 = z, the unit-stride dim) the var of a part's access
// group has, at compile time: the compiler target's `group_dims` (3-D solutions; parts without the table -- wrappers, hand-written
// parts -- say 7 and get the run-time strides).  An operand without x is loaded once per block instead of once per plane, one
// without z is one value per row instead of a vector, one with neither y nor z is uniform (a scalar load).
template <class P, class = void>
struct GroupDims { static constexpr unsigned get(int) { return 7; } };
template <class P>
struct GroupDims<P, std::void_t<decltype(P::group_dims)>> { static constexpr unsigned get(int g) { return P::group_dims[g]; } };

struct ReadOff {
    signed char g;            // access-group index
    signed char dx, dy, dz;   // offsets by domain_idx (x = domain_idx 0)
};

// One term of a part's *linear star form* (see ykh_starlin.hpp): coefficient c multiplies the read of
// the star group at offset (dx,dy,dz); at most one of the offsets is non-zero.
struct LinTerm {
    signed char dx, dy, dz;
    double c;
};

struct PartMeta {
    const char* name;
    int n_groups;
    const AccessGroup* groups;
    int n_reads;
    const ReadOff* reads;
    int n_writes;
    const int* writes;        // group indices written
    int fp_ops;               // per-point stats, as the reference reports them
    int points_read;
    int points_written;
    int stage;                // owning stage
    bool has_domain_cond;
    bool has_step_cond;
    bool is_scratch = false;                       // writes scratch vars: evaluated over the box grown by their halos
    bool (*step_cond)(long long t) = nullptr;      // IF_STEP predicate (host); null = always
    bool has_step_cond_dev = false;                // IF_STEP predicate that reads var values: evaluated by the point kernel
    bool uses_step_value = false;                  // the VALUE of the step index enters an equation or a device-side condition
                                                   // (launches of different steps differ in more than their base pointers:
                                                   // such solutions are not replayed from a captured step graph)
};

// One workgroup of a *planned launch* (ykh_plan.cpp `plan_blocks`, Solution::launch_planned): the (y, z) tile the block owns,
// clipped to the rank box, and the x range it marches.  flags bit 0: the block counts towards the launch's completion signal
// (its output is needed by a neighbour rank: the halo exchange starts when all such blocks have finished, while the rest of the
// launch is still running).
struct BlockDesc {
    int x0, x1, y0, y1, z0, z1;
    int flags;
    int start;       // planned start time, in plane-iterations from the beginning of the launch (planner's estimate; not read on the device)
};
enum { BLOCK_SIGNALS = 1 };

// a list of part types in evaluation order (fusion groups of the generated headers, YKH_FOR_EACH_FUSE_GROUP)
template <class... Ps>
struct PartList { static constexpr int N = (int)sizeof...(Ps); };

struct StageMeta {
    const char* name;
    int n_parts;
    const int* parts;
};

struct SolnMeta {
    const char* name;
    const char* description;
    const char* target;        // "cdna4_hip"
    int elem_bytes;
    int ndims;
    const DimMeta* dims;       // step dim first, then domain dims (outer->inner), then misc dims
    int step_dir;              // +1 forward, -1 reverse
    int n_vars;
    const VarMeta* vars;
    int n_parts;
    const PartMeta* parts;
    int n_stages;
    const StageMeta* stages;
};

}  // namespace ykh
