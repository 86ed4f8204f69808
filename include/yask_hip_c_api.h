/* yask_hip_c_api.h -- C ABI of the MI355X (cdna4_hip) YASK kernel library.
 *
 * One shared library per stencil solution, named like the reference's
 * (lib/libyask_kernel.<stencil>.<arch>.so, src/common/common.mk:211-216):
 *     libyask_kernel.<stencil>.cdna4_hip.so
 * Every library exports exactly the symbols declared here.  The reference has no C ABI: its
 * boundary is the C++ virtual API of include/yask_kernel_api.hpp, include/aux/yk_solution_api.hpp
 * and include/aux/yk_var_api.hpp, normally reached from other languages through SWIG
 * (src/kernel/swig/yask_kernel_api.i).  Each entry point below names the C++ method it stands for
 * ("replaces: file:line"); argument meaning, index conventions (global / overall-domain indices,
 * inclusive slice bounds, row-major buffers in the var's own dim order) and error behaviour follow
 * that method.  Bindings: INTEGRATION.md shows the yk_* C++ adapter (yask_amd/cxxapi) and the
 * ctypes binding (yask_amd/_capi.py) built on these calls.
 *
 * Conventions
 *   - plain C types only; handles are opaque pointers owned by the library;
 *   - every function returning `int` returns 0 on success and non-zero on failure; the failure
 *     text (always starting with "YASK error: ", like yask::yask_exception::get_message(),
 *     include/yask_common_api.hpp:125-179) is then available from yk_last_error();
 *   - functions returning a value report failure through yk_last_error_code() != 0;
 *   - like the reference API the library is not thread-safe: one calling thread per process,
 *     and collective calls (prepare/run/exchange/tune) must be made on all ranks;
 *   - there is no CPU fallback: creating an env without a visible AMD GPU fails.
 */
#ifndef YASK_HIP_C_API_H
#define YASK_HIP_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t yk_idx_t;               /* yask::idx_t, include/yask_common_api.hpp:86 */
typedef struct yk_env_s* yk_env_h;      /* yask::yk_env      */
typedef struct yk_solution_s* yk_soln_h;/* yask::yk_solution */
typedef struct yk_var_s* yk_var_h;      /* yask::yk_var (borrowed from its solution; valid until yk_free_solution) */

/* ---- errors ---- */
const char* yk_last_error(void);        /* text of the last failure on this thread ("" if none) */
int yk_last_error_code(void);           /* 0 if the last call succeeded */
void yk_clear_error(void);

/* ---- factory: replaces yk_factory, include/yask_kernel_api.hpp:82-161 ---- */
const char* yk_get_version_string(void);                 /* yk_factory::get_version_string, :91 */
yk_env_h yk_new_env(void);                               /* yk_factory::new_env(), :123 */
void yk_free_env(yk_env_h env);
yk_soln_h yk_new_solution(yk_env_h env);                 /* yk_factory::new_solution(env), :145 */
yk_soln_h yk_new_solution_from(yk_env_h env, yk_soln_h source);   /* new_solution(env, source), :156: copies settings */
void yk_free_solution(yk_soln_h soln);

/* ---- env: replaces yk_env, include/yask_kernel_api.hpp:167-295 ---- */
int yk_env_get_num_ranks(yk_env_h env);                  /* :238 */
int yk_env_get_rank_index(yk_env_h env);                 /* :245 */
int yk_env_global_barrier(yk_env_h env);                 /* :253 */
yk_idx_t yk_env_sum_over_ranks(yk_env_h env, yk_idx_t v);/* :262 */
void yk_env_set_trace_enabled(yk_env_h env, int enable); /* yk_env::set_trace_enabled, :221 */
/* Multi-GPU: one process per GPU. The reference takes an MPI communicator (new_env(MPI_Comm), :136);
 * here the host states rank/size and installs a halo transport. */
int yk_env_set_ranks(yk_env_h env, int rank, int num_ranks);
/* PCI bus id of the device this process computes on ("0000:05:00.0"; the reference prints the host name of each rank,
 * setup.cpp:120-135): what a multi-device test asserts differs between the ranks.  Returns the length, < 0 on error. */
int yk_env_get_device_bus_id(yk_env_h env, char* out, int cap);
typedef struct {
    int peer;            /* neighbour rank */
    void* send_buf;      /* contiguous device buffers */
    void* recv_buf;
    size_t send_bytes, recv_bytes;
    int tag;
    int key;    /* which of the receiver's buffers, named alike on both ends: 0 = its packed buffer for this direction,
                 * (var ordinal * 16 + step slot) + 1 = the planes of that var slot (in-place x faces); plus 4096 x the
                 * solution's ordinal within its env */
} yk_halo_msg;
/* start: enqueue all transfers ordered after prior work on `stream` (a hipStream_t);
 * wait : make later work on `stream` see the received bytes. Return 0 on success. */
typedef int (*yk_exchange_fn)(void* user, int nmsgs, const yk_halo_msg* msgs, void* stream);
typedef int (*yk_allreduce_fn)(void* user, int op /*0 sum, 1 min, 2 max*/, long long* val);
int yk_env_set_transport(yk_env_h env, yk_exchange_fn start, yk_exchange_fn wait, yk_allreduce_fn allreduce, void* user);
/* Built-in transport: RCCL ncclSend/ncclRecv over xGMI. `unique_id` is the 128-byte ncclUniqueId made
 * by yk_rccl_get_unique_id() on rank 0 and distributed by the host (MPI_Bcast, torch.distributed ...). */
int yk_rccl_get_unique_id(void* unique_id_128);
int yk_env_init_rccl(yk_env_h env, const void* unique_id_128, int rank, int num_ranks);
/* What yk_factory::new_env() / new_env(MPI_Comm) do for a compiled host (setup.cpp:38-137) when the job was started
 * by a launcher instead of MPI_Init: reads RANK / WORLD_SIZE / LOCAL_RANK (torchrun), else OMPI_COMM_WORLD_* or
 * PMI_RANK / PMI_SIZE (mpirun), binds the process to GPU LOCAL_RANK, distributes the ncclUniqueId from rank 0 over
 * a TCP rendezvous on MASTER_ADDR : MASTER_PORT+1 and calls yk_env_init_rccl().  World size 1: returns 0, no-op.
 * YASK_HIP_TRANSPORT=tcp selects a host-staged TCP halo transport instead of RCCL (several ranks on ONE GPU: tests). */
int yk_env_init_from_launcher(yk_env_h env);
/* the host-staged TCP transport alone: full mesh of sockets; every rank listens on a kernel-assigned port, the port table
 * is gathered and handed out by rank 0 on `base_port` */
int yk_env_init_tcp(yk_env_h env, int rank, int num_ranks, const char* addr, int base_port);
/* Device-to-device halo transport between the ranks of ONE host (yask_amd/csrc/ykh_ipc.cpp): the sender copies its packed
 * halo straight into the receiver's buffer, mapped through HIP IPC memory handles (hipMemcpyAsync: SDMA over xGMI between
 * devices -- no compute units, unlike RCCL's send/recv kernels, so the copy proceeds while a stencil launch owns every CU);
 * ordering by flag words in device memory, everything stream-ordered.  The counterpart of the reference progressing its
 * MPI requests during the interior (adv_halo_exchange, src/kernel/lib/halo.cpp:494-574).  Ranks may share a device (tests).
 * Control messages and the scalar all-reduce run over the same TCP mesh as yk_env_init_tcp(). YASK_HIP_TRANSPORT=ipc. */
int yk_env_init_ipc(yk_env_h env, int rank, int num_ranks, const char* addr, int base_port);
/* Control-plane counters of the installed halo transport (only the IPC transport keeps them; the reference counts its MPI
 * traffic in yk_stats, context.hpp:319-328): out[0] buffer registrations sent over the TCP mesh, out[1] their bytes -- both
 * stay flat once every channel has been used once, i.e. the host is out of the loop of an exchange --, out[2] collective
 * begin calls (one 8-byte all-reduce per run_solution() / exchange_halos() call), out[3] collective resets, out[4] device
 * operations enqueued (flag kernels + copies), out[5] kind of mailbox memory (0 uncached device, 1 fine-grained device,
 * 2 plain device -- refused across devices --, 3 pinned host).  Returns how many values it wrote, 0 = no counters. */
int yk_env_get_transport_counters(yk_env_h env, long long* out, int cap);
/* Timing instrument, not a transport: this ONE process plays rank `rank` of `num_ranks`; what it sends to a neighbour comes back
 * as what it expects from that neighbour (a device-to-device copy on the communication stream; the halo DATA are therefore those
 * of a reflecting boundary).  Runs the full launch / pack / copy / unpack / wait schedule of a decomposed job's rank on one GPU:
 * tools/overlap_probe.py measures with it how much of the exchange each schedule hides. */
int yk_env_init_mirror(yk_env_h env, int rank, int num_ranks);
/* that mesh alone (no GPU needed): connect, one SUM all-reduce of the ranks over it, close; 0 on success */
int yk_tcp_mesh_check(int rank, int num_ranks, const char* addr, int base_port, long long* sum);
/* the rendezvous alone (no GPU needed): rank 0 serves `nbytes` of `buf` to the other ranks; 0 on success */
int yk_rendezvous_bcast(int rank, int num_ranks, const char* addr, int port, void* buf, size_t nbytes);
/* Executes the installed halo transport once on this rank with itself as the peer (a grouped self send/recv of
 * `nbytes` device bytes + one all-reduce) and verifies the bytes: 0 = the transport works.  With the RCCL transport
 * on one GPU (num_ranks 1) this is the only way its ncclSend/ncclRecv path runs outside a multi-GPU job. */
int yk_env_transport_loopback(yk_env_h env, size_t nbytes);
/* Achievable-bandwidth probe of THIS device, same streams as the solutions: 16-byte-per-lane streaming kernels over
 * `bytes` per array; kind 0 = copy (1 read : 1 write), 1 = the iso3dfd mix (3 reads : 1 write), 2 = read only.
 * Returns GB/s of (arrays touched x bytes) / time, best of `reps`; < 0 on error.  bench.py prints it next to the
 * roofline fraction so that a number can be read against what this box delivers right now. */
double yk_env_probe_bandwidth(yk_env_h env, int kind, size_t bytes, int reps);

/* ---- decomposition planning: pure index arithmetic, callable WITHOUT a GPU (used by multi-process
 * CPU tests and by hosts that want to size buffers before creating a solution).  Same code that
 * prepare_solution() runs.  Replaces: StencilContext::setup_rank (src/kernel/lib/setup.cpp:169-524),
 * get_compact_factors (src/common/tuple.cpp:355-430), halo-buffer geometry alloc_mpi_data
 * (src/kernel/lib/alloc.cpp:456-859). Domain dims are indexed 0..ndims-1 outer->inner (x,y,z). */
typedef struct {
    yk_idx_t global_size[3];      /* in/out: overall-domain size (0 = derive from local_size) */
    yk_idx_t local_size[3];       /* in/out: rank-domain size (0 = derive from global_size)   */
    yk_idx_t num_ranks[3];        /* in/out: rank grid (0 = choose the most compact one)      */
    yk_idx_t rank_index[3];       /* out: this rank's coordinates in the grid                 */
    yk_idx_t rank_offset[3];      /* out: global index of this rank's first domain point      */
    int num_neighbors;            /* out */
    int neighbor_rank[26];
    int neighbor_offset[26][3];   /* each in {-1,0,+1} */
} yk_rank_plan_t;
int yk_plan_rank(int ndims, int num_ranks, int rank, yk_rank_plan_t* plan);
typedef struct { yk_idx_t first[3], size[3]; } yk_box_t;   /* rank-local indices: 0 = first domain point */
/* Slab of a var (halo sizes, L1 norm as yk_var reports them) exchanged with the neighbour at `offset`:
 * sending != 0 -> the part of my domain the neighbour needs; else -> the part of my halo it fills.
 * Returns 1 and fills *box if there is one, 0 if nothing travels, <0 on error. */
int yk_plan_halo_slab(int ndims, const yk_rank_plan_t* plan, const int* neighbor_offset,
                      const yk_idx_t* halo_left, const yk_idx_t* halo_right, int l1_norm, int sending, yk_box_t* box);

/* Wave-front temporal tiling along the outermost domain dim (-Mbt / -bt; the reference's mega-block wave-fronts,
 * src/kernel/lib/context.cpp:482-745,1181-1525): the launches, in order, that apply `nphases` (step, stage) phases to
 * [lo, hi) slab by slab -- triples (phase, first, end) in out3.  Returns their number (<= cap are written).  No GPU needed. */
int yk_plan_wavefront(yk_idx_t lo, yk_idx_t hi, yk_idx_t width, yk_idx_t angle, yk_idx_t nphases, yk_idx_t* out3, int cap);

/* Planned launch of a decomposed rank: the reference computes a rank's exterior first, starts the halo
 * exchange and computes the interior meanwhile (StencilContext::run_solution, src/kernel/lib/context.cpp:377-478; exterior width
 * `-min_exterior`, settings.hpp:245).  Here that is ONE launch of the marching kernel over the rank box [0, n) whose workgroups
 * take (tile, x-range) descriptors in this order: x-face slabs, the y/z tiles a neighbour needs (in x-chunks short enough to be
 * done after shell_pct % of the launch), then the interior in pieces sized against the simulated CU time line.  flags bit 0 =
 * the block is part of the shell (it counts towards the signal that releases the exchange).  Writes <= cap descriptors, returns
 * their number; info[0..4] = shell blocks, simulated end of the shell, simulated end of the launch, the undivided box simulated
 * the same way (plane-iterations), mode used (3 rounds of equal blocks -- modes 0 and 3 --, 1 greedy budgets, 2 uniform chunks,
 * 4 halves).  No GPU needed. */
typedef struct { int x0, x1, y0, y1, z0, z1, flags, start; } yk_block_desc_t;
int yk_plan_blocks(const yk_idx_t* n3, const int* has_lo3, const int* has_hi3, const yk_idx_t* width3, int tile_y, int tile_z,
                   int overhead, int num_cus, int shell_pct, int mode, yk_block_desc_t* out, int cap, yk_idx_t* info5);

/* Pipelined half-exchanges (`-hip_halves`).  The reference progresses its halo messages while it computes
 * (adv_halo_exchange, src/kernel/lib/halo.cpp:494-574, called per micro-block, context.cpp:1037-1040); here a step of a decomposed
 * rank is TWO launches in regular order -- the outer x-half [0, q1) u [q2, nx) and the inner half [q1, q2) -- and each is followed
 * by the exchange of its own part of the faces, which travels while the other half is computed.  yk_plan_blocks(mode = 4) returns
 * the two launches' blocks (info5[0] = blocks of the first launch; no block signals);
 *   yk_plan_halves      the cut planes q1, q2 for a box of nx planes whose x neighbours need `xwidth` planes; returns 1, or 0
 *                       when the box is too short;
 *   yk_plan_halves_slab the x-ranges (first, size pairs in out4; returns their number, 0..2) of a halo slab [lo, lo + n) that
 *                       travel with half `half`; slabs of neighbours offset in x travel whole with half 0.   No GPU needed. */
int yk_plan_halves(yk_idx_t nx, yk_idx_t xwidth, yk_idx_t* q1, yk_idx_t* q2);
int yk_plan_halves_slab(int half, int x_neighbor, yk_idx_t lo, yk_idx_t n, yk_idx_t q1, yk_idx_t q2, yk_idx_t* out4);

/* ---- solution: replaces yk_solution, include/aux/yk_solution_api.hpp:82-1292 ---- */
const char* yk_solution_get_name(yk_soln_h s);                               /* :90 */
const char* yk_solution_get_description(yk_soln_h s);                        /* :98 */
const char* yk_solution_get_target(yk_soln_h s);                             /* :107 -> "cdna4_hip" */
int yk_solution_is_offloaded(yk_soln_h s);                                   /* :114 -> 1 */
int yk_solution_get_element_bytes(yk_soln_h s);                              /* :121 */
const char* yk_solution_get_step_dim_name(yk_soln_h s);                      /* :130 */
int yk_solution_get_num_domain_dims(yk_soln_h s);                            /* :140 */
const char* yk_solution_get_domain_dim_name(yk_soln_h s, int i);             /* get_domain_dim_names, :149 */
int yk_solution_get_num_misc_dims(yk_soln_h s);
const char* yk_solution_get_misc_dim_name(yk_soln_h s, int i);               /* get_misc_dim_names, :161 */
int yk_solution_set_rank_domain_size(yk_soln_h s, const char* dim, yk_idx_t size);     /* :187 */
yk_idx_t yk_solution_get_rank_domain_size(yk_soln_h s, const char* dim);               /* :221 */
int yk_solution_set_overall_domain_size(yk_soln_h s, const char* dim, yk_idx_t size);  /* :244 */
yk_idx_t yk_solution_get_overall_domain_size(yk_soln_h s, const char* dim);            /* :281 */
int yk_solution_set_block_size(yk_soln_h s, const char* dim, yk_idx_t size);           /* :316 */
yk_idx_t yk_solution_get_block_size(yk_soln_h s, const char* dim);                     /* :356 */
int yk_solution_set_num_ranks(yk_soln_h s, const char* dim, yk_idx_t n);               /* :404 */
yk_idx_t yk_solution_get_num_ranks(yk_soln_h s, const char* dim);                      /* :436 */
int yk_solution_set_rank_index(yk_soln_h s, const char* dim, yk_idx_t n);              /* :475 */
yk_idx_t yk_solution_get_rank_index(yk_soln_h s, const char* dim);                     /* :505 */
/* apply_command_line_options, :557: returns unrecognised tokens in `rem` (NUL terminated, truncated to rem_cap) */
int yk_solution_apply_command_line_options(yk_soln_h s, const char* args, char* rem, size_t rem_cap);
const char* yk_solution_get_command_line_help(yk_soln_h s);                  /* :586 */
const char* yk_solution_get_command_line_values(yk_soln_h s);                /* :596 */
int yk_solution_get_num_vars(yk_soln_h s);                                   /* :607 */
yk_var_h yk_solution_get_var(yk_soln_h s, const char* name);                 /* :616 */
yk_var_h yk_solution_get_var_by_index(yk_soln_h s, int i);                   /* get_vars, :624 */
int yk_solution_prepare(yk_soln_h s);                                        /* prepare_solution, :638 */
yk_idx_t yk_solution_get_first_rank_domain_index(yk_soln_h s, const char* dim);        /* :654 */
yk_idx_t yk_solution_get_last_rank_domain_index(yk_soln_h s, const char* dim);         /* :681 */
int yk_solution_run(yk_soln_h s, yk_idx_t first_step_index, yk_idx_t last_step_index); /* run_solution, :729/:759 */
int yk_solution_end(yk_soln_h s);                                            /* end_solution, :775 */
int yk_solution_exchange_halos(yk_soln_h s);                                 /* exchange_halos, :791 */
int yk_solution_copy_vars_to_device(yk_soln_h s);                            /* :799 (no-op: vars live on the device) */
int yk_solution_copy_vars_from_device(yk_soln_h s);                          /* :808 (no-op) */
typedef struct {                                                             /* yk_stats, :1300-1348 */
    yk_idx_t num_elements;       /* get_num_elements      */
    yk_idx_t num_steps_done;     /* get_num_steps_done    */
    yk_idx_t num_writes_done;    /* get_num_writes_done   */
    yk_idx_t est_fp_ops_done;    /* get_est_fp_ops_done   */
    double elapsed_secs;         /* get_elapsed_secs      */
    /* extensions */
    yk_idx_t num_reads_done;
    double halo_secs;
    double points_per_sec;       /* "throughput (num-points/sec)", soln_apis.cpp:455-461 */
    /* time breakdown of multi-rank runs, from HIP events on the compute / communication streams: the reference's
     * halo pack / unpack / wait and exterior / interior timers (context.hpp:319-328, printed by soln_apis.cpp:500-540) */
    double halo_pack_secs, halo_xfer_secs, halo_unpack_secs;
    double halo_wait_secs;       /* compute stream idle until the halos landed = communication NOT hidden by the interior */
    double exterior_secs, interior_secs;
    yk_idx_t halo_bytes_sent, halo_bytes_recv, halo_msgs_sent;   /* this rank, since the last get_stats() */
    yk_idx_t fused_passes;       /* launches that advanced TWO steps (ykh_starlin2.hpp); each counts as 2 of num_steps_done */
    yk_idx_t graph_replays;      /* hipGraphLaunch calls of captured step chains (-hip_step_graphs) ... */
    yk_idx_t graph_steps;        /* ... and the steps they advanced (part of num_steps_done) */
} yk_stats_t;
int yk_solution_get_stats(yk_soln_h s, yk_stats_t* out);                     /* get_stats, :819 (clears the counters) */
int yk_solution_clear_stats(yk_soln_h s);                                    /* clear_stats, :824 */
/* Per-step durations (ms, HIP events on the compute stream) of the last run_solution() call; recorded when the
 * option -hip_step_timers is on.  Returns the number of steps available (<= cap are copied). */
int yk_solution_get_step_times(yk_soln_h s, float* ms, int cap);
int yk_solution_reset_auto_tuner(yk_soln_h s, int enable, int verbose);      /* :838 */
int yk_solution_is_auto_tuner_enabled(yk_soln_h s);                          /* :855 */
int yk_solution_run_auto_tuner_now(yk_soln_h s, int verbose);                /* :880 */
int yk_solution_set_min_pad_size(yk_soln_h s, const char* dim, yk_idx_t n);  /* :927 */
yk_idx_t yk_solution_get_min_pad_size(yk_soln_h s, const char* dim);         /* :941 */
int yk_solution_set_step_wrap(yk_soln_h s, int do_wrap);                     /* set_step_wrap, :1212 */
int yk_solution_get_step_wrap(yk_soln_h s);                                  /* get_step_wrap, :1220 */
yk_var_h yk_solution_new_var(yk_soln_h s, const char* name, int ndims, const char* const* dims);      /* :994 */
yk_var_h yk_solution_new_fixed_size_var(yk_soln_h s, const char* name, int ndims, const char* const* dims,
                                        const yk_idx_t* sizes);                                      /* :1069 */
/* extensions used by harness/validation (the reference reaches into StencilContext for these,
 * src/kernel/yask_main.cpp:572-616) */
yk_idx_t yk_solution_compare_data(yk_soln_h s, yk_soln_h ref, double epsilon);   /* compare_data, context.cpp:1529 */
int yk_solution_set_streams(yk_soln_h s, void* compute_stream, void* comm_stream);
const char* yk_solution_get_kernel_variant(yk_soln_h s, int part);
int yk_solution_get_num_kernel_variants(yk_soln_h s, int part);
const char* yk_solution_get_kernel_variant_name(yk_soln_h s, int part, int i);
/* bytes of scratch per thread of variant i's kernel (> 0: the compiler spilled registers; such shapes are never
 * chosen by default or by the auto-tuner) */
yk_idx_t yk_solution_get_kernel_variant_scratch_bytes(yk_soln_h s, int part, int i);
/* Bounding box, inside this rank's domain, of part `part`'s sub-domain (IF_DOMAIN) condition, found by
 * prepare_solution() as the reference's find_bounding_box() does (src/kernel/lib/setup.cpp:1082-1169); the part is
 * only ever launched inside it.  first/last: 3 rank-local indices each (last inclusive).  Returns 1 if the part has
 * such a box, 0 if it is unconditional (first/last = the rank's domain), 2 if the condition holds nowhere in this
 * rank (first > last), -1 on error. */
int yk_solution_get_part_bounding_box(yk_soln_h s, int part, yk_idx_t* first, yk_idx_t* last);
/* The list of FULL boxes of part `part`'s sub-domain condition in this rank: non-overlapping rectangles that hold valid points
 * only and together all of them -- the reference's StencilPartBase::_bb_list (src/kernel/lib/setup.cpp:1235-1500), which its
 * kernels walk box by box.  prepare_solution() finds it (two device reductions, no per-point mask) where the condition does not
 * fill its bounding box but is a handful of slabs; the part then runs its unpredicated kernels on each box.  Fills up to `cap`
 * boxes (3 rank-local indices each in first / last, last inclusive) and returns their number: 0 = no list (unconditional part,
 * a condition that fills its bounding box, or one left to the point kernel's per-point predicate), -1 on error. */
int yk_solution_get_part_full_boxes(yk_soln_h s, int part, int cap, yk_idx_t* first, yk_idx_t* last);
/* Scratch vars on chip.  The reference evaluates scratch vars per micro-block into per-thread arrays that stay in cache
 * (src/kernel/lib/stencil_calc.cpp:40-289); here a 2-D solution's run of scratch stages and the stage they feed can run as ONE
 * kernel per step with the scratch vars in the LDS of a workgroup's tile (csrc/ykh_fused.hpp).  Returns the number of such fused
 * groups the prepared solution is using (0: every part is a sweep of its own -- not legal for this solution, switched off with
 * YASK_HIP_FUSE_SCRATCH=0, or slower in prepare_solution()'s timing); for group g < that number, info[0..5] (when non-null) =
 * parts in it, levels-independent scratch vars, LDS slots they share, LDS bytes, tile rows, tile columns.  -1 on error. */
int yk_solution_get_fused_groups(yk_soln_h s, int g, long long* info);
/* Work of one stencil part, per step on this rank -- the facts the reference prints per part and stage in
 * Stage::init_work_stats (src/kernel/lib/stencil_calc.cpp:461-598: points to eval, reads / writes / est FP-ops per point, the
 * input / output var lists), plus what a bandwidth-bound GPU kernel is measured against: the COMPULSORY HBM bytes per point =
 * (distinct arrays read + arrays written) x element size, where an array is one (var, step offset, misc indices) access group of
 * a var that spans every domain dim of the solution (lower-dimensional vars -- coefficient lines, sponge profiles -- stay in
 * cache and count 0) and scratch vars are left out (a cost of this implementation, reported separately).  tools/generic_table.py. */
typedef struct {
    const char* name;
    int stage, is_scratch, has_condition;
    int fp_ops, points_read, points_written;          /* per point, as the reference reports them */
    int arrays_read, arrays_written;                  /* full-dimensional, non-scratch */
    int scratch_arrays_read, scratch_arrays_written;
    yk_idx_t points;                                  /* points one step evaluates (sub-domain box clipped to this rank, x outer dim) */
    double compulsory_bytes_per_point;
} yk_part_info_t;
int yk_solution_get_part_info(yk_soln_h s, int part, yk_part_info_t* out);
/* Launch part `part` of step t once with variant i (or the selected one if i < 0) on the compute
 * stream, bracketed by HIP events; returns the kernel duration in ms in *ms. Used by bench.py. */
int yk_solution_time_part(yk_soln_h s, int part, int variant, yk_idx_t xchunk, yk_idx_t t, int reps, float* ms);
/* the same over a sub-box of this rank's domain (rank-local first/last, last inclusive): what an exterior slab of a
 * decomposed run costs with a given kernel shape (tools/slab_kernels.py) */
int yk_solution_time_part_box(yk_soln_h s, int part, int variant, yk_idx_t xchunk, const yk_idx_t* first, const yk_idx_t* last,
                              yk_idx_t t, int reps, float* ms);

/* Compute side of one step as a rank with neighbours on the given sides (has_lo / has_hi per domain dim) would run it --
 * exterior slabs, then the split interior, no communication -- and the undivided box: ms3 = {exterior, interior, whole}.
 * Measured on ONE GPU (tools/decomp_cost.py): what the exterior-first schedule costs before any byte travels. */
int yk_solution_time_decomposed_step(yk_soln_h s, const int* has_lo3, const int* has_hi3, int reps, float* ms3);

/* ---- var: replaces yk_var, include/aux/yk_var_api.hpp:185-1490 ---- */
const char* yk_var_get_name(yk_var_h v);                                     /* :195 */
int yk_var_get_num_dims(yk_var_h v);                                         /* :204 */
const char* yk_var_get_dim_name(yk_var_h v, int i);                          /* get_dim_names, :212 */
int yk_var_is_dim_used(yk_var_h v, const char* dim);                         /* :230 */
int yk_var_is_fixed_size(yk_var_h v);                                        /* :237 */
yk_idx_t yk_var_get_first_local_index(yk_var_h v, const char* dim);          /* :253 */
yk_idx_t yk_var_get_last_local_index(yk_var_h v, const char* dim);           /* :275 */
yk_idx_t yk_var_get_alloc_size(yk_var_h v, const char* dim);                 /* :294 */
yk_idx_t yk_var_get_first_valid_step_index(yk_var_h v);                      /* :322 */
yk_idx_t yk_var_get_last_valid_step_index(yk_var_h v);                       /* :335 */
yk_idx_t yk_var_get_rank_domain_size(yk_var_h v, const char* dim);           /* :350 */
yk_idx_t yk_var_get_first_rank_domain_index(yk_var_h v, const char* dim);    /* :371 */
yk_idx_t yk_var_get_last_rank_domain_index(yk_var_h v, const char* dim);     /* :393 */
yk_idx_t yk_var_get_left_halo_size(yk_var_h v, const char* dim);             /* :410 */
yk_idx_t yk_var_get_right_halo_size(yk_var_h v, const char* dim);            /* :421 */
yk_idx_t yk_var_get_first_rank_halo_index(yk_var_h v, const char* dim);      /* :434 */
yk_idx_t yk_var_get_last_rank_halo_index(yk_var_h v, const char* dim);       /* :455 */
yk_idx_t yk_var_get_left_pad_size(yk_var_h v, const char* dim);              /* :474 */
yk_idx_t yk_var_get_right_pad_size(yk_var_h v, const char* dim);             /* :486 */
yk_idx_t yk_var_get_left_extra_pad_size(yk_var_h v, const char* dim);        /* :500 */
yk_idx_t yk_var_get_right_extra_pad_size(yk_var_h v, const char* dim);       /* :514 */
yk_idx_t yk_var_get_first_misc_index(yk_var_h v, const char* dim);           /* :526 */
yk_idx_t yk_var_get_last_misc_index(yk_var_h v, const char* dim);            /* :538 */
int yk_var_set_left_min_pad_size(yk_var_h v, const char* dim, yk_idx_t n);   /* :1163 */
int yk_var_set_right_min_pad_size(yk_var_h v, const char* dim, yk_idx_t n);  /* :1187 */
int yk_var_set_min_pad_size(yk_var_h v, const char* dim, yk_idx_t n);        /* :1197 */
int yk_var_set_left_halo_size(yk_var_h v, const char* dim, yk_idx_t n);      /* :1213 */
int yk_var_set_right_halo_size(yk_var_h v, const char* dim, yk_idx_t n);     /* :1229 */
int yk_var_set_halo_size(yk_var_h v, const char* dim, yk_idx_t n);           /* :1240 */
int yk_var_set_first_misc_index(yk_var_h v, const char* dim, yk_idx_t idx);  /* :1286 */
int yk_var_set_alloc_size(yk_var_h v, const char* dim, yk_idx_t n);          /* :1264 (step / misc dims) */
int yk_var_are_indices_local(yk_var_h v, const yk_idx_t* indices);           /* :552 */
double yk_var_get_element(yk_var_h v, const yk_idx_t* indices);              /* :577 */
yk_idx_t yk_var_set_element(yk_var_h v, double val, const yk_idx_t* indices, int strict_indices);      /* :604 */
yk_idx_t yk_var_add_to_element(yk_var_h v, double val, const yk_idx_t* indices, int strict_indices);   /* :772 */
yk_idx_t yk_var_get_elements_in_slice_f32(yk_var_h v, float* buf, size_t buf_elems,
                                          const yk_idx_t* first, const yk_idx_t* last);                /* :699 */
yk_idx_t yk_var_get_elements_in_slice_f64(yk_var_h v, double* buf, size_t buf_elems,
                                          const yk_idx_t* first, const yk_idx_t* last);                /* :727 */
yk_idx_t yk_var_set_elements_in_slice_f32(yk_var_h v, const float* buf, size_t buf_elems,
                                          const yk_idx_t* first, const yk_idx_t* last);                /* :876 */
yk_idx_t yk_var_set_elements_in_slice_f64(yk_var_h v, const double* buf, size_t buf_elems,
                                          const yk_idx_t* first, const yk_idx_t* last);                /* :905 */
yk_idx_t yk_var_set_elements_in_slice_same(yk_var_h v, double val, const yk_idx_t* first,
                                           const yk_idx_t* last, int strict_indices);                  /* :820 */
/* set_elements_in_slice(source var, first source, first target, last target), yk_var_api.hpp:936 */
yk_idx_t yk_var_set_elements_in_slice_from_var(yk_var_h v, yk_var_h source, const yk_idx_t* first_source,
                                               const yk_idx_t* first_target, const yk_idx_t* last_target);
int yk_var_set_all_elements_same(yk_var_h v, double val);                                             /* :800 */
typedef struct {                                                             /* yk_var::yk_reduction_result, :960-1030 */
    int reduction_mask;
    yk_idx_t num_elements_reduced;
    double sum, sum_squares, product, max, min;
} yk_reduction_t;
/* mask bits as yk_var::yk_reduction_mask: 1 sum, 2 sum of squares, 4 product, 8 max, 16 min */
int yk_var_reduce_elements_in_slice(yk_var_h v, int mask, const yk_idx_t* first, const yk_idx_t* last,
                                    int strict_indices, yk_reduction_t* out);                         /* :1040 */
int yk_var_get_halo_exchange_l1_norm(yk_var_h v);                            /* :1101 */
int yk_var_set_halo_exchange_l1_norm(yk_var_h v, int norm);                  /* :1118 */
int yk_var_is_dynamic_step_alloc(yk_var_h v);                                /* :1131 */
int yk_var_is_storage_allocated(yk_var_h v);                                 /* :1316 */
yk_idx_t yk_var_get_num_storage_bytes(yk_var_h v);                           /* :1325 */
yk_idx_t yk_var_get_num_storage_elements(yk_var_h v);                        /* :1333 */
int yk_var_alloc_storage(yk_var_h v);                                        /* :1342 */
int yk_var_release_storage(yk_var_h v);                                      /* :1352 */
int yk_var_is_storage_layout_identical(yk_var_h v, yk_var_h other);          /* :1363 */
int yk_var_fuse_vars(yk_var_h v, yk_var_h source);                           /* :1395 */
/* :1437.  The storage is in HBM: the pointer returned is a host copy in the same layout which the library keeps coherent
 * from this call on -- copied to the device before every API call that uses the var (the caller may have written through
 * the pointer), copied back after every call that changes it (run_solution, set_*, exchange_halos) -- until the storage is
 * released.  No extra call is needed for correctness; the two extensions below only trade the copies away. */
void* yk_var_get_raw_storage_buffer(yk_var_h v);
int yk_var_sync_raw_storage_to_device(yk_var_h v);                           /* extension: push the host copy now */
int yk_var_release_raw_storage_buffer(yk_var_h v);                           /* extension: push, then stop the coherency copies (pointer invalid) */
/* Var placement: prepare_solution() draws several sets of var allocations, times a step on each and keeps the fastest
 * (option -hip_placement_trials).  ms[i] = ms per step on set i of the last prepare_solution(), *chosen =
 * the set kept.  Returns the number of sets timed (0: no search -- small solution, or vars that already held data). */
int yk_solution_get_placement_trials(yk_soln_h s, int* chosen, float* ms, int cap);
void* yk_var_get_device_storage(yk_var_h v);                                 /* device pointer of the allocation */
/* extension: fill domain+halo of every step slot with offset + scale*H(hash_id, slot, x, y, z), the
 * layout-independent logical-index hash shared with the oracle (oracle/stencil_oracle.c). */
int yk_var_set_elements_hash(yk_var_h v, double offset, double scale, int hash_id);

#ifdef __cplusplus
}
#endif
#endif /* YASK_HIP_C_API_H */
