"""ctypes binding of the C ABI in include/yask_hip_c_api.h.

Loads `yask_amd/lib/libyask_kernel.<stencil>.cdna4_hip.so` (built in-tree by yask_amd/csrc/Makefile)
and declares the prototype of every exported entry point.  There is deliberately no fallback: if the
library is missing the import of a solution fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path

_PKG = Path(__file__).resolve().parent
# YASK_HIP_LIB_DIR: another directory of kernel libraries (e.g. a `make YKH_PROFILING=1 LIBDIR=...` build with the sweep shapes)
LIB_DIR = Path(os.environ["YASK_HIP_LIB_DIR"]) if os.environ.get("YASK_HIP_LIB_DIR") else _PKG / "lib"
HEADER = _PKG.parent / "include" / "yask_hip_c_api.h"

idx_t = C.c_int64


class YkStats(C.Structure):
    _fields_ = [("num_elements", idx_t), ("num_steps_done", idx_t), ("num_writes_done", idx_t),
                ("est_fp_ops_done", idx_t), ("elapsed_secs", C.c_double), ("num_reads_done", idx_t),
                ("halo_secs", C.c_double), ("points_per_sec", C.c_double),
                ("halo_pack_secs", C.c_double), ("halo_xfer_secs", C.c_double), ("halo_unpack_secs", C.c_double),
                ("halo_wait_secs", C.c_double), ("exterior_secs", C.c_double), ("interior_secs", C.c_double),
                ("halo_bytes_sent", idx_t), ("halo_bytes_recv", idx_t), ("halo_msgs_sent", idx_t),
                ("fused_passes", idx_t), ("graph_replays", idx_t), ("graph_steps", idx_t)]


class YkReduction(C.Structure):
    _fields_ = [("reduction_mask", C.c_int), ("num_elements_reduced", idx_t), ("sum", C.c_double),
                ("sum_squares", C.c_double), ("product", C.c_double), ("max", C.c_double), ("min", C.c_double)]


class YkHaloMsg(C.Structure):
    _fields_ = [("peer", C.c_int), ("send_buf", C.c_void_p), ("recv_buf", C.c_void_p),
                ("send_bytes", C.c_size_t), ("recv_bytes", C.c_size_t), ("tag", C.c_int), ("key", C.c_int)]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(YkHaloMsg), C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_longlong))

_H = C.c_void_p          # opaque handles
_S = C.c_char_p
_IP = C.POINTER(idx_t)

# name -> (restype, argtypes); kept in the same order as the header.
class BlockDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("x0", "x1", "y0", "y1", "z0", "z1", "flags", "start")]


class PartInfo(C.Structure):
    _fields_ = [("name", C.c_char_p), ("stage", C.c_int), ("is_scratch", C.c_int), ("has_condition", C.c_int),
                ("fp_ops", C.c_int), ("points_read", C.c_int), ("points_written", C.c_int),
                ("arrays_read", C.c_int), ("arrays_written", C.c_int), ("scratch_arrays_read", C.c_int), ("scratch_arrays_written", C.c_int),
                ("points", C.c_longlong), ("compulsory_bytes_per_point", C.c_double)]


class RankPlan(C.Structure):
    """yk_rank_plan_t"""
    _fields_ = [("global_size", idx_t * 3), ("local_size", idx_t * 3), ("num_ranks", idx_t * 3), ("rank_index", idx_t * 3),
                ("rank_offset", idx_t * 3), ("num_neighbors", C.c_int), ("neighbor_rank", C.c_int * 26),
                ("neighbor_offset", (C.c_int * 3) * 26)]


class Box(C.Structure):
    """yk_box_t"""
    _fields_ = [("first", idx_t * 3), ("size", idx_t * 3)]


PROTOTYPES = {
    "yk_solution_get_kernel_variant_scratch_bytes": (idx_t, [_H, C.c_int, C.c_int]),
    "yk_solution_get_part_bounding_box": (C.c_int, [_H, C.c_int, C.POINTER(idx_t), C.POINTER(idx_t)]),
    "yk_solution_get_part_full_boxes": (C.c_int, [_H, C.c_int, C.c_int, C.POINTER(idx_t), C.POINTER(idx_t)]),
    "yk_solution_get_fused_groups": (C.c_int, [_H, C.c_int, C.POINTER(C.c_longlong)]),
    "yk_solution_clear_stats": (C.c_int, [_H]),
    "yk_solution_get_step_times": (C.c_int, [_H, C.POINTER(C.c_float), C.c_int]),
    "yk_solution_get_placement_trials": (C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int]),
    "yk_env_init_from_launcher": (C.c_int, [_H]),
    "yk_env_init_tcp": (C.c_int, [_H, C.c_int, C.c_int, _S, C.c_int]),
    "yk_env_init_ipc": (C.c_int, [_H, C.c_int, C.c_int, _S, C.c_int]),
    "yk_env_init_mirror": (C.c_int, [_H, C.c_int, C.c_int]),
    "yk_rendezvous_bcast": (C.c_int, [C.c_int, C.c_int, _S, C.c_int, C.c_void_p, C.c_size_t]),
    "yk_tcp_mesh_check": (C.c_int, [C.c_int, C.c_int, _S, C.c_int, C.POINTER(C.c_longlong)]),
    "yk_env_transport_loopback": (C.c_int, [_H, C.c_size_t]),
    "yk_env_get_transport_counters": (C.c_int, [_H, C.POINTER(C.c_longlong), C.c_int]),
    "yk_env_probe_bandwidth": (C.c_double, [_H, C.c_int, C.c_size_t, C.c_int]),
    "yk_solution_set_min_pad_size": (C.c_int, [_H, _S, idx_t]),
    "yk_solution_get_min_pad_size": (idx_t, [_H, _S]),
    "yk_solution_set_step_wrap": (C.c_int, [_H, C.c_int]),
    "yk_solution_get_step_wrap": (C.c_int, [_H]),
    "yk_var_set_elements_in_slice_from_var": (idx_t, [_H, _H, C.POINTER(idx_t), C.POINTER(idx_t), C.POINTER(idx_t)]),
    "yk_plan_wavefront": (C.c_int, [idx_t, idx_t, idx_t, idx_t, idx_t, C.POINTER(idx_t), C.c_int]),
    "yk_plan_rank": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(RankPlan)]),
    "yk_plan_blocks": (C.c_int, [C.POINTER(idx_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(idx_t), C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.POINTER(BlockDesc), C.c_int, C.POINTER(idx_t)]),
    "yk_plan_halves": (C.c_int, [idx_t, idx_t, C.POINTER(idx_t), C.POINTER(idx_t)]),
    "yk_plan_halves_slab": (C.c_int, [C.c_int, C.c_int, idx_t, idx_t, idx_t, idx_t, C.POINTER(idx_t)]),
    "yk_plan_halo_slab": (C.c_int, [C.c_int, C.POINTER(RankPlan), C.POINTER(C.c_int), C.POINTER(idx_t), C.POINTER(idx_t),
                                    C.c_int, C.c_int, C.POINTER(Box)]),
    "yk_last_error": (_S, []),
    "yk_last_error_code": (C.c_int, []),
    "yk_clear_error": (None, []),
    "yk_get_version_string": (_S, []),
    "yk_new_env": (_H, []),
    "yk_free_env": (None, [_H]),
    "yk_new_solution": (_H, [_H]),
    "yk_new_solution_from": (_H, [_H, _H]),
    "yk_free_solution": (None, [_H]),
    "yk_env_get_num_ranks": (C.c_int, [_H]),
    "yk_env_get_rank_index": (C.c_int, [_H]),
    "yk_env_global_barrier": (C.c_int, [_H]),
    "yk_env_sum_over_ranks": (idx_t, [_H, idx_t]),
    "yk_env_set_trace_enabled": (None, [_H, C.c_int]),
    "yk_env_set_ranks": (C.c_int, [_H, C.c_int, C.c_int]),
    "yk_env_get_device_bus_id": (C.c_int, [_H, C.c_char_p, C.c_int]),
    "yk_env_set_transport": (C.c_int, [_H, EXCHANGE_FN, EXCHANGE_FN, ALLREDUCE_FN, C.c_void_p]),
    "yk_rccl_get_unique_id": (C.c_int, [C.c_void_p]),
    "yk_env_init_rccl": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    "yk_solution_get_name": (_S, [_H]),
    "yk_solution_get_description": (_S, [_H]),
    "yk_solution_get_target": (_S, [_H]),
    "yk_solution_is_offloaded": (C.c_int, [_H]),
    "yk_solution_get_element_bytes": (C.c_int, [_H]),
    "yk_solution_get_step_dim_name": (_S, [_H]),
    "yk_solution_get_num_domain_dims": (C.c_int, [_H]),
    "yk_solution_get_domain_dim_name": (_S, [_H, C.c_int]),
    "yk_solution_get_num_misc_dims": (C.c_int, [_H]),
    "yk_solution_get_misc_dim_name": (_S, [_H, C.c_int]),
    "yk_solution_set_rank_domain_size": (C.c_int, [_H, _S, idx_t]),
    "yk_solution_get_rank_domain_size": (idx_t, [_H, _S]),
    "yk_solution_set_overall_domain_size": (C.c_int, [_H, _S, idx_t]),
    "yk_solution_get_overall_domain_size": (idx_t, [_H, _S]),
    "yk_solution_set_block_size": (C.c_int, [_H, _S, idx_t]),
    "yk_solution_get_block_size": (idx_t, [_H, _S]),
    "yk_solution_set_num_ranks": (C.c_int, [_H, _S, idx_t]),
    "yk_solution_get_num_ranks": (idx_t, [_H, _S]),
    "yk_solution_set_rank_index": (C.c_int, [_H, _S, idx_t]),
    "yk_solution_get_rank_index": (idx_t, [_H, _S]),
    "yk_solution_apply_command_line_options": (C.c_int, [_H, _S, C.c_char_p, C.c_size_t]),
    "yk_solution_get_command_line_help": (_S, [_H]),
    "yk_solution_get_command_line_values": (_S, [_H]),
    "yk_solution_get_num_vars": (C.c_int, [_H]),
    "yk_solution_get_var": (_H, [_H, _S]),
    "yk_solution_get_var_by_index": (_H, [_H, C.c_int]),
    "yk_solution_prepare": (C.c_int, [_H]),
    "yk_solution_get_first_rank_domain_index": (idx_t, [_H, _S]),
    "yk_solution_get_last_rank_domain_index": (idx_t, [_H, _S]),
    "yk_solution_run": (C.c_int, [_H, idx_t, idx_t]),
    "yk_solution_end": (C.c_int, [_H]),
    "yk_solution_exchange_halos": (C.c_int, [_H]),
    "yk_solution_copy_vars_to_device": (C.c_int, [_H]),
    "yk_solution_copy_vars_from_device": (C.c_int, [_H]),
    "yk_solution_get_stats": (C.c_int, [_H, C.POINTER(YkStats)]),
    "yk_solution_reset_auto_tuner": (C.c_int, [_H, C.c_int, C.c_int]),
    "yk_solution_is_auto_tuner_enabled": (C.c_int, [_H]),
    "yk_solution_run_auto_tuner_now": (C.c_int, [_H, C.c_int]),
    "yk_solution_new_var": (_H, [_H, _S, C.c_int, C.POINTER(C.c_char_p)]),
    "yk_solution_new_fixed_size_var": (_H, [_H, _S, C.c_int, C.POINTER(C.c_char_p), _IP]),
    "yk_solution_compare_data": (idx_t, [_H, _H, C.c_double]),
    "yk_solution_set_streams": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "yk_solution_get_kernel_variant": (_S, [_H, C.c_int]),
    "yk_solution_get_num_kernel_variants": (C.c_int, [_H, C.c_int]),
    "yk_solution_get_kernel_variant_name": (_S, [_H, C.c_int, C.c_int]),
    "yk_solution_get_part_info": (C.c_int, [_H, C.c_int, C.POINTER(PartInfo)]),
    "yk_solution_time_part": (C.c_int, [_H, C.c_int, C.c_int, idx_t, idx_t, C.c_int, C.POINTER(C.c_float)]),
    "yk_solution_time_decomposed_step": (C.c_int, [_H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_float)]),
    "yk_solution_time_part_box": (C.c_int, [_H, C.c_int, C.c_int, idx_t, _IP, _IP, idx_t, C.c_int, C.POINTER(C.c_float)]),
    "yk_var_get_name": (_S, [_H]),
    "yk_var_get_num_dims": (C.c_int, [_H]),
    "yk_var_get_dim_name": (_S, [_H, C.c_int]),
    "yk_var_is_dim_used": (C.c_int, [_H, _S]),
    "yk_var_is_fixed_size": (C.c_int, [_H]),
    "yk_var_get_first_local_index": (idx_t, [_H, _S]),
    "yk_var_get_last_local_index": (idx_t, [_H, _S]),
    "yk_var_get_alloc_size": (idx_t, [_H, _S]),
    "yk_var_get_first_valid_step_index": (idx_t, [_H]),
    "yk_var_get_last_valid_step_index": (idx_t, [_H]),
    "yk_var_get_rank_domain_size": (idx_t, [_H, _S]),
    "yk_var_get_first_rank_domain_index": (idx_t, [_H, _S]),
    "yk_var_get_last_rank_domain_index": (idx_t, [_H, _S]),
    "yk_var_get_left_halo_size": (idx_t, [_H, _S]),
    "yk_var_get_right_halo_size": (idx_t, [_H, _S]),
    "yk_var_get_first_rank_halo_index": (idx_t, [_H, _S]),
    "yk_var_get_last_rank_halo_index": (idx_t, [_H, _S]),
    "yk_var_get_left_pad_size": (idx_t, [_H, _S]),
    "yk_var_get_right_pad_size": (idx_t, [_H, _S]),
    "yk_var_get_left_extra_pad_size": (idx_t, [_H, _S]),
    "yk_var_get_right_extra_pad_size": (idx_t, [_H, _S]),
    "yk_var_get_first_misc_index": (idx_t, [_H, _S]),
    "yk_var_get_last_misc_index": (idx_t, [_H, _S]),
    "yk_var_set_left_min_pad_size": (C.c_int, [_H, _S, idx_t]),
    "yk_var_set_right_min_pad_size": (C.c_int, [_H, _S, idx_t]),
    "yk_var_set_min_pad_size": (C.c_int, [_H, _S, idx_t]),
    "yk_var_set_left_halo_size": (C.c_int, [_H, _S, idx_t]),
    "yk_var_set_right_halo_size": (C.c_int, [_H, _S, idx_t]),
    "yk_var_set_halo_size": (C.c_int, [_H, _S, idx_t]),
    "yk_var_set_first_misc_index": (C.c_int, [_H, _S, idx_t]),
    "yk_var_set_alloc_size": (C.c_int, [_H, _S, idx_t]),
    "yk_var_are_indices_local": (C.c_int, [_H, _IP]),
    "yk_var_get_element": (C.c_double, [_H, _IP]),
    "yk_var_set_element": (idx_t, [_H, C.c_double, _IP, C.c_int]),
    "yk_var_add_to_element": (idx_t, [_H, C.c_double, _IP, C.c_int]),
    "yk_var_get_elements_in_slice_f32": (idx_t, [_H, C.c_void_p, C.c_size_t, _IP, _IP]),
    "yk_var_get_elements_in_slice_f64": (idx_t, [_H, C.c_void_p, C.c_size_t, _IP, _IP]),
    "yk_var_set_elements_in_slice_f32": (idx_t, [_H, C.c_void_p, C.c_size_t, _IP, _IP]),
    "yk_var_set_elements_in_slice_f64": (idx_t, [_H, C.c_void_p, C.c_size_t, _IP, _IP]),
    "yk_var_set_elements_in_slice_same": (idx_t, [_H, C.c_double, _IP, _IP, C.c_int]),
    "yk_var_set_all_elements_same": (C.c_int, [_H, C.c_double]),
    "yk_var_reduce_elements_in_slice": (C.c_int, [_H, C.c_int, _IP, _IP, C.c_int, C.POINTER(YkReduction)]),
    "yk_var_get_halo_exchange_l1_norm": (C.c_int, [_H]),
    "yk_var_set_halo_exchange_l1_norm": (C.c_int, [_H, C.c_int]),
    "yk_var_is_dynamic_step_alloc": (C.c_int, [_H]),
    "yk_var_is_storage_allocated": (C.c_int, [_H]),
    "yk_var_get_num_storage_bytes": (idx_t, [_H]),
    "yk_var_get_num_storage_elements": (idx_t, [_H]),
    "yk_var_alloc_storage": (C.c_int, [_H]),
    "yk_var_release_storage": (C.c_int, [_H]),
    "yk_var_is_storage_layout_identical": (C.c_int, [_H, _H]),
    "yk_var_fuse_vars": (C.c_int, [_H, _H]),
    "yk_var_get_raw_storage_buffer": (C.c_void_p, [_H]),
    "yk_var_sync_raw_storage_to_device": (C.c_int, [_H]),
    "yk_var_release_raw_storage_buffer": (C.c_int, [_H]),
    "yk_var_get_device_storage": (C.c_void_p, [_H]),
    "yk_var_set_elements_hash": (C.c_int, [_H, C.c_double, C.c_double, C.c_int]),
}


def header_symbols() -> list[str]:
    """Names of all functions declared in include/yask_hip_c_api.h."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(yk_[a-z0-9_]+)\s*\(", txt)) - {"yk_exchange_fn", "yk_allreduce_fn"})


def lib_path(stencil: str) -> Path:
    return LIB_DIR / f"libyask_kernel.{stencil}.cdna4_hip.so"


_loaded: dict[str, C.CDLL] = {}


def ensure_built(stencils=("iso3dfd",)) -> None:
    """Entry points that may start on a box holding only the sources (tests' `gpu` fixture, smoke(), bench.py) call
    this first: if a kernel library is missing it is compiled in-tree with hipcc (minutes), exactly as
    __graft_entry__.build() does.  This builds the HIP library -- it is not a fallback around it; load() stays strict."""
    import shutil
    import subprocess
    import sys
    missing = [s for s in stencils if not lib_path(s).exists()]
    if not missing:
        return
    if not shutil.which("hipcc") and not Path("/opt/rocm/bin/hipcc").exists():
        return                       # load() will raise with the build instructions
    print(f"yask_amd: building kernel libraries (missing: {', '.join(missing)}) ...", file=sys.stderr, flush=True)
    env = dict(os.environ)
    env["PATH"] = env.get("PATH", "") + ":/opt/rocm/bin"
    subprocess.check_call(["make", "-C", str(_PKG / "csrc"), f"-j{min(16, os.cpu_count() or 4)}"], env=env,
                          stdout=subprocess.DEVNULL)


def load(stencil: str) -> C.CDLL:
    """dlopen the stencil's kernel library and attach prototypes. Raises if it is not built."""
    if stencil in _loaded:
        return _loaded[stencil]
    p = lib_path(stencil)
    if not p.exists():
        raise ImportError(
            f"{p} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` or "
            f"`make -C yask_amd/csrc STENCILS={stencil}` (hipcc, --offload-arch=gfx950). "
            "There is no CPU fallback for the cdna4_hip kernel library.")
    lib = C.CDLL(str(p), mode=getattr(os, "RTLD_LOCAL", 0))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)      # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _loaded[stencil] = lib
    return lib
