"""Python face of the cdna4_hip YASK kernel library.

Mirrors the classes and method names of the reference's SWIG module `yask_kernel`
(src/kernel/swig/yask_kernel_api.i over include/yask_kernel_api.hpp, include/aux/yk_solution_api.hpp,
include/aux/yk_var_api.hpp): `yk_factory`, `yk_env`, `yk_solution`, `yk_var`, `yk_stats`.  Errors raise
`RuntimeError` whose text starts with "YASK error: ", exactly as the SWIG wrapper maps
yask::yask_exception (yask_kernel_api.i:75-82).  Every call goes through the C ABI
(include/yask_hip_c_api.h) of libyask_kernel.<stencil>.cdna4_hip.so; numpy arrays stand in for the
SWIG `pybuffer` slices.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _capi

idx_t = _capi.idx_t


def _b(s):
    return s.encode() if isinstance(s, str) else s


class _Lib:
    """A loaded kernel library + error translation."""

    def __init__(self, stencil):
        self.stencil = stencil
        self.c = _capi.load(stencil)

    def check(self):
        if self.c.yk_last_error_code():
            msg = self.c.yk_last_error().decode()
            self.c.yk_clear_error()
            raise RuntimeError(msg)

    def call(self, name, *args):
        r = getattr(self.c, name)(*args)
        self.check()
        return r

    def call_rc(self, name, *args):
        rc = getattr(self.c, name)(*args)
        if rc != 0:
            self.check()
            raise RuntimeError("YASK error: %s failed" % name)
        return rc


class yask_output:
    """yask_output and its four kinds (include/yask_common_api.hpp:225-275): where the library's debug output goes."""

    def __init__(self, kind, name=None):
        self._kind, self._name, self._buf, self._fh = kind, name, [], None
        if kind == "file":
            try:
                self._fh = open(name, "w")       # created (truncated) at construction, like yask_file_output
            except OSError as e:
                raise RuntimeError("YASK error: cannot open '%s' for output: %s" % (name, e))

    def write(self, text):
        if self._kind == "file":
            self._fh.write(text); self._fh.flush()
        elif self._kind == "string":
            self._buf.append(text)
        elif self._kind == "stdout":
            import sys
            sys.stdout.write(text)
        return self

    def get_filename(self): return self._name or ""
    def get_string(self): return "".join(self._buf)
    def discard(self): self._buf = []
    def close(self):
        if self._fh:
            self._fh.close(); self._fh = None


class yask_output_factory:
    """yask_output_factory (include/yask_common_api.hpp:184-232)."""
    def new_file_output(self, file_name): return yask_output("file", file_name)
    def new_string_output(self): return yask_output("string")
    def new_stdout_output(self): return yask_output("stdout")
    def new_null_output(self): return yask_output("null")


class yk_stats:
    """yk_stats (include/aux/yk_solution_api.hpp:1300-1348)."""

    def __init__(self, st):
        self._st = st

    def get_num_elements(self): return int(self._st.num_elements)
    def get_num_steps_done(self): return int(self._st.num_steps_done)
    def get_num_writes_done(self): return int(self._st.num_writes_done)
    def get_est_fp_ops_done(self): return int(self._st.est_fp_ops_done)
    def get_elapsed_secs(self): return float(self._st.elapsed_secs)
    # extensions
    def get_num_reads_done(self): return int(self._st.num_reads_done)
    def get_halo_secs(self): return float(self._st.halo_secs)
    def get_points_per_sec(self): return float(self._st.points_per_sec)
    # time breakdown of multi-rank runs (the reference prints these in get_stats(), soln_apis.cpp:500-540)
    def get_halo_pack_secs(self): return float(self._st.halo_pack_secs)
    def get_halo_xfer_secs(self): return float(self._st.halo_xfer_secs)
    def get_halo_unpack_secs(self): return float(self._st.halo_unpack_secs)
    def get_halo_wait_secs(self): return float(self._st.halo_wait_secs)
    def get_exterior_secs(self): return float(self._st.exterior_secs)
    def get_interior_secs(self): return float(self._st.interior_secs)
    def get_halo_bytes_sent(self): return int(self._st.halo_bytes_sent)
    def get_halo_bytes_recv(self): return int(self._st.halo_bytes_recv)
    def get_halo_msgs_sent(self): return int(self._st.halo_msgs_sent)
    def get_num_fused_passes(self): return int(self._st.fused_passes)
    def get_num_graph_replays(self): return int(self._st.graph_replays)
    def get_num_graph_steps(self): return int(self._st.graph_steps)

    def get_comm_hidden_fraction(self):
        """share of the halo-exchange time (pack + transport + unpack) that ran under the interior kernel"""
        comm = self.get_halo_pack_secs() + self.get_halo_xfer_secs() + self.get_halo_unpack_secs()
        return max(0.0, 1.0 - self.get_halo_wait_secs() / comm) if comm > 0 else None


class yk_reduction_result:
    def __init__(self, r):
        self._r = r

    def get_reduction_mask(self): return int(self._r.reduction_mask)
    def get_num_elements_reduced(self): return int(self._r.num_elements_reduced)
    def get_sum(self): return float(self._r.sum)
    def get_sum_squares(self): return float(self._r.sum_squares)
    def get_product(self): return float(self._r.product)
    def get_max(self): return float(self._r.max)
    def get_min(self): return float(self._r.min)


class yk_var:
    """yk_var (include/aux/yk_var_api.hpp:185-1490). Indices are overall-domain (global) indices in
    the var's own dim order; slice bounds are inclusive."""

    yk_sum_reduction, yk_sum_squares_reduction, yk_product_reduction, yk_max_reduction, yk_min_reduction = 1, 2, 4, 8, 16

    def __init__(self, lib, h, soln):
        self._lib, self._h, self._soln = lib, h, soln

    def _idx(self, v):
        v = list(v)
        n = self.get_num_dims()
        if len(v) != n:
            raise RuntimeError("YASK error: %d indices provided for var '%s' with %d dims" % (len(v), self.get_name(), n))
        return (idx_t * n)(*[int(x) for x in v])

    def get_name(self): return self._lib.call("yk_var_get_name", self._h).decode()
    def get_num_dims(self): return self._lib.call("yk_var_get_num_dims", self._h)
    def get_dim_names(self): return [self._lib.call("yk_var_get_dim_name", self._h, i).decode() for i in range(self.get_num_dims())]
    def get_num_domain_dims(self):
        dd = self._soln.get_domain_dim_names()
        return sum(1 for d in self.get_dim_names() if d in dd)
    def is_dim_used(self, dim): return bool(self._lib.call("yk_var_is_dim_used", self._h, _b(dim)))
    def is_fixed_size(self): return bool(self._lib.call("yk_var_is_fixed_size", self._h))
    # deprecated aliases kept by the reference (yk_var_api.hpp:1472-1487)
    def get_first_rank_alloc_index(self, dim): return self.get_first_local_index(dim)
    def get_last_rank_alloc_index(self, dim): return self.get_last_local_index(dim)
    def set_numa_preferred(self, numa_node): return False          # device memory: no NUMA policy
    def get_numa_preferred(self): return -9                         # yask_numa_none
    def get_first_valid_step_index(self): return self._lib.call("yk_var_get_first_valid_step_index", self._h)
    def get_last_valid_step_index(self): return self._lib.call("yk_var_get_last_valid_step_index", self._h)

    def _vecget(self, fn):
        return [fn(d) for d in self.get_dim_names()]

    def get_first_local_index_vec(self): return self._vecget(self.get_first_local_index)
    def get_last_local_index_vec(self): return self._vecget(self.get_last_local_index)
    def get_alloc_size_vec(self): return self._vecget(self.get_alloc_size)
    def _domvec(self, fn): return [fn(d) for d in self.get_dim_names() if d in self._soln.get_domain_dim_names()]
    def get_rank_domain_size_vec(self): return self._domvec(self.get_rank_domain_size)
    def get_first_rank_domain_index_vec(self): return self._domvec(self.get_first_rank_domain_index)
    def get_last_rank_domain_index_vec(self): return self._domvec(self.get_last_rank_domain_index)
    def get_first_rank_halo_index_vec(self): return self._domvec(self.get_first_rank_halo_index)
    def get_last_rank_halo_index_vec(self): return self._domvec(self.get_last_rank_halo_index)

    def are_indices_local(self, indices): return bool(self._lib.call("yk_var_are_indices_local", self._h, self._idx(indices)))
    def get_element(self, indices): return self._lib.call("yk_var_get_element", self._h, self._idx(indices))
    def set_element(self, val, indices, strict_indices=True):
        return self._lib.call("yk_var_set_element", self._h, float(val), self._idx(indices), int(strict_indices))
    def add_to_element(self, val, indices, strict_indices=True):
        return self._lib.call("yk_var_add_to_element", self._h, float(val), self._idx(indices), int(strict_indices))

    def _np_dtype(self):
        return np.float32 if self._soln.get_element_bytes() == 4 else np.float64

    @staticmethod
    def _as_array(buf, dtype, writable):
        """numpy view of a caller buffer (ndarray, memoryview from `ndarray.data`, bytearray ...), as the SWIG
        `pybuffer` typemaps of src/kernel/swig/yask_kernel_api.i accept them."""
        if isinstance(buf, np.ndarray):
            return buf
        a = np.frombuffer(buf, dtype=dtype)
        if writable and not a.flags.writeable:
            raise RuntimeError("YASK error: buffer is read-only")
        return a

    def get_elements_in_slice(self, *args, buffer=None):
        """get_elements_in_slice(first_indices, last_indices[, buffer]) -> numpy array shaped (last-first+1) per dim
        (row-major, var dim order); or, SWIG order, get_elements_in_slice(buffer, first_indices, last_indices) ->
        number of elements read into the caller's float32/float64 buffer (yk_var_api.hpp:699-744)."""
        swig_order = len(args) == 3 and not isinstance(args[0], (list, tuple)) and not (
            isinstance(args[0], np.ndarray) and args[0].dtype.kind in "iu")
        if swig_order:
            buffer, first_indices, last_indices = args
        else:
            first_indices, last_indices = args[0], args[1]
            if len(args) > 2:
                buffer = args[2]
        f, l = list(first_indices), list(last_indices)
        shape = [int(b) - int(a) + 1 for a, b in zip(f, l)]
        if buffer is None:
            buffer = np.empty(shape, dtype=self._np_dtype())
        arr = self._as_array(buffer, self._np_dtype(), True)
        if arr.dtype not in (np.float32, np.float64) or not arr.flags.c_contiguous:
            raise RuntimeError("YASK error: get_elements_in_slice needs a C-contiguous float32/float64 buffer")
        fn = "yk_var_get_elements_in_slice_f32" if arr.dtype == np.float32 else "yk_var_get_elements_in_slice_f64"
        n = self._lib.call(fn, self._h, arr.ctypes.data_as(C.c_void_p), arr.size, self._idx(f), self._idx(l))
        return n if swig_order else arr

    def set_elements_in_slice(self, buffer, first_indices, last_indices):
        a = self._as_array(buffer, self._np_dtype(), False)
        a = np.ascontiguousarray(a)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(self._np_dtype())
        fn = "yk_var_set_elements_in_slice_f32" if a.dtype == np.float32 else "yk_var_set_elements_in_slice_f64"
        return self._lib.call(fn, self._h, a.ctypes.data_as(C.c_void_p), a.size, self._idx(first_indices), self._idx(last_indices))

    def set_elements_in_slice_same(self, val, first_indices, last_indices, strict_indices=True):
        return self._lib.call("yk_var_set_elements_in_slice_same", self._h, float(val), self._idx(first_indices),
                              self._idx(last_indices), int(strict_indices))

    def set_all_elements_same(self, val): self._lib.call_rc("yk_var_set_all_elements_same", self._h, float(val))

    def reduce_elements_in_slice(self, reduction_mask, first_indices, last_indices, strict_indices=True):
        r = _capi.YkReduction()
        self._lib.call_rc("yk_var_reduce_elements_in_slice", self._h, int(reduction_mask), self._idx(first_indices),
                          self._idx(last_indices), int(strict_indices), C.byref(r))
        return yk_reduction_result(r)

    def format_indices(self, indices):
        return ", ".join("%s=%d" % (d, i) for d, i in zip(self.get_dim_names(), indices))

    def get_halo_exchange_l1_norm(self): return self._lib.call("yk_var_get_halo_exchange_l1_norm", self._h)
    def set_halo_exchange_l1_norm(self, n): self._lib.call_rc("yk_var_set_halo_exchange_l1_norm", self._h, int(n))
    def is_dynamic_step_alloc(self): return bool(self._lib.call("yk_var_is_dynamic_step_alloc", self._h))
    def is_storage_allocated(self): return bool(self._lib.call("yk_var_is_storage_allocated", self._h))
    def get_num_storage_bytes(self): return self._lib.call("yk_var_get_num_storage_bytes", self._h)
    def get_num_storage_elements(self): return self._lib.call("yk_var_get_num_storage_elements", self._h)
    def alloc_storage(self): self._lib.call_rc("yk_var_alloc_storage", self._h)
    def release_storage(self): self._lib.call_rc("yk_var_release_storage", self._h)
    def is_storage_layout_identical(self, other): return bool(self._lib.call("yk_var_is_storage_layout_identical", self._h, other._h))
    def fuse_vars(self, source): self._lib.call_rc("yk_var_fuse_vars", self._h, source._h)
    def get_raw_storage_buffer(self): return self._lib.call("yk_var_get_raw_storage_buffer", self._h)
    def release_raw_storage_buffer(self): self._lib.call_rc("yk_var_release_raw_storage_buffer", self._h)   # extension
    def get_device_storage(self): return self._lib.call("yk_var_get_device_storage", self._h)
    # extension (not in the reference API): layout-independent deterministic init
    def set_elements_hash(self, offset=0.0, scale=1.0, hash_id=0):
        self._lib.call_rc("yk_var_set_elements_hash", self._h, float(offset), float(scale), int(hash_id))


def _dim_getter(cname):
    def f(self, dim):
        return self._lib.call(cname, self._h, _b(dim))
    return f


def _dim_setter(cname):
    def f(self, dim, n):
        self._lib.call_rc(cname, self._h, _b(dim), int(n))
    return f


for _n in ["first_local_index", "last_local_index", "alloc_size", "rank_domain_size", "first_rank_domain_index",
           "last_rank_domain_index", "left_halo_size", "right_halo_size", "first_rank_halo_index", "last_rank_halo_index",
           "left_pad_size", "right_pad_size", "left_extra_pad_size", "right_extra_pad_size", "first_misc_index",
           "last_misc_index"]:
    setattr(yk_var, "get_" + _n, _dim_getter("yk_var_get_" + _n))
for _n in ["left_min_pad_size", "right_min_pad_size", "min_pad_size", "left_halo_size", "right_halo_size", "halo_size",
           "first_misc_index", "alloc_size"]:
    setattr(yk_var, "set_" + _n, _dim_setter("yk_var_set_" + _n))


class yk_solution:
    """yk_solution (include/aux/yk_solution_api.hpp:82-1292)."""

    def __init__(self, lib, h, env):
        self._lib, self._h, self._env = lib, h, env
        self._vars = {}

    def __del__(self):
        try:
            if self._h:
                self._lib.c.yk_free_solution(self._h)
                self._h = None
        except Exception:
            pass

    def get_name(self): return self._lib.call("yk_solution_get_name", self._h).decode()
    def get_description(self): return self._lib.call("yk_solution_get_description", self._h).decode()
    def get_target(self): return self._lib.call("yk_solution_get_target", self._h).decode()
    def is_offloaded(self): return bool(self._lib.call("yk_solution_is_offloaded", self._h))
    def get_element_bytes(self): return self._lib.call("yk_solution_get_element_bytes", self._h)
    def get_step_dim_name(self): return self._lib.call("yk_solution_get_step_dim_name", self._h).decode()
    def get_num_domain_dims(self): return self._lib.call("yk_solution_get_num_domain_dims", self._h)
    def get_domain_dim_names(self):
        return [self._lib.call("yk_solution_get_domain_dim_name", self._h, i).decode() for i in range(self.get_num_domain_dims())]
    def get_misc_dim_names(self):
        n = self._lib.call("yk_solution_get_num_misc_dims", self._h)
        return [self._lib.call("yk_solution_get_misc_dim_name", self._h, i).decode() for i in range(n)]

    def _vec_set(self, setter, vals):
        dd = self.get_domain_dim_names()
        vals = list(vals)
        if len(vals) != len(dd):
            raise RuntimeError("YASK error: %d values provided for %d domain dims" % (len(vals), len(dd)))
        for d, v in zip(dd, vals):
            setter(d, v)

    def set_rank_domain_size_vec(self, vals): self._vec_set(self.set_rank_domain_size, vals)
    def get_rank_domain_size_vec(self): return [self.get_rank_domain_size(d) for d in self.get_domain_dim_names()]
    def set_overall_domain_size_vec(self, vals): self._vec_set(self.set_overall_domain_size, vals)
    def get_overall_domain_size_vec(self): return [self.get_overall_domain_size(d) for d in self.get_domain_dim_names()]
    def set_num_ranks_vec(self, vals): self._vec_set(self.set_num_ranks, vals)
    def get_num_ranks_vec(self): return [self.get_num_ranks(d) for d in self.get_domain_dim_names()]
    def set_rank_index_vec(self, vals): self._vec_set(self.set_rank_index, vals)
    def get_rank_index_vec(self): return [self.get_rank_index(d) for d in self.get_domain_dim_names()]
    def set_block_size_vec(self, vals): self._vec_set(self.set_block_size, vals)
    def get_block_size_vec(self): return [self.get_block_size(d) for d in self.get_domain_dim_names()]
    def get_first_rank_domain_index_vec(self): return [self.get_first_rank_domain_index(d) for d in self.get_domain_dim_names()]
    def get_last_rank_domain_index_vec(self): return [self.get_last_rank_domain_index(d) for d in self.get_domain_dim_names()]
    def get_num_outer_threads(self): return 1
    def get_num_inner_threads(self): return 1

    def apply_command_line_options(self, args):
        if not isinstance(args, str):
            args = " ".join(args)
        buf = C.create_string_buffer(4096)
        self._lib.call_rc("yk_solution_apply_command_line_options", self._h, _b(args), buf, 4096)
        return buf.value.decode()

    def get_command_line_help(self): return self._lib.call("yk_solution_get_command_line_help", self._h).decode()
    def get_command_line_values(self): return self._lib.call("yk_solution_get_command_line_values", self._h).decode()
    def get_num_vars(self): return self._lib.call("yk_solution_get_num_vars", self._h)

    def _wrap_var(self, h):
        if not h:
            self._lib.check()
            raise RuntimeError("YASK error: var not found")
        key = int(h)
        if key not in self._vars:
            self._vars[key] = yk_var(self._lib, h, self)
        return self._vars[key]

    def get_var(self, name):
        h = self._lib.c.yk_solution_get_var(self._h, _b(name))
        self._lib.check()
        return self._wrap_var(h)

    def get_vars(self):
        return [self._wrap_var(self._lib.call("yk_solution_get_var_by_index", self._h, i)) for i in range(self.get_num_vars())]

    def prepare_solution(self):
        self._lib.call_rc("yk_solution_prepare", self._h)
        # the reference reports the prepared configuration through the env's debug output (soln_apis.cpp:137-249)
        out = yk_env.get_debug_output()
        if out is not None and getattr(out, "_kind", "null") != "null":
            dd = self.get_domain_dim_names()
            out.write("Solution '%s' prepared on target '%s': overall-domain %s, rank-domain %s, num-ranks %s, kernel(s) %s\n" % (
                self.get_name(), self.get_target(), "*".join(str(self.get_overall_domain_size(d)) for d in dd),
                "*".join(str(self.get_rank_domain_size(d)) for d in dd), "*".join(str(self.get_num_ranks(d)) for d in dd),
                ", ".join(self.get_kernel_variant(p) for p in range(self.get_num_parts()))))
            pl = self.get_placement_trials()
            if pl:
                out.write("Var placement: %d sets of allocations timed (ms per step: %s), kept set %d\n" % (
                    len(pl["ms_per_step_of_each_set"]), ", ".join("%.4f" % m for m in pl["ms_per_step_of_each_set"]), pl["kept"]))

    def set_debug_output(self, debug): yk_env.set_debug_output(debug)

    def run_solution(self, first_step_index, last_step_index=None):
        if last_step_index is None:
            last_step_index = first_step_index
        self._lib.call_rc("yk_solution_run", self._h, int(first_step_index), int(last_step_index))

    def end_solution(self): self._lib.call_rc("yk_solution_end", self._h)
    def exchange_halos(self): self._lib.call_rc("yk_solution_exchange_halos", self._h)
    def copy_vars_to_device(self): self._lib.call_rc("yk_solution_copy_vars_to_device", self._h)
    def copy_vars_from_device(self): self._lib.call_rc("yk_solution_copy_vars_from_device", self._h)

    def get_stats(self):
        st = _capi.YkStats()
        self._lib.call_rc("yk_solution_get_stats", self._h, C.byref(st))
        return yk_stats(st)

    def clear_stats(self): self._lib.call_rc("yk_solution_clear_stats", self._h)

    def get_step_times(self):
        """per-step ms (HIP events) of the last run_solution(); needs the option -hip_step_timers"""
        n = self._lib.call("yk_solution_get_step_times", self._h, None, 0)
        buf = (C.c_float * max(1, n))()
        self._lib.call("yk_solution_get_step_times", self._h, buf, n)
        return [float(buf[i]) for i in range(n)]
    def get_placement_trials(self):
        """Var placement of the last prepare_solution() (-hip_placement_trials): ms per step measured on each set of var
        allocations drawn, and which set was kept; None when no search ran (small solution, or vars that already held data)."""
        chosen, ms = C.c_int(0), (C.c_float * 32)()
        n = self._lib.call("yk_solution_get_placement_trials", self._h, C.byref(chosen), ms, 32)
        if n <= 0:
            return None
        return {"ms_per_step_of_each_set": [round(float(ms[i]), 4) for i in range(min(n, 32))], "kept": int(chosen.value)}
    def set_min_pad_size(self, dim, n): self._lib.call_rc("yk_solution_set_min_pad_size", self._h, _b(dim), n)
    def get_min_pad_size(self, dim): return self._lib.call("yk_solution_get_min_pad_size", self._h, _b(dim))
    def set_step_wrap(self, do_wrap): self._lib.call_rc("yk_solution_set_step_wrap", self._h, int(bool(do_wrap)))
    def get_step_wrap(self): return bool(self._lib.call("yk_solution_get_step_wrap", self._h))

    def reset_auto_tuner(self, enable, verbose=False): self._lib.call_rc("yk_solution_reset_auto_tuner", self._h, int(enable), int(verbose))
    def is_auto_tuner_enabled(self): return bool(self._lib.call("yk_solution_is_auto_tuner_enabled", self._h))
    def run_auto_tuner_now(self, verbose=True): self._lib.call_rc("yk_solution_run_auto_tuner_now", self._h, int(verbose))

    def new_var(self, name, dims):
        arr = (C.c_char_p * len(dims))(*[_b(d) for d in dims])
        h = self._lib.c.yk_solution_new_var(self._h, _b(name), len(dims), arr)
        self._lib.check()
        return self._wrap_var(h)

    def new_fixed_size_var(self, name, dims, dim_sizes):
        arr = (C.c_char_p * len(dims))(*[_b(d) for d in dims])
        sz = (idx_t * len(dim_sizes))(*[int(s) for s in dim_sizes])
        h = self._lib.c.yk_solution_new_fixed_size_var(self._h, _b(name), len(dims), arr, sz)
        self._lib.check()
        return self._wrap_var(h)

    # ---- extensions beyond the reference API
    def compare_data(self, ref, epsilon=1e-3):
        """Number of in-domain mismatches vs `ref` (StencilContext::compare_data, context.cpp:1529-1547)."""
        return self._lib.call("yk_solution_compare_data", self._h, ref._h, float(epsilon))
    def set_streams(self, compute_stream, comm_stream):
        self._lib.call_rc("yk_solution_set_streams", self._h, C.c_void_p(compute_stream), C.c_void_p(comm_stream))
    def get_num_parts(self):
        n = 0
        while True:
            try:
                if self._lib.c.yk_solution_get_num_kernel_variants(self._h, n) <= 0:
                    self._lib.c.yk_clear_error()
                    return n
            except Exception:
                return n
            n += 1

    def get_kernel_variant(self, part=0): return self._lib.call("yk_solution_get_kernel_variant", self._h, part).decode()
    def get_kernel_variant_scratch_bytes(self, part, i):
        return self._lib.call("yk_solution_get_kernel_variant_scratch_bytes", self._h, int(part), int(i))

    def get_part_bounding_box(self, part):
        """(kind, first, last): kind 0 = unconditional part, 1 = box of its IF_DOMAIN condition in this rank
        (rank-local indices, last inclusive), 2 = the condition holds nowhere in this rank."""
        first, last = (idx_t * 3)(), (idx_t * 3)()
        kind = self._lib.call("yk_solution_get_part_bounding_box", self._h, int(part), first, last)
        return kind, list(first), list(last)

    def get_part_full_boxes(self, part):
        """The full boxes of a sub-domain part in this rank (the reference's _bb_list): [(first, last), ...] in rank-local indices,
        last inclusive; [] where there is no list (unconditional, a condition that fills its bounding box, or per-point predicate)."""
        cap = 64
        first, last = (idx_t * (3 * cap))(), (idx_t * (3 * cap))()
        n = self._lib.call("yk_solution_get_part_full_boxes", self._h, int(part), cap, first, last)
        return [(list(first[3 * i:3 * i + 3]), list(last[3 * i:3 * i + 3])) for i in range(min(n, cap))]

    def get_fused_groups(self):
        """Fused scratch groups in use (csrc/ykh_fused.hpp: a run of scratch stages and the stage they feed as one kernel per step, scratch
        vars in the LDS): [] when every part is a sweep of its own, else one dict per group."""
        info = (C.c_longlong * 6)()
        n = self._lib.call("yk_solution_get_fused_groups", self._h, -1, None)
        out = []
        for g in range(max(0, n)):
            self._lib.call("yk_solution_get_fused_groups", self._h, g, info)
            out.append(dict(parts=info[0], scratch_vars=info[1], lds_slots=info[2], lds_bytes=info[3], tile=(info[4], info[5])))
        return out

    def get_kernel_variant_names(self, part=0):
        n = self._lib.call("yk_solution_get_num_kernel_variants", self._h, part)
        return [self._lib.call("yk_solution_get_kernel_variant_name", self._h, part, i).decode() for i in range(n)]
    def get_part_info(self, part=0):
        """Work of one stencil part per step on this rank (the reference's per-part work stats, stencil_calc.cpp:461-598) and its
        compulsory HBM bytes per point; dict of the fields of yk_part_info_t."""
        from ._capi import PartInfo
        pi = PartInfo()
        self._lib.call_rc("yk_solution_get_part_info", self._h, int(part), C.byref(pi))
        d = {k: getattr(pi, k) for k, _ in PartInfo._fields_}
        d["name"] = d["name"].decode() if d["name"] else ""
        return d

    def time_part(self, part=0, variant=-1, xchunk=0, t=0, reps=1):
        """Average HIP-event duration (ms) of `reps` launches of one stencil part."""
        ms = C.c_float(0)
        self._lib.call_rc("yk_solution_time_part", self._h, int(part), int(variant), int(xchunk), int(t), int(reps), C.byref(ms))
        return float(ms.value)

    def time_decomposed_step(self, has_lo, has_hi, reps=5):
        """(exterior ms, interior ms, whole-box ms): one step's launches as a rank with neighbours on the given sides."""
        lo, hi, ms = (C.c_int * 3)(*[int(bool(x)) for x in has_lo]), (C.c_int * 3)(*[int(bool(x)) for x in has_hi]), (C.c_float * 3)()
        self._lib.call_rc("yk_solution_time_decomposed_step", self._h, lo, hi, int(reps), ms)
        return tuple(float(x) for x in ms)

    def time_part_box(self, first, last, part=0, variant=-1, xchunk=0, t=0, reps=1):
        """Same over a sub-box (rank-local indices, last inclusive): e.g. an exterior slab of a decomposed run."""
        ms = C.c_float(0)
        f, l = (idx_t * 3)(*[int(x) for x in first]), (idx_t * 3)(*[int(x) for x in last])
        self._lib.call_rc("yk_solution_time_part_box", self._h, int(part), int(variant), int(xchunk), f, l, int(t), int(reps), C.byref(ms))
        return float(ms.value)


for _n in ["rank_domain_size", "overall_domain_size", "block_size", "num_ranks", "rank_index"]:
    setattr(yk_solution, "get_" + _n, _dim_getter("yk_solution_get_" + _n))
    setattr(yk_solution, "set_" + _n, _dim_setter("yk_solution_set_" + _n))
for _n in ["first_rank_domain_index", "last_rank_domain_index"]:
    setattr(yk_solution, "get_" + _n, _dim_getter("yk_solution_get_" + _n))


class yk_env:
    """yk_env (include/yask_kernel_api.hpp:167-295)."""
    _trace = False

    def __init__(self, lib, h):
        self._lib, self._h = lib, h
        self._keep = []     # keep ctypes callbacks alive

    def __del__(self):
        try:
            if self._h:
                self._lib.c.yk_free_env(self._h)
                self._h = None
        except Exception:
            pass

    def get_num_ranks(self): return self._lib.call("yk_env_get_num_ranks", self._h)
    def get_rank_index(self): return self._lib.call("yk_env_get_rank_index", self._h)
    def global_barrier(self): self._lib.call_rc("yk_env_global_barrier", self._h)
    def sum_over_ranks(self, v): return self._lib.call("yk_env_sum_over_ranks", self._h, int(v))
    def finalize(self): pass
    def exit(self, code): os._exit(code)
    @staticmethod
    def set_trace_enabled(enable): yk_env._trace = bool(enable)
    @staticmethod
    def is_trace_enabled(): return yk_env._trace
    _debug = None            # a yask_output (yask_amd.kernel.yask_output); None = the default stdout output, created lazily

    @staticmethod
    def set_debug_output(debug): yk_env._debug = debug
    @staticmethod
    def get_debug_output():
        if yk_env._debug is None:
            yk_env._debug = yask_output_factory().new_stdout_output()
        return yk_env._debug
    @staticmethod
    def disable_debug_output(): yk_env._debug = yask_output_factory().new_null_output()

    # ---- multi-GPU set-up (one process per GPU)
    def get_device_bus_id(self):
        """PCI bus id of the device this process computes on."""
        buf = C.create_string_buffer(64)
        self._lib.call("yk_env_get_device_bus_id", self._h, buf, 64)
        return buf.value.decode()

    def set_ranks(self, rank, num_ranks): self._lib.call_rc("yk_env_set_ranks", self._h, int(rank), int(num_ranks))

    def set_transport(self, start, wait, allreduce):
        """Install Python callables as halo transport (see yask_amd.dist for torch.distributed ones)."""
        s, w, a = _capi.EXCHANGE_FN(start), _capi.EXCHANGE_FN(wait), _capi.ALLREDUCE_FN(allreduce)
        self._keep += [s, w, a]
        self._lib.call_rc("yk_env_set_transport", self._h, s, w, a, None)

    def init_rccl(self, unique_id: bytes, rank, num_ranks):
        buf = C.create_string_buffer(unique_id, 128)
        self._lib.call_rc("yk_env_init_rccl", self._h, buf, int(rank), int(num_ranks))

    def rccl_get_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._lib.call_rc("yk_rccl_get_unique_id", buf)
        return buf.raw

    def init_from_launcher(self):
        """Native bootstrap (no torch): RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* -> TCP rendezvous of the ncclUniqueId -> RCCL
        (or the host-staged TCP transport when YASK_HIP_TRANSPORT=tcp); what the C++ yk_factory::new_env() does."""
        self._lib.call_rc("yk_env_init_from_launcher", self._h)

    def init_tcp(self, rank, num_ranks, addr="127.0.0.1", base_port=29600):
        self._lib.call_rc("yk_env_init_tcp", self._h, int(rank), int(num_ranks), _b(addr), int(base_port))

    def init_ipc(self, rank, num_ranks, addr="127.0.0.1", base_port=29600):
        """Device-to-device transport between the ranks of one host (HIP IPC handles + stream-ordered flags, ykh_ipc.cpp)."""
        self._lib.call_rc("yk_env_init_ipc", self._h, int(rank), int(num_ranks), _b(addr), int(base_port))

    def init_mirror(self, rank, num_ranks):
        """Timing instrument (see yk_env_init_mirror): this process plays one rank of a decomposed job, its messages come back to it."""
        self._lib.call_rc("yk_env_init_mirror", self._h, int(rank), int(num_ranks))

    def transport_loopback(self, nbytes=1 << 22):
        """Run the installed halo transport once with this rank as its own peer and verify the bytes."""
        self._lib.call_rc("yk_env_transport_loopback", self._h, int(nbytes))

    def get_transport_counters(self):
        """Control-plane counters of the installed halo transport (yk_env_get_transport_counters); {} when it keeps none."""
        buf = (C.c_longlong * 8)()
        n = int(self._lib.call("yk_env_get_transport_counters", self._h, buf, 8))
        names = ("ctl_msgs", "ctl_bytes", "begins", "resets", "dev_ops", "mailbox_kind")
        return {k: int(buf[i]) for i, k in enumerate(names[:max(n, 0)])}

    def probe_bandwidth(self, kind=1, nbytes=1 << 30, reps=3):
        """GB/s a 16-byte-per-lane streaming kernel gets on this device now: kind 0 copy, 1 three reads + one write, 2 read."""
        return float(self._lib.call("yk_env_probe_bandwidth", self._h, int(kind), int(nbytes), int(reps)))


class yk_factory:
    """yk_factory (include/yask_kernel_api.hpp:82-161). The reference builds one Python module per
    stencil; here the stencil is chosen at construction (default: $YASK_STENCIL or 'iso3dfd')."""

    def __init__(self, stencil=None):
        self.stencil = stencil or os.environ.get("YASK_STENCIL", "iso3dfd")
        self._lib = _Lib(self.stencil)

    def get_version_string(self): return self._lib.call("yk_get_version_string").decode()

    def new_env(self):
        h = self._lib.c.yk_new_env()
        self._lib.check()
        if not h:
            raise RuntimeError("YASK error: cannot create env")
        return yk_env(self._lib, h)

    def new_solution(self, env, source=None):
        if source is None:
            h = self._lib.c.yk_new_solution(env._h)
        else:
            h = self._lib.c.yk_new_solution_from(env._h, source._h)
        self._lib.check()
        if not h:
            raise RuntimeError("YASK error: cannot create solution")
        return yk_solution(self._lib, h, env)
