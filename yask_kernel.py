"""`import yask_kernel` -- the module name of the reference's SWIG-generated Python binding
(src/kernel/swig/yask_kernel_api.i; one module per stencil build), provided over the cdna4_hip kernel
library so that programs written against the reference's Python API -- e.g. its own
src/kernel/tests/yask_kernel_api_test.py -- run unchanged:

    YASK_STENCIL=test_3d python yask_kernel_api_test.py

The stencil that the reference bakes into the module at build time is chosen here by $YASK_STENCIL
(default 'iso3dfd'); `yk_factory()` takes no argument, as in the reference."""
from yask_amd.kernel import yk_env, yk_factory, yk_solution, yk_stats, yk_var  # noqa: F401


class _Cvar:
    """SWIG exposes C++ global constants through `cvar` (yask_common_api.hpp:96-114)."""
    yask_numa_local = -1
    yask_numa_interleave = -2
    yask_numa_none = -9
    yask_numa_offload = -11
    yask_numa_host = -12


cvar = _Cvar()


class yask_output:
    def __init__(self, kind, name=None):
        self._kind, self._name, self._buf = kind, name, []

    def get_filename(self): return self._name or ""
    def get_string(self): return "".join(self._buf)
    def discard(self): self._buf = []


class yask_output_factory:
    """yask_output_factory (include/yask_common_api.hpp:184-232): objects routing the library's debug output."""
    def new_file_output(self, file_name): return yask_output("file", file_name)
    def new_string_output(self): return yask_output("string")
    def new_stdout_output(self): return yask_output("stdout")
    def new_null_output(self): return yask_output("null")
