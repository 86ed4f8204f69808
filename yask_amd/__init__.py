"""yask_amd -- MI355X-native (gfx950 / CDNA4) stencil runtime behind the YASK kernel API.

Only the run_solution() hot path of intel/yask is implemented here: hand-shaped HIP
kernels per stencil part, a HIP launch scheduler, device-resident vars and per-GPU domain
decomposition with RCCL halo exchange.  `yask_amd.kernel` mirrors the reference's `yask_kernel`
Python module; `yask_amd.dist` wires one process per GPU through torch.distributed.
"""
from .kernel import yk_factory, yk_env, yk_solution, yk_var, yk_stats  # noqa: F401

__version__ = "0.1.0"
