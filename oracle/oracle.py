"""oracle/oracle.py -- numpy face of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (yask_amd/) never does.  It wraps oracle/stencil_oracle.c (a plain-C restatement of the
reference's iso3dfd / 3axis / ssg arithmetic, see that file's header for reference file:line) and
offers `run_*` helpers that advance a whole solution N steps with the reference's step-slot rules.

Parity status: pinned -- see tests/test_oracle_vs_reference.py and tests/golden/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libstencil_oracle.so"
_lib = None

SSG_FIELDS = ["v_bl_w", "v_tl_v", "v_tr_u", "s_bl_yz", "s_br_xz", "s_tl_xx", "s_tl_yy", "s_tl_zz", "s_tr_xy"]
SSG_COEFFS = ["rho", "mu", "lambda", "lambdamu2"]
# Var ordinals as the reference registers them (yk_solution::get_vars() order), used as hash ids.
VAR_IDS = {
    "iso3dfd": {"p": 0, "v": 1},
    "3axis": {"A": 0},
    # ElasticStencilBase declares rho first, then SSGElasticStencil's vars in declaration order.
    "ssg": {"rho": 0, "v_bl_w": 1, "v_tl_v": 2, "v_tr_u": 3, "s_bl_yz": 4, "s_br_xz": 5, "s_tl_xx": 6,
            "s_tl_yy": 7, "s_tl_zz": 8, "s_tr_xy": 9, "mu": 10, "lambda": 11, "lambdamu2": 12},
}


def build(force: bool = False) -> Path:
    """Compile oracle/stencil_oracle.c with gcc (OpenMP on) into oracle/libstencil_oracle.so."""
    src = [_HERE / "stencil_oracle.c", _HERE / "stencil_oracle_body.inc"]
    if force or not _LIB_PATH.exists() or any(s.stat().st_mtime > _LIB_PATH.stat().st_mtime for s in src):
        cmd = ["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden",
               "-o", str(_LIB_PATH), str(src[0]), "-lm"]
        subprocess.check_call(cmd, cwd=str(_HERE))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.yo_hash_unit.restype = C.c_double
        _lib.yo_hash_unit.argtypes = [C.c_int64] * 5
    return _lib


def _suf(dtype):
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def hash_unit(vid, slot, x, y, z) -> float:
    return lib().yo_hash_unit(int(vid), int(slot), int(x), int(y), int(z))


def iso3dfd_coeffs(radius: int) -> np.ndarray:
    c = np.zeros(radius + 1, dtype=np.float64)
    lib().yo_iso3dfd_coeffs(_ptr(c), C.c_int(radius))
    return c


def center_fd_coefficients(order: int, radius: int) -> np.ndarray:
    c = np.zeros(2 * radius + 1, dtype=np.float64)
    lib().yo_center_fd_coefficients(_ptr(c), C.c_int(order), C.c_int(radius))
    return c


def fill(shape, H, vid, slot, offset=0.0, scale=1.0, dtype=np.float32, origin=(0, 0, 0)) -> np.ndarray:
    """Box [nx+2H, ny+2H, nz+2H] filled with offset + scale*hash(vid, slot, global x, y, z)."""
    nx, ny, nz = shape
    a = np.empty((nx + 2 * H, ny + 2 * H, nz + 2 * H), dtype=dtype)
    getattr(lib(), "yo_fill_" + _suf(dtype))(
        _ptr(a), C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_int64(H), C.c_int64(vid), C.c_int64(slot),
        C.c_double(offset), C.c_double(scale), C.c_int64(origin[0]), C.c_int64(origin[1]), C.c_int64(origin[2]))
    return a


def interior(a, H):
    return a[H:a.shape[0] - H, H:a.shape[1] - H, H:a.shape[2] - H] if H else a


DEFAULT_INIT = {
    "iso3dfd": {"p": (0.0, 1.0), "v": (150.0, 50.0)},
    "3axis": {"A": (0.0, 1.0)},
    "ssg": {**{f: (0.0, 1.0e-3) for f in SSG_FIELDS}, "rho": (1.5, 0.5), "mu": (1.5, 0.5),
            "lambda": (1.5, 0.5), "lambdamu2": (1.5, 0.5)},
}


def run_iso3dfd(shape, steps, radius=8, dtype=np.float32, init=None, origin=(0, 0, 0), first_step=0):
    """Advance iso3dfd `steps` steps from hashed initial data. Returns {(name, step): interior array}.
    Slots: p has 2; slot = t mod 2; p(t+1) overwrites p(t-1)."""
    init = {**DEFAULT_INIT["iso3dfd"], **(init or {})}
    ids = VAR_IDS["iso3dfd"]
    H = radius
    nx, ny, nz = shape
    p = [fill(shape, H, ids["p"], s, *init["p"], dtype=dtype, origin=origin) for s in (0, 1)]
    v = fill(shape, H, ids["v"], 0, *init["v"], dtype=dtype, origin=origin)
    fn = getattr(lib(), "yo_iso3dfd_step_" + _suf(dtype))
    for t in range(first_step, first_step + steps):
        cur, io = p[t % 2], p[(t + 1) % 2]
        fn(_ptr(cur), _ptr(io), _ptr(v), C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_int64(H), C.c_int(radius))
    last = first_step + steps
    out = {("v", 0): interior(v, H).copy()}
    for t in (last - 1, last):
        out[("p", t)] = interior(p[t % 2], H).copy()
    return out


def run_axis3(shape, steps, radius=4, dtype=np.float64, init=None, origin=(0, 0, 0), first_step=0):
    init = {**DEFAULT_INIT["3axis"], **(init or {})}
    H = radius
    nx, ny, nz = shape
    a = [fill(shape, H, 0, s, *init["A"], dtype=dtype, origin=origin) for s in (0, 1)]
    fn = getattr(lib(), "yo_axis3_step_" + _suf(dtype))
    for t in range(first_step, first_step + steps):
        fn(_ptr(a[t % 2]), _ptr(a[(t + 1) % 2]), C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_int64(H), C.c_int(radius))
    last = first_step + steps
    return {("A", t): interior(a[t % 2], H).copy() for t in (last - 1, last)}


def run_ssg(shape, steps, dtype=np.float32, init=None, origin=(0, 0, 0), first_step=0):
    init = {**DEFAULT_INIT["ssg"], **(init or {})}
    ids = VAR_IDS["ssg"]
    H = 4
    nx, ny, nz = shape
    f = {n: fill(shape, H, ids[n], 0, *init[n], dtype=dtype, origin=origin) for n in SSG_FIELDS}
    k = {n: fill(shape, H, ids[n], 0, *init[n], dtype=dtype, origin=origin) for n in SSG_COEFFS}
    fa = (C.c_void_p * 9)(*[f[n].ctypes.data for n in SSG_FIELDS])
    ka = (C.c_void_p * 4)(*[k[n].ctypes.data for n in SSG_COEFFS])
    s1 = getattr(lib(), "yo_ssg_stage1_" + _suf(dtype))
    s2 = getattr(lib(), "yo_ssg_stage2_" + _suf(dtype))
    for _ in range(steps):
        s1(fa, ka, C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_int64(H))
        s2(fa, ka, C.c_int64(nx), C.c_int64(ny), C.c_int64(nz), C.c_int64(H))
    last = first_step + steps
    out = {(n, last): interior(f[n], H).copy() for n in SSG_FIELDS}
    out.update({(n, 0): interior(k[n], H).copy() for n in SSG_COEFFS})
    return out


def within_tolerance(val, ref, eps=1e-3):
    """The reference's own comparison rule (src/kernel/lib/realv.hpp:974-994):
    absolute if |ref| <= 1 else relative."""
    val = np.asarray(val, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = np.abs(val - ref)
    return np.where(np.abs(ref) > 1.0, d / np.abs(ref), d) <= eps


def rel_linf(val, ref):
    """max|val-ref| / max(1, max|ref|): the repo's stated parity metric."""
    val = np.asarray(val, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.max(np.abs(val - ref)) / max(1.0, float(np.max(np.abs(ref)))))


def load_ref_dump(prefix):
    """Read a ref_driver -out PREFIX dump -> {(name, step): array}."""
    import json
    prefix = str(prefix)
    man = json.load(open(prefix + ".json"))
    dt = np.float32 if man["elem_bytes"] == 4 else np.float64
    out = {}
    d = os.path.dirname(prefix)
    for v in man["vars"]:
        out[(v["name"], v["step"])] = np.fromfile(os.path.join(d, v["file"]), dtype=dt).reshape(v["shape"])
    return out


def lattice(n, stride=32, edge=9, tile=0):
    """Sample indices along one dim for the BASELINE-size fixtures: every point of the `edge`-wide boundary layers
    (where clamped loads, masks and halo reads live) plus every `stride`-th point in between.  tile > 0 (the multi-tile
    fixtures of the generic kernels): also both sides of every multiple of `tile` -- m-1 and m -- which is where a
    workgroup's tile ends and its neighbour's begins (tile halos, ring wrap, x-chunk seams)."""
    n, stride, edge, tile = int(n), int(stride), int(edge), int(tile)
    idx = set(range(min(edge, n))) | set(range(0, n, stride)) | set(range(max(0, n - edge), n))
    if tile > 0:
        for m in range(tile, n, tile):
            idx |= {m - 1, m}
    return np.array(sorted(idx), dtype=np.int64)


def lattice_sample(a, stride=32, edge=9, tile=0):
    """a[ix][:, iy][:, :, iz] on the lattice of each of the first three dims (works on memmaps); further dims are kept whole."""
    ix, iy, iz = (lattice(s, stride, edge, tile) for s in a.shape[:3])
    return np.ascontiguousarray(np.asarray(a[ix])[:, iy][:, :, iz])
