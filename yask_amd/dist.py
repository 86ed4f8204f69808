"""One process per GPU: wiring of yk_env to torch.distributed (RCCL on ROCm, gloo for CPU-side tests).

The reference initialises MPI inside yk_factory::new_env() (src/kernel/lib/setup.cpp:38-137) and
derives the rank grid in prepare_solution() (setup.cpp:169-524).  Here the launcher
(`python -m torch.distributed.run`) provides RANK / LOCAL_RANK / WORLD_SIZE; this module binds the
process to its GPU, tells the env its rank, and installs a halo transport:

  "rccl"  (default on GPUs): the library's built-in RCCL send/recv (yask_amd/csrc/ykh_rccl.cpp); the
           128-byte ncclUniqueId is created on rank 0 and broadcast through torch.distributed.
  "ipc":   the library's device-to-device transport between the GPUs of one host: copies straight into the neighbour's
           buffers (HIP IPC memory handles; SDMA over xGMI, no compute units), ordered by flag words in device memory
           (yask_amd/csrc/ykh_ipc.cpp).  Ranks may share a device (one-GPU tests).
  "torch": Python callbacks that move the packed halo buffers with torch.distributed P2P ops
           (batch_isend_irecv on zero-copy views of the library's device buffers when the backend is
           nccl; staged through host memory for gloo, which is how the N>1 path is tested on one GPU
           or on CPU-only machines).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend=None):
    """Initialise torch.distributed from the launcher's environment (no-op for world size 1)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            # YASK_DIST_BACKEND=gloo: several ranks on ONE GPU (tests; RCCL needs one device per rank)
            backend = os.environ.get("YASK_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class _DevBuf:
    """Zero-copy view of a raw device pointer for torch (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class TorchTransport:
    """Halo transport over torch.distributed P2P. `staged=True` copies through host memory (gloo)."""

    def __init__(self, group=None, staged=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        backend = dist.get_backend(group)
        self.staged = (backend != "nccl") if staged is None else staged
        self._pending = []
        self._keep = []

    def _view(self, ptr, nbytes):
        return self.torch.as_tensor(_DevBuf(ptr, nbytes), device="cuda")

    def start(self, user, n, msgs, stream):
        try:
            torch, dist = self.torch, self.dist
            self._pending, self._keep = [], []
            if self.staged:
                # gloo path (tests): wait for the pack kernels, stage through pinned host buffers
                from . import _hip
                _hip.stream_synchronize(stream)
                ops = []
                for i in range(n):
                    m = msgs[i]
                    if m.recv_bytes:
                        r = torch.empty(m.recv_bytes, dtype=torch.uint8)
                        ops.append(dist.P2POp(dist.irecv, r, m.peer, self.group))
                        self._pending.append((m.recv_buf, r))
                    if m.send_bytes:
                        s = torch.empty(m.send_bytes, dtype=torch.uint8)
                        _hip.memcpy_dtoh(s.data_ptr(), m.send_buf, m.send_bytes)
                        ops.append(dist.P2POp(dist.isend, s, m.peer, self.group))
                        self._keep.append(s)
                self._reqs = dist.batch_isend_irecv(ops) if ops else []
            else:
                ext = torch.cuda.ExternalStream(int(stream))
                with torch.cuda.stream(ext):
                    ops = []
                    for i in range(n):
                        m = msgs[i]
                        if m.recv_bytes:
                            ops.append(dist.P2POp(dist.irecv, self._view(m.recv_buf, m.recv_bytes), m.peer, self.group))
                        if m.send_bytes:
                            ops.append(dist.P2POp(dist.isend, self._view(m.send_buf, m.send_bytes), m.peer, self.group))
                    self._reqs = dist.batch_isend_irecv(ops) if ops else []
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            print("yask_amd.dist: halo transport start failed:", repr(e), flush=True)
            return 1

    def wait(self, user, n, msgs, stream):
        try:
            torch = self.torch
            if self.staged:
                from . import _hip
                for r in self._reqs:
                    r.wait()
                for dst, host in self._pending:
                    _hip.memcpy_htod(dst, host.data_ptr(), host.numel())
            else:
                ext = torch.cuda.ExternalStream(int(stream))
                with torch.cuda.stream(ext):
                    for r in self._reqs:
                        r.wait()      # stream-level wait for nccl work
            self._reqs, self._pending, self._keep = [], [], []
            return 0
        except Exception as e:
            print("yask_amd.dist: halo transport wait failed:", repr(e), flush=True)
            return 1

    def allreduce(self, user, op, val):
        try:
            torch, dist = self.torch, self.dist
            dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
            t = torch.tensor([val[0]], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MIN, 2: dist.ReduceOp.MAX}[op], group=self.group)
            val[0] = int(t.item())
            return 0
        except Exception as e:
            print("yask_amd.dist: all-reduce failed:", repr(e), flush=True)
            return 1


def new_env(factory, transport="rccl", strict=False):
    """yk_factory.new_env() for a torch.distributed job. Returns (env, transport_name).
    strict: a failing RCCL set-up raises on every rank instead of falling back to the torch transport (bench.py:
    a number must not silently be measured on another transport)."""
    import torch
    import torch.distributed as dist
    env = factory.new_env()
    rank, _, world = env_rank()
    if world <= 1 or not dist.is_initialized():
        return env, "none"
    env.set_ranks(rank, world)
    used = transport
    if transport == "rccl":
        ok = 1
        try:
            ids = [env.rccl_get_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            env.init_rccl(ids[0], rank, world)
        except Exception as e:
            print(f"yask_amd.dist[{rank}]: native RCCL transport unavailable ({e!r})", flush=True)
            ok = 0
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        flag = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if strict:
                raise RuntimeError("YASK error: the native RCCL halo transport could not be initialised on every rank "
                                   "(see the messages above); refusing to fall back to the torch transport")
            print(f"yask_amd.dist[{rank}]: falling back to the torch.distributed halo transport", flush=True)
            used = "torch"
    if transport == "ipc":
        # device-to-device copies into the neighbour's buffers (HIP IPC handles) + stream-ordered flag words (csrc/ykh_ipc.cpp);
        # its control mesh listens next to the launcher's port
        port = int(os.environ.get("MASTER_PORT", "29533")) + 48 + int(os.environ.get("YASK_IPC_PORT_OFFSET", "0"))
        ok = 1
        try:
            env.init_ipc(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), port)
        except Exception as e:
            print(f"yask_amd.dist[{rank}]: IPC halo transport unavailable ({e!r})", flush=True)
            ok = 0
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        flag = torch.tensor([ok], dtype=torch.int64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if strict:
                raise RuntimeError("YASK error: the IPC halo transport could not be initialised on every rank (see the messages above)")
            used = "torch"
    if used == "torch":
        tr = TorchTransport()
        env._transport = tr
        env.set_transport(tr.start, tr.wait, tr.allreduce)
    return env, used
