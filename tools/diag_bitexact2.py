#!/usr/bin/env python
"""GPU diagnostic, second pass: per-plane histogram of where the slab schedule with the interior in two launches differs from the
one-rank run, with the x-chunk length pinned (tools/diag_bitexact.py samples five groups of planes with the default chunking)."""
import socket
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import diag_bitexact as D  # noqa: E402

D.PLANES = list(range(0, 1024, 4))
D.STEPS = 1


def hist(name, a, b, yo=0, zo=0):
    bad = {}
    for x in sorted(a):
        if x in b:
            pa = a[x][yo:yo + b[x].shape[0], zo:zo + b[x].shape[1]]
            n = int((pa != b[x]).sum())
            if n:
                bad[x] = n
    xs = sorted(bad)
    runs, start = [], None
    for x in xs:                      # runs of consecutive sampled planes
        if start is None:
            start = prev = x
        elif x - prev > 2:
            runs.append((start, prev)); start = x
        prev = x
    if start is not None:
        runs.append((start, prev))
    print(f"{name}: {sum(bad.values())} points differ on {len(bad)} of {len(a)} sampled planes; runs of planes: {runs[:40]}", flush=True)


def main():
    import multiprocessing as mp
    a, ka = D.run_one("")
    print("one rank default kernel", ka, flush=True)
    for o in ("-hip_xchunk 256", "-hip_xchunk 103"):
        b, _ = D.run_one(o)
        hist(f"one rank '{o}'", a, b)
    L = "-no-hip_planned_launch -no-hip_thin_slab_point_kernel"
    for opts in (L, L + " -hip_xchunk 512", L + " -hip_xchunk 256", L + " -hip_xchunk 128", L + " -hip_xchunk 103"):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=D.worker, args=(r, 2, port, q, (1, 1, 2), opts, 1, D.PLANES)) for r in range(2)]
        for p in procs:
            p.start()
        parts = [q.get(timeout=120) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        for rank, f, l, out in sorted(parts):
            hist(f"two ranks (1,1,2) '{opts}': rank {rank}", a, out, f[1], f[2])


if __name__ == "__main__":
    main()
