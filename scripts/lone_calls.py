#!/usr/bin/env python3
"""Lone calls -- one launch, one synchronise, nothing in flight beside it (the reference's own usage: one call per map) -- for
a rocprofv3 kernel trace (scripts/ktimeline.py prints the dispatches of the last calls).  usage: lone_calls.py [grid] [calls]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import workload  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

g = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tx, walls, X, Y = workload(50, g)
with Context(0) as ctx:
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    p = make_params(min_order=0, max_order=2)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        ctx.launch(p, tx)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        time.sleep(0.002)
    print(f"{g}x{g}: launch -> synchronise, ms:", " ".join(f"{t:.3f}" for t in ts), "| median of the last 8:", f"{np.median(ts[-8:]):.3f}")
