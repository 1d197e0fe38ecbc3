#!/usr/bin/env python3
"""configs[3] in hard then hard_sigmoid validity on ONE context, launch by launch (wall ms, launch -> synchronise): how long the
list pools take to grow when the validity mode changes on the same grid (bench.py's strong_cfg4 legs)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

tx, walls, X, Y = workload(200, 2048)
with Context(0) as ctx:
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    for approx, n in ((False, 8), (True, 10), (False, 3), (True, 4)):
        p = make_params(min_order=0, max_order=3, approx=approx)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            ctx.launch(p, tx)
            ctx.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("approx", approx, " ".join(f"{t:.1f}" for t in ts), "ms", ctx.debug_region_stats(), flush=True)
