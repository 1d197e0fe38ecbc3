#!/usr/bin/env python3
"""Forward sweep of the bench workload in every validity mode (hard, hard_sigmoid, sigmoid) and a few alphas."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tx, walls, X, Y = workload(grid=g)
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    for kw in (dict(approx=False), dict(approx=True, function="hard_sigmoid"), dict(approx=True, function="hard_sigmoid", alpha=50.0),
               dict(approx=True, function="sigmoid"), dict(approx=True, function="sigmoid", alpha=1000.0)):
        p = make_params(max_order=2, **kw)
        for _ in range(3): ctx.launch(p, tx)
        ctx.synchronize(); t = time.perf_counter()
        for _ in range(10): ctx.launch(p, tx)
        ctx.synchronize(); dt = (time.perf_counter() - t) / 10
        print(f"{kw}: {dt*1e3:.3f} ms per map", flush=True)
