#!/usr/bin/env python3
"""Host time to enqueue one sweep launch vs the pipelined step time (is the loop host-bound?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
tx, walls, X, Y = workload(grid=1024)
for opt in ("pipeline=1", "pipeline=0"):
    with Context(0) as c:
        c.set_scene(walls); c.set_grid(X, Y)
        k, v = opt.split("="); c.set_option(k, int(v))
        p = make_params(max_order=2, approx=False)
        for _ in range(5): c.launch(p, tx)
        c.synchronize()
        n = 300
        t0 = time.perf_counter()
        for _ in range(n): c.launch(p, tx)
        t1 = time.perf_counter()
        c.synchronize()
        t2 = time.perf_counter()
        print(f"{opt}: host enqueue {1e3*(t1-t0)/n:.4f} ms per launch, total {1e3*(t2-t0)/n:.4f} ms per step, drain after the last enqueue {1e3*(t2-t1):.3f} ms", flush=True)
