#!/usr/bin/env python3
"""TX-grid sweep (accumulate_on_transmitters_grid_over_paths) on the bench workload: culled vs exhaustive kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd import _lib as L
from differt2d_amd.engine import Context, make_params
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rx, walls, X, Y = workload(grid=g)
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    for approx in (False, True):
        p = make_params(max_order=2, approx=approx, grid_role=L.GRID_TX)
        res = {}
        for name, opt in (("culled", 0), ("exhaustive", 1)):
            ctx.set_option("txg_exhaustive", opt)
            for _ in range(2): ctx.launch(p, rx)
            ctx.synchronize(); t = time.perf_counter()
            n = 10 if opt == 0 else 3
            for _ in range(n): ctx.launch(p, rx)
            ctx.synchronize(); dt = (time.perf_counter() - t) / n
            res[name] = ctx.get_map()
            print(f"approx={approx} {name}: {dt*1e3:.3f} ms per map", flush=True)
        print("   identical:", np.array_equal(res["culled"], res["exhaustive"], equal_nan=True))
        ctx.set_option("txg_exhaustive", 0)
    # value + grad (per-cell d/d tx, scene VJP)
    for approx in (False, True):
        p = make_params(max_order=2, approx=approx, grid_role=L.GRID_TX)
        for name, strict in (("culled", False), ("exhaustive (strict_nan)", True)):
            p.strict_nan = int(strict)
            for _ in range(2): ctx.launch_vg(p, rx, scene_vjp=True)
            ctx.synchronize(); t = time.perf_counter()
            n = 10 if not strict else 3
            for _ in range(n): ctx.launch_vg(p, rx, scene_vjp=True)
            ctx.synchronize()
            print(f"value+grad approx={approx} {name}: {(time.perf_counter() - t) / n * 1e3:.3f} ms", flush=True)
