#!/usr/bin/env python3
"""Where a wave's time goes (instrumented build, bench workload)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tx, walls, X, Y = workload(grid=g)
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    for approx in (False, True):
        for _ in range(3):  # (work history, last-segment masks)
            ctx.launch(make_params(max_order=2, approx=approx), tx)
        st = ctx.launch_stats(make_params(max_order=2, approx=approx), tx).astype(np.float64)
        waves = (g // 8) ** 2
        names = ["prologue", "order0", "order1", "order2(total)", "order2 exact part"]
        print(f"approx={approx} grid {g}: per-patch ticks:", {n: round(st[10 + i] / waves) for i, n in enumerate(names)},
              " exact cands/patch", st[0] / waves, "cull levels/patch", st[9] / waves)
