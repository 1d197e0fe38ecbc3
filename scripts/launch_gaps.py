#!/usr/bin/env python3
"""Where a pipelined step goes (needs a -DD2D_AB_TIMELINE build: AB_CMD="python scripts/launch_gaps.py" scripts/ab_build.sh
"tl:-DD2D_AB_TIMELINE"): first workgroup start / last workgroup end of every sweep of a back-to-back sequence, from the
100 MHz real-time counter read inside the kernel -- no profiler attached, nothing serialised."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
from differt2d_amd import _lib as L
tx, walls, X, Y = workload()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for approx in (False, True):
    with Context(0) as ctx:
        ctx.set_scene(walls); ctx.set_grid(X, Y)
        for kv in sys.argv[2:]:
            k, v = kv.split("="); ctx.set_option(k, int(v))
        p = make_params(max_order=2, approx=approx)
        for _ in range(8):
            ctx.launch(p, tx)
        ctx.synchronize()
        ctx.set_option("tl_ring", 1)
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.launch(p, tx)
        ctx.synchronize()
        wall = (time.perf_counter() - t0) / n * 1e3
        w = np.zeros(1024, np.uint32)
        L.check(ctx._lib.d2d_debug_get_work(ctx._ctx, w, -7))
        r = w.view(np.uint64).reshape(256, 2)[:n].astype(np.int64)
        start, end = r[:, 0] / 100.0, r[:, 1] / 100.0  # us
        dur = end - start
        gap = start[1:] - end[:-1]
        period = np.diff(start)
        k = slice(10, n)  # steady state
        print(f"approx={approx}: wall {wall:.4f} ms per step over {n}; sweep span mean {dur[k].mean():.1f} us (min {dur[k].min():.1f}, max {dur[k].max():.1f}); "
              f"gap between sweeps mean {gap[10:].mean():.1f} us (min {gap[10:].min():.1f}, max {gap[10:].max():.1f}); period mean {period[10:].mean():.1f} us")
        print("   launches 20..33: start", np.round(start[20:34] - start[20], 1).tolist())
        print("   launches 20..33: end  ", np.round(end[20:34] - start[20], 1).tolist())
