#!/usr/bin/env python3
"""Residency timeline of the sweep kernel (needs a -DD2D_AB_TIMELINE build: AB_CMD="python scripts/timeline.py" scripts/ab_build.sh "tl:-DD2D_AB_TIMELINE"): how many patches are in
flight over the launch, from per-patch start/end stamps of the 100 MHz real-time counter."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vg = len(sys.argv) > 2 and sys.argv[2] == "vg"  # time the value+grad sweep instead of the forward sweep
tx, walls, X, Y = workload(grid=g)
T = (g // 8) ** 2
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    ctx.set_option("split_max_tiles", 0)
    order = None
    for approx in (False, True):
        p = make_params(max_order=2, approx=approx)
        ctx.set_option("cost_history", 0)   # the stamps overwrite the work history in this build
        for _ in range(2):
            if vg:
                ctx.launch_vg(p, tx, scene_vjp=True)
            else:
                ctx.launch(p, tx)
        ctx.synchronize()
        w = ctx.debug_get_work(T)
        t0 = (w >> 16).astype(np.int64); t1 = (w & 0xffff).astype(np.int64)
        base = np.min(t0)  # wrap-around handling: stamps are 16 bits of a 10 ns counter (655 us)
        t0 = (t0 - base) & 0xffff; t1 = (t1 - base) & 0xffff
        dur = (t1 - t0) & 0xffff
        end = t0 + dur
        total = end.max()
        print(f"approx={approx}: launch spans {total/100:.1f} us; patch latency mean {dur.mean()/100:.1f} us, max {dur.max()/100:.1f} us; last start at {t0.max()/100:.1f} us")
        edges = np.linspace(0, total, 21)
        for a_, b_ in zip(edges[:-1], edges[1:]):
            mid = 0.5 * (a_ + b_)
            inflight = ((t0 <= mid) & (end > mid)).sum()
            print(f"   t={mid/100:6.1f} us: {inflight:5d} patches in flight ({inflight/1024:.1f} per SIMD), started so far {(t0 <= mid).sum()}")
