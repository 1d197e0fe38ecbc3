#!/usr/bin/env python3
"""Residency timeline of the sweep kernel (needs a -DD2D_AB_TIMELINE build: AB_CMD="python scripts/timeline.py" scripts/ab_build.sh "tl:-DD2D_AB_TIMELINE"): how many patches are in
flight over the launch, from per-patch start/end stamps of the 100 MHz real-time counter."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
from differt2d_amd import _lib as L
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
vg = len(sys.argv) > 2 and sys.argv[2] == "vg"  # time the value+grad sweep instead of the forward sweep
tx, walls, X, Y = workload(grid=g)
T = (g // 8) ** 2
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    for kv in sys.argv[3:]:
        k, v = kv.split("="); ctx.set_option(k, int(v))
    for approx in (False, True):
        p = make_params(max_order=2, approx=approx)
        for _ in range(4):  # the last launch runs with the work history and the schedule of a steady state
            if vg:
                ctx.launch_vg(p, tx, scene_vjp=True)
            else:
                ctx.launch(p, tx)
        ctx.synchronize()
        cap = 2 * (T + 4 * 4096)
        w = np.zeros(cap, np.uint32)
        L.check(ctx._lib.d2d_debug_get_work(ctx._ctx, w, -cap))  # diagnostic build: stamps of the last forward launch's workgroups
        w = w.reshape(-1, 2)
        keep = w[:, 0] != 0
        w2 = w[keep, 1]
        w = w[keep, 0]
        s0 = (w >> 16).astype(np.int64); e0 = (w & 0xffff).astype(np.int64)
        # wrap-around handling: stamps are 16 bits of a 10 ns counter (655 us); workgroup 0 starts first or nearly so
        t0 = (s0 - s0[0] + 2000) & 0xffff
        t0 -= t0.min()
        dur = (e0 - s0) & 0xffff
        pro = ((w2 >> 16).astype(np.int64) - s0) & 0xffff       # prologue
        o01 = ((w2 & 0xffff).astype(np.int64) - (w2 >> 16).astype(np.int64)) & 0xffff  # orders 0 and 1 (part 0 of a cut patch; every uncut patch)
        end = t0 + dur
        total = end.max()
        print(f"approx={approx}: {len(w)} workgroups; launch spans {total/100:.1f} us; workgroup latency mean {dur.mean()/100:.1f} us, max {dur.max()/100:.1f} us; last start at {t0.max()/100:.1f} us; wave-time {dur.sum()/100:.0f} us = {dur.sum()/total/1024:.2f} waves per SIMD on average")
        edges = np.linspace(0, total, 21)
        for a_, b_ in zip(edges[:-1], edges[1:]):
            mid = 0.5 * (a_ + b_)
            inflight = ((t0 <= mid) & (end > mid)).sum()
            print(f"   t={mid/100:6.1f} us: {inflight:5d} in flight ({inflight/1024:.1f} per SIMD), started so far {(t0 <= mid).sum()}")
        H4 = int(os.environ.get("TL_QUARTERS", "704"))  # workgroups [0, H4) are the parts of the patches cut in four
        if len(w) > H4:
            for part in range(4 if H4 else 0):
                d = dur[part:H4:4] / 100
                print(f"   part {part}: prologue mean {pro[part:H4:4].mean()/100:.1f} us, orders 0-1 mean {o01[part:H4:4].mean()/100:.1f} us")
                print(f"   part {part} of the cut patches: mean {d.mean():.1f} us, p90 {np.percentile(d, 90):.1f}, max {d.max():.1f}; end of the last {((t0 + dur)[part:H4:4]).max()/100:.1f}")
            rest = dur[H4:] / 100
            for lo_, hi_ in [(0, 16), (16, 64), (64, 176), (176, 256), (256, 1024), (1024, 4096), (4096, 8192), (8192, len(rest))]:
                d = rest[lo_:hi_]
                e = (t0 + dur)[H4 + lo_:H4 + hi_] / 100
                print(f"   uncut patches {lo_}..{hi_}: prologue mean {pro[H4+lo_:H4+hi_].mean()/100:.1f} us, orders 0-1 mean {o01[H4+lo_:H4+hi_].mean()/100:.1f} us")
                print(f"   uncut patches {lo_}..{hi_} of the schedule: duration mean {d.mean():.1f} us, max {d.max():.1f}; start {t0[H4+lo_:H4+hi_].min()/100:.1f}..{t0[H4+lo_:H4+hi_].max()/100:.1f}; last end {e.max():.1f}")
        order = np.argsort(-dur)[:8]
        print("   longest workgroups (index, start us, duration us):", [(int(i), float(t0[i]/100), float(dur[i]/100)) for i in order])
