#!/usr/bin/env python3
"""Option lab: time the forward sweep of a bench workload under several d2d_set_option settings in one process.

usage: opt_lab.py [--workload cfg2] [--grid G] [--approx 0|1] [--steps K] name=value,name=value  [more settings ...]
Every positional argument is one setting (comma-separated option assignments; "-" = defaults).  Prints wall ms per
step (launch sequence) and the sweep kernel's own ms (HIP events around it)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, workload  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--grid", type=int, default=None)
ap.add_argument("--approx", type=int, default=0)
ap.add_argument("--function", default="hard_sigmoid")
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--grad", type=int, default=0)
ap.add_argument("--orders", default=None, help="min,max")
ap.add_argument("settings", nargs="*", default=["-"])
args = ap.parse_args()

n_walls, wl_grid, max_order = WORKLOADS[args.workload][:3]
tx, walls, X, Y = workload(n_walls=n_walls, grid=args.grid or wl_grid)
ref = None
for setting in args.settings:
    with Context(0) as ctx:
        ctx.set_scene(walls)
        ctx.set_grid(X, Y)
        ctx.set_option("time_kernel", 1)
        if setting != "-":
            for kv in setting.split(","):
                k, v = kv.split("=")
                ctx.set_option(k, int(v))
        lo, hi = (int(x) for x in args.orders.split(",")) if args.orders else (0, max_order)
        p = make_params(min_order=lo, max_order=hi, approx=bool(args.approx), function=args.function)
        launch = (lambda: ctx.launch_vg(p, tx, scene_vjp=True)) if args.grad else (lambda: ctx.launch(p, tx))
        for _ in range(3):
            launch()
        ctx.synchronize()
        t = time.perf_counter()
        km = []
        for _ in range(args.steps):
            launch()
            km.append(ctx.last_kernel_ms())
        ctx.synchronize()
        dt = (time.perf_counter() - t) / args.steps
        # (last_kernel_ms synchronises: the wall figure below is launch + wait per step, not a pipelined rate)
        out = ctx.get_map()
        if ref is None:
            ref = out
        same = np.array_equal(out, ref, equal_nan=True)
        rs = ctx.debug_region_stats()
        print(f"{setting:40s} wall {dt*1e3:8.3f} ms  kernel {np.median(km):8.3f} ms (min {min(km):.3f})  same_as_first={same} lists={rs}", flush=True)
