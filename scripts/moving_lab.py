#!/usr/bin/env python3
"""Moving transmitter (random walk, sigma 0.01 per step): sweep-kernel and wall time per step under option settings."""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
ap = argparse.ArgumentParser()
ap.add_argument("--approx", type=int, default=0)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("settings", nargs="*", default=["-"])
args = ap.parse_args()
tx0, walls, X, Y = workload(grid=1024)
rng = np.random.default_rng(7)
path = np.clip(tx0 + np.cumsum(rng.normal(0, 0.01, (args.steps + 3, 2)), axis=0), 0.02, 0.98).astype(np.float32)
for setting in args.settings:
    with Context(0) as ctx:
        ctx.set_scene(walls); ctx.set_grid(X, Y); ctx.set_option("time_kernel", 1)
        if setting != "-":
            for kv in setting.split(","):
                k, v = kv.split("="); ctx.set_option(k, int(v))
        p = make_params(max_order=2, approx=bool(args.approx))
        for i in range(3):
            ctx.launch(p, path[i])
        ctx.synchronize()
        km = []
        t = time.perf_counter()
        for i in range(3, 3 + args.steps):
            ctx.launch(p, path[i]); km.append(ctx.last_kernel_ms())
        ctx.synchronize()
        dt = (time.perf_counter() - t) / args.steps
        print(f"{setting:40s} wall {dt*1e3:8.3f} ms  kernel mean {np.mean(km):8.3f} median {np.median(km):8.3f} ms", flush=True)
