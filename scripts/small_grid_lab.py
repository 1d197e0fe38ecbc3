#!/usr/bin/env python3
"""Small launches: ms per step (pipelined launches, one wait) and sweep-kernel time of cfg2's scene at small grids, with the
candidate-sharing kernel chosen automatically (coop_waves = -1), forced (4 / 8 / 16) and off (0).
usage: small_grid_lab.py [variants, e.g. -1,0] [sizes, e.g. 64,128,300] [steps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import workload  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

variants = [v if v == "u" else int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "-1,0").split(",")]  # "u": one wave per patch
sizes = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "64,96,128,160,200,256,300,320,384,512").split(",")]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
MODES = {"hard": dict(approx=False), "hsig": dict(approx=True, function="hard_sigmoid"), "sig": dict(approx=True, function="sigmoid")}
print("mode size " + " ".join(f"coop={v}:step/kernel[shape]" for v in variants), flush=True)
for mode, mkw in MODES.items():
    if only and mode not in only:
        continue
    for g in sizes:
        tx, walls, X, Y = workload(grid=g)
        row = []
        for v in variants:
            with Context(0) as ctx:
                if v == "u":
                    ctx.set_option("split_max_tiles", 0)
                    v = 0
                ctx.set_option("coop_waves", v)
                if v > 0:
                    ctx.set_option("coop_max_tiles", 1 << 20)
                ctx.set_scene(walls)
                ctx.set_grid(X, Y)
                p = make_params(min_order=0, max_order=2, **mkw)
                for _ in range(20):
                    ctx.launch(p, tx)
                ctx.synchronize()
                t = time.perf_counter()
                for _ in range(steps):
                    ctx.launch(p, tx)
                ctx.synchronize()
                dt = (time.perf_counter() - t) / steps * 1e3
                ctx.set_option("time_kernel", 1)
                km = []
                for _ in range(20):
                    ctx.launch(p, tx)
                    km.append(ctx.last_kernel_ms())
                row.append(f"{dt:.4f}/{np.mean(km):.4f}{list(ctx.sweep_shape())}")
        print(mode, g, " ".join(row), flush=True)
