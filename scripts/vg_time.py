#!/usr/bin/env python3
"""Time of the DEFAULT value+grad step (culled sweep + NaN scan beside it + scene VJP) at BASELINE.json configs[2], both grid roles,
hard and hard_sigmoid: wall ms per step over N back-to-back steps and the kernels' time on their stream.  For A/B builds
(scripts/ab_build.sh with AB_CMD="python scripts/vg_time.py"; D2D_LIB selects the library).
usage: python scripts/vg_time.py [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
tx, walls, X, Y = workload()
out = []
with Context(0) as ctx:
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    for role, rname in ((L.GRID_RX, "rx"), (L.GRID_TX, "tx")):
        for mname, mode in (("hard", dict(approx=False)), ("hsig", dict(approx=True))):
            p = make_params(min_order=0, max_order=2, grid_role=role, **mode)
            for _ in range(5):
                ctx.launch_vg(p, tx, scene_vjp=True)
            ctx.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(steps):
                    ctx.launch_vg(p, tx, scene_vjp=True)
                ctx.synchronize()
                best = min(best, (time.perf_counter() - t0) / steps * 1e3)
            out.append(f"{rname} {mname} {best:.4f}")
print("value+grad ms per step (best of 3 x %d): " % steps + " | ".join(out))
