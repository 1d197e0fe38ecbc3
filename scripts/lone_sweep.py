#!/usr/bin/env python3
"""The forward sweep kernel's own time when nothing runs beside it (a synchronisation after every launch: the next launch's
preparation cannot overlap it) against its time in the pipelined sequence of bench.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import workload  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

tx, walls, X, Y = workload(50, 1024)
with Context(0) as ctx:
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    ctx.set_option("time_kernel", 1)
    for approx in (False, True):
        p = make_params(min_order=0, max_order=2, approx=approx)
        for _ in range(5):
            ctx.launch(p, tx)
        ctx.synchronize()
        lone = []
        for _ in range(30):
            ctx.launch(p, tx)
            lone.append(ctx.last_kernel_ms())  # (waits for the launch)
        piped = []
        for _ in range(6):
            for _ in range(20):
                ctx.launch(p, tx)
            piped.append(ctx.last_kernel_ms())
        print(f"approx={approx}: sweep kernel alone {np.mean(lone):.4f} ms (min {np.min(lone):.4f}); last of 20 pipelined launches {np.mean(piped):.4f} ms")
