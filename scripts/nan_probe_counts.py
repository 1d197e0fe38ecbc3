#!/usr/bin/env python3
"""Counters of the NaN scan at cfg3, both grid roles (GPU box): (patch, candidate) pairs probed cell by cell, NaN cells."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

tx, walls, X, Y = workload()
with Context(0) as ctx:
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    ctx.set_option("nan_scan_stats", 1)
    for role, name in ((L.GRID_RX, "rx"), (L.GRID_TX, "tx")):
        for mode in (dict(approx=False), dict(approx=True)):
            p = make_params(min_order=0, max_order=2, grid_role=role, **mode)
            ctx.launch_vg(p, tx, scene_vjp=True)
            ctx.synchronize()
            print(name, mode, ctx.debug_nan_scan(), flush=True)
