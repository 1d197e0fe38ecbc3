#!/usr/bin/env python3
"""Stress of the cut-in-four path (cross-workgroup hand-over through global memory): many repeated launches of the bench
workload, every map compared bit for bit with the first."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tx, walls, X, Y = workload()
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    bad = 0
    for approx in (False, True):
        p = make_params(max_order=2, approx=approx)
        ctx.set_option("heavy_split", 0)
        ctx.launch(p, tx); ref = ctx.get_map()
        for h in (64, 512):
            ctx.set_option("heavy_split", h)
            for i in range(n):
                ctx.launch(p, tx)
                if i % 3 == 0 or i < 5:
                    got = ctx.get_map()
                    if not np.array_equal(got, ref):
                        bad += 1
                        print("MISMATCH approx", approx, "heavy", h, "launch", i, int((got != ref).sum()), "cells")
    print(f"stress: {bad} mismatching maps")
    sys.exit(1 if bad else 0)
