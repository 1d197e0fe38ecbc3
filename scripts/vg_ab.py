#!/usr/bin/env python3
"""A/B of the value+grad step at cfg3 (GPU box): the NaN scan behind the sweep (round 4) or beside it on a stream of its own
(lowest / highest priority), both grid roles, hard and hard_sigmoid; results must be identical.
usage: python scripts/vg_ab.py [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
tx, walls, X, Y = workload()
with Context(0) as ctx:
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    for role in (L.GRID_RX, L.GRID_TX):
        for mode in (dict(approx=False), dict(approx=True), dict(approx=True, function="sigmoid")):
            p = make_params(min_order=0, max_order=2, grid_role=role, **mode)
            ref = None
            n = steps if mode.get("function") != "sigmoid" else max(3, steps // 10)
            for label, opts in (("behind", dict(nan_scan_async=0)), ("beside, low prio", dict(nan_scan_async=1, nan_scan_prio=0)),
                                ("beside, high prio", dict(nan_scan_async=1, nan_scan_prio=1)), ("no scan", dict(nan_scan=0))):
                ctx.set_option("nan_scan", 1)
                for k, v in opts.items():
                    ctx.set_option(k, v)
                for _ in range(3):
                    ctx.launch_vg(p, tx, scene_vjp=True)
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    ctx.launch_vg(p, tx, scene_vjp=True)
                ctx.synchronize()
                ms = (time.perf_counter() - t0) / n * 1e3
                g = ctx.get_grad_rx()
                txb, wb = ctx.get_scene_vjp()
                same = ""
                if label == "behind":
                    ref = (g, txb, wb)
                elif label != "no scan":
                    same = " identical to 'behind': %s" % (np.array_equal(g, ref[0], equal_nan=True) and np.array_equal(np.isnan(wb), np.isnan(ref[2]))
                                                           and np.array_equal(np.isnan(txb), np.isnan(ref[1])))
                print(f"{'TX' if role == L.GRID_TX else 'RX'} grid {mode}: {label:18s} {ms:7.3f} ms per step{same}", flush=True)
    ctx.set_option("nan_scan", 1)
