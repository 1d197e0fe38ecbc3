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
# a different transmitter every launch: a part's list read stale (the previous launch's bytes at the same addresses, from a
# cache that should have been bypassed) then differs from what this launch wrote -- with one transmitter it would not
txs = [tx, (tx + np.float32([0.013, -0.021])).astype(np.float32), (tx + np.float32([-0.3, 0.25])).astype(np.float32)]
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    bad = 0
    for approx in (False, True):
        p = make_params(max_order=2, approx=approx)
        ctx.set_option("heavy_split", 0)
        refs = []
        for t in txs:
            ctx.launch(p, t); refs.append(ctx.get_map())
        for h in (-1, 64, 1024):
            ctx.set_option("heavy_split", h)
            for i in range(n):
                k = i % len(txs)
                ctx.launch(p, txs[k])
                if i % 2 == 0 or i < 8:
                    got = ctx.get_map()
                    if not np.array_equal(got, refs[k], equal_nan=True):
                        bad += 1
                        print("MISMATCH approx", approx, "heavy", h, "launch", i, int((got != refs[k]).sum()), "cells")
        # back to back without any host synchronisation, a different transmitter each launch: the launches
        # accumulate into the map (D2D_OUT_ADD), so one read-back at the end observes every one of them -- a part's list
        # read stale (the previous launch's bytes at the same addresses) would change the sum
        import differt2d_amd._lib as L
        p_add = make_params(max_order=2, approx=approx, out_mode=L.OUT_ADD)
        for h in (-1, 64, 1024):
            ctx.set_option("heavy_split", h)
            for burst in (16, 61):
                ctx.launch(p, txs[0])  # overwrite: the sum starts from refs[0]
                want = refs[0].copy()
                for i in range(burst):
                    k = (i * 7 + 1) % len(txs)
                    ctx.launch(p_add, txs[k])
                    want = want + refs[k]  # fp32, in launch order
                got = ctx.get_map()
                if not np.array_equal(got, want, equal_nan=True):
                    bad += 1
                    print("MISMATCH (no host sync) approx", approx, "heavy", h, "burst", burst, int((got != want).sum()), "cells")
        print(f"approx={approx}: done", flush=True)
    print(f"stress: {bad} mismatching maps")
    sys.exit(1 if bad else 0)
