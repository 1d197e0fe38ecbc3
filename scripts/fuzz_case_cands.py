#!/usr/bin/env python3
"""Which candidates of a fuzz case's cell make up the difference between the sweep and the oracle: the literal trace kernel's
validity of every candidate for that cell, the ones near the missing amount first.
usage: fuzz_case_cands.py <seed> <case> <row> <col> [big]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

from fuzz_parity import random_case  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402
from differt2d_amd.scene import all_path_candidates  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402

seed, target, r, c = (int(v) for v in sys.argv[1:5])
big = "big" in sys.argv[5:]
rng = np.random.default_rng(seed)
for case in range(target + 1):
    walls, tx, X, Y, kw, allowed = random_case(rng, big=big and case % 2 == 1)
role_tx = target % 3 == 2
nodes = None if allowed is None else [int(i) for i in np.flatnonzero(np.asarray(allowed) == 0)]  # (the nodes never visited)
cands = all_path_candidates(len(walls), kw["min_order"], kw["max_order"], filter_nodes=nodes)
cell = np.array([X[r, c], Y[r, c]], np.float32)
with Context(0) as ctx:
    ctx.set_option("hidden_min_tiles", 0)
    ctx.set_scene(walls)
    ctx.set_candidate_mask(allowed)
    got = ctx.power_map(tx, X, Y, grid_role=L.GRID_TX if role_tx else L.GRID_RX, **kw)
    want = CO.power_map(walls, tx, X, Y, allowed=allowed, prune=True, grid_role="tx" if role_tx else "rx", **kw)
    miss = float(want[r, c]) - float(got[r, c])
    print("cell", r, c, cell, "got", got[r, c], "want", want[r, c], "missing", miss, "candidates", len(cands))
    p = {k: v for k, v in kw.items() if k not in ("min_order", "max_order", "fun", "height")}
    t, rx = (cell, tx) if role_tx else (tx, cell)
    out = ctx.trace_paths(make_params(max_order=L.D2D_MAX_ORDER, **p), t[None], rx[None], cands)
    v = out["valid"][0].astype(np.float64)
    print("sum of the traced validities", v.sum(), "(fun = one)")
    order = np.argsort(np.abs(v - miss))[:8]
    for i in order:
        print("  candidate", i, [int(x) for x in cands[i]], "valid", v[i], "on", out["on"][0, i], "hit", out["hit"][0, i], "xys", out["xys"][0, i, : len(cands[i]) + 2].round(4).tolist())
    # which of them does the culled sweep lose?  Each candidate alone (candidate mask = its walls), culled against exhaustive
    if role_tx:
        for i in np.argsort(np.abs(v - miss))[:40]:
            m = np.zeros(len(walls), np.uint8)
            m[[int(x) for x in cands[i]]] = 1
            ctx.set_candidate_mask(m)
            ctx.set_option("txg_exhaustive", 0)
            a_ = ctx.power_map(tx, X, Y, grid_role=L.GRID_TX, **kw)
            ctx.set_option("txg_exhaustive", 1)
            b_ = ctx.power_map(tx, X, Y, grid_role=L.GRID_TX, **kw)
            ctx.set_option("txg_exhaustive", 0)
            if not np.array_equal(a_, b_, equal_nan=True):
                d = np.argwhere(a_ != b_)
                print("  LOST: candidate", i, [int(x) for x in cands[i]], "cells", d[:8].tolist(), "culled", a_[tuple(d[0])], "exhaustive", b_[tuple(d[0])])
