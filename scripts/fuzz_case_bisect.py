#!/usr/bin/env python3
"""Shrinks the candidate mask of a failing TX-grid fuzz case while the culled sweep still differs from the exhaustive one.
usage: fuzz_case_bisect.py <seed> <case> [big]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

from fuzz_parity import random_case  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context  # noqa: E402

seed, target = int(sys.argv[1]), int(sys.argv[2])
big = "big" in sys.argv[3:]
rng = np.random.default_rng(seed)
for case in range(target + 1):
    walls, tx, X, Y, kw, allowed = random_case(rng, big=big and case % 2 == 1)
role = L.GRID_TX if target % 3 == 2 else L.GRID_RX
with Context(0) as ctx:
    ctx.set_option("hidden_min_tiles", 0)
    ctx.set_scene(walls)

    def differs(mask):
        ctx.set_candidate_mask(mask)
        ctx.set_option("txg_exhaustive", 0)
        a_ = ctx.power_map(tx, X, Y, grid_role=role, **kw)
        ctx.set_option("txg_exhaustive", 1)
        b_ = ctx.power_map(tx, X, Y, grid_role=role, **kw)
        ctx.set_option("txg_exhaustive", 0)
        return int((a_ != b_).sum()), a_, b_

    cur = np.ones(len(walls), np.uint8) if allowed is None else np.asarray(allowed, np.uint8).copy()
    n, _, _ = differs(cur)
    print("start:", int(cur.sum()), "walls allowed,", n, "cells differ")
    changed = True
    while changed and cur.sum() > 2:
        changed = False
        idx = np.flatnonzero(cur)
        for chunk in np.array_split(idx, min(len(idx), 8)):
            trial = cur.copy()
            trial[chunk] = 0
            if trial.sum() >= 2 and differs(trial)[0] > 0:
                cur = trial
                changed = True
                break
    n, a_, b_ = differs(cur)
    d = np.argwhere(a_ != b_)
    print("minimal mask:", np.flatnonzero(cur).tolist(), "cells differing", n, d[:6].tolist())
    for r, c in d[:4]:
        print("  ", r, c, "culled", a_[r, c], "exhaustive", b_[r, c])
    print("kw", kw, "tx", tx.tolist(), "grid", X.shape, "x", X[0, [0, -1]].tolist(), "y", Y[[0, -1], 0].tolist())
    for w in np.flatnonzero(cur):
        print("   wall", int(w), walls[w].tolist())
