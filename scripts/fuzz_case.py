#!/usr/bin/env python3
"""One case of scripts/fuzz_parity.py again, with options: fuzz_case.py <seed> <case> [big] [name=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

from fuzz_parity import random_case  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402

seed, target = int(sys.argv[1]), int(sys.argv[2])
big = "big" in sys.argv[3:]
opts = [a.split("=") for a in sys.argv[3:] if "=" in a]
rng = np.random.default_rng(seed)
for case in range(target + 1):
    walls, tx, X, Y, kw, allowed = random_case(rng, big=big and case % 2 == 1)
case = target
role_tx = case % 3 == 2
want = CO.power_map(walls, tx, X, Y, allowed=allowed, prune=True, grid_role="tx" if role_tx else "rx", **kw)
want0 = CO.power_map(walls, tx, X, Y, allowed=allowed, prune=False, grid_role="tx" if role_tx else "rx", **kw)
print("case", case, "N", len(walls), "grid", X.shape, kw, "role_tx", role_tx, "| oracle prune vs plain differ:", int((want != want0).sum()))
with Context(0) as ctx:
    ctx.set_option("hidden_min_tiles", 0)
    ctx.set_scene(walls)
    ctx.set_candidate_mask(allowed)
    base = {"split_max_tiles": -1 if case % 5 == 4 else (8192 if case % 2 == 0 else 0), "coop_waves": -1 if case % 5 == 4 else 0, "split_sigmoid": 1,
            "sched_min_tiles": 1 if case % 4 < 2 else 1 << 40, "region_lists": 0 if case % 7 == 6 else 1, "region_size": (4, 2, 1)[case % 3],
            "region_size_top": (16, 4, 3)[(case // 3) % 3]}
    for variant in [{}] + [{k: int(v)} for k, v in opts]:
        for k, v in {**base, **variant}.items():
            ctx.set_option(k, v)
        got = ctx.power_map(tx, X, Y, grid_role=L.GRID_TX if role_tx else L.GRID_RX, **kw)
        again = ctx.power_map(tx, X, Y, grid_role=L.GRID_TX if role_tx else L.GRID_RX, **kw)
        d = np.argwhere(~np.isclose(got, want, rtol=2e-5, atol=1e-5 * max(1.0, float(np.nanmax(np.abs(want)))), equal_nan=True))
        print(variant, "cells beyond tolerance:", len(d), "| bit-different:", int((got != want).sum()), "| second launch differs:", int((got != again).sum()), "| max |want|", float(np.nanmax(np.abs(want))))
        for r, c in d[:6]:
            print("   ", r, c, "got", got[r, c], "want", want[r, c], "plain oracle", want0[r, c])
