#!/usr/bin/env python3
"""Diagnostic (GPU box): which candidate makes a cell's gradient NaN in a `fuzz_parity.py --grad` case -- the exhaustive kernel
with the candidate mask narrowed to one object at a time (order-1 cases), the traced path of that candidate, and the oracle's
verdict.  usage: diag_nan_case.py <seed> <case> <row> <col>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

from fuzz_parity import random_case  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402

F = np.float32
seed, target, r, c = (int(v) for v in sys.argv[1:5])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    while True:
        walls, tx, X, Y, kw, allowed = random_case(rng)
        if len(walls):
            break
    if kw["max_order"] == 3 and X.size > 1600:
        X, Y = X[:40, :40], Y[:40, :40]
    kw["fun"] = str(rng.choice(["received_power", "one", "length", "length_squared"]))
role = L.GRID_TX if target % 3 == 2 else L.GRID_RX
Xc, Yc = X[r:r + 1, c:c + 1], Y[r:r + 1, c:c + 1]
print("case", target, kw, "cell", Xc[0, 0], Yc[0, 0], "tx", tx)
with Context(0) as ctx:
    ctx.set_scene(walls)
    for j in range(len(walls)):
        mask = np.zeros(len(walls), np.uint8)
        mask[j] = 1
        if allowed is not None and not allowed[j]:
            continue
        ctx.set_candidate_mask(mask)
        out = {s: ctx.value_and_grads(tx, Xc, Yc, strict_nan=s, grid_role=role, **kw) for s in (True, False)}
        v, g = CO.power_map_grad(walls, tx, Xc, Yc, allowed=mask, grid_role="tx" if role == L.GRID_TX else "rx", **kw)
        flag = "  <== differs" if np.isnan(out[True]["grad_rx"]).any() != np.isnan(g).any() else ""
        print(f"object {j} {walls[j].reshape(-1)}: exhaustive {out[True]['grad_rx'][0, 0]} culled {out[False]['grad_rx'][0, 0]} oracle {g[0, 0]} value {out[True]['value'][0, 0]!r} / {v[0, 0]!r}{flag}")
        if flag:
            p = make_params(**{**kw, "min_order": 0, "max_order": 4})
            a, b = (np.stack([Xc[0], Yc[0]], -1), tx[None]) if role == L.GRID_TX else (tx[None], np.stack([Xc[0], Yc[0]], -1))
            tr = ctx.trace_paths(p, a, b, [np.array([j], np.int32)])
            print("     traced:", {k: (v_[0, 0][:3] if k == "xys" else v_[0, 0]) for k, v_ in tr.items()})
