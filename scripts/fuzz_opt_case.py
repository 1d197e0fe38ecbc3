#!/usr/bin/env python3
"""One case of scripts/fuzz_opt.py again: fuzz_opt_case.py <seed> <case> [row col] -- the oracle's side runs anywhere (C jets in
fp32 / fp64, and reverse-mode autodiff of oracle/ref.py through the Adam loop under torch, fp32 / fp64); the GPU's (both
gradient kernels) if there is one."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

from fuzz_opt import random_case  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import ref as R  # noqa: E402

F = np.float32
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    kinds, xys, phis, fixed, X, Y, kw, lo, hi, cands, theta0 = random_case(rng)
role = "tx" if target % 3 == 2 else "rx"
th = [t[: sum(kinds[int(i)] != 2 for i in c)] for c, t in zip(cands, theta0)]
print("case", target, "kinds", kinds.tolist(), "grid", X.shape, "orders", lo, hi, kw, "role", role, "candidates", len(cands))
okw = {k: v for k, v in kw.items() if k != "steps"}
cond = CO.opt_conditioning(kinds, xys, phis, fixed, X, Y, cands, th, kw["steps"], with_grad=True, grid_role=role, **okw)
got = None
try:
    from differt2d_amd.engine import Context

    with Context(0) as ctx:
        ctx.set_scene(xys, kinds, phis)
        ctx.set_theta0(theta0)
        got = {}
        for m in (0, 1):
            ctx.set_option("opt_grad_mode", m)
            got[m] = ctx.value_and_grads(fixed, X, Y, min_order=lo, max_order=hi, grid_role=L.GRID_TX if role == "tx" else L.GRID_RX, **kw)
except Exception as e:  # noqa: BLE001
    print("(no GPU side:", type(e).__name__, ")")
cells = [(int(sys.argv[3]), int(sys.argv[4]))] if len(sys.argv) > 4 else []
rkw = dict(solver=kw["solver"], steps=kw["steps"], approx=kw["approx"], alpha=kw["alpha"], tol=kw["tol"], patch=kw["patch"], fun=kw["fun"], grid_role=role)
if kw["approx"]:
    rkw["function"] = kw["function"]
for r, c in cells:
    print(f"cell ({r}, {c}): stable {bool(cond['stable'][r, c])} value32 {cond['value32'][r, c]!r} value64 {cond['value64'][r, c]!r}")
    print(f"    C jets: grad32 {cond['grad32'][r, c]} grad64 {cond['grad64'][r, c]}")
    for dt in ("float64", "float32"):
        w = R.opt_value_and_grads(kinds, np.asarray(xys, np.float64), phis, fixed, X[r:r + 1, c:c + 1], Y[r:r + 1, c:c + 1], cands, th, dtype=dt, **rkw)
        print(f"    ref.py reverse mode {dt}: value {float(w['value'][0, 0])!r} grad {np.asarray(w['grad_cell'][0, 0], np.float64)}")
    if got is not None:
        for m in (0, 1):
            print(f"    GPU opt_grad_mode {m}: value {got[m]['value'][r, c]!r} grad {got[m]['grad_rx'][r, c].astype(np.float64)}")
