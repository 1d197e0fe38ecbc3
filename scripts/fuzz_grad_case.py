#!/usr/bin/env python3
"""One case of `scripts/fuzz_parity.py --grad` again: fuzz_grad_case.py <seed> <case> [row col]  -- the oracle's side runs
anywhere (C duals, and fp64 / fp32 reverse-mode autodiff of oracle/ref.py for the cell given); the GPU's if there is one."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

from fuzz_parity import random_case  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402

F = np.float32
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    while True:
        walls, tx, X, Y, kw, allowed = random_case(rng)
        if len(walls):
            break
    if kw["max_order"] == 3 and X.size > 1600:
        X, Y = X[:40, :40], Y[:40, :40]
    kw["fun"] = str(rng.choice(["received_power", "one", "length", "length_squared"]))
role = "tx" if target % 3 == 2 else "rx"
strict = target % 4 == 3
print("case", target, "N", len(walls), "grid", X.shape, kw, "role", role, "strict", strict, "allowed", None if allowed is None else allowed.tolist())
okw = dict(kw, grid_role=role, allowed=allowed)
value, grad, gabs, kink, amp = CO.power_map_grad(walls, tx, X, Y, with_gabs=True, with_kink=True, with_amp=True, **okw)
cells = [(int(sys.argv[3]), int(sys.argv[4]))] if len(sys.argv) > 4 else []
got = None
try:
    from differt2d_amd import _lib as L
    from differt2d_amd.engine import Context

    with Context(0) as ctx:
        ctx.set_scene(walls)
        ctx.set_candidate_mask(allowed)
        got = {s: ctx.value_and_grads(tx, X, Y, strict_nan=s, grid_role=L.GRID_TX if role == "tx" else L.GRID_RX, **kw) for s in (False, True)}
except Exception as e:  # noqa: BLE001
    print("(no GPU side:", type(e).__name__, ")")
if got is not None:
    for s_ in (False, True):
        gn, on = np.isnan(got[s_]["grad_rx"]).any(-1), np.isnan(grad).any(-1)
        d = np.argwhere(gn != on)
        print(f"NaN cells: GPU {'exhaustive' if s_ else 'culled'} {int(gn.sum())}, oracle {int(on.sum())}; differing: {d[:12].tolist()}")
        for r_, c_ in d[:4]:
            print(f"    ({r_}, {c_}) = ({X[r_, c_]!r}, {Y[r_, c_]!r}): GPU {got[s_]['grad_rx'][r_, c_]} oracle {grad[r_, c_]} value {value[r_, c_]!r}")
if got is not None and not cells:
    g = got[strict]["grad_rx"].astype(np.float64)
    rel = 3e-4 if kw["approx"] and kw["function"] == "sigmoid" else 1e-5
    bar = rel * gabs[..., None] + rel * np.abs(grad) + 1e-6 * float(np.nanmax(np.abs(grad), initial=0.0)) + 1e-30
    bad = np.argwhere((np.abs(g - grad) > bar).any(-1) & np.isfinite(grad).all(-1) & np.isfinite(g).all(-1))
    cells = [tuple(b) for b in bad[:8]]
    print(len(bad), "cells beyond the bar")
from oracle import ref as R  # noqa: E402

for r, c in cells:
    print(f"cell ({r}, {c}) = ({X[r, c]!r}, {Y[r, c]!r}): value {value[r, c]!r} C-dual grad {grad[r, c]} gabs {gabs[r, c]:.4e} kink {bool(kink[r, c])} amp {amp[r, c]:.3e}")
    Xc, Yc = X[r:r + 1, c:c + 1], Y[r:r + 1, c:c + 1]
    rkw = {k: v for k, v in kw.items() if k != "height"}
    if kw["fun"] == "received_power":
        rkw["fun_kwargs"] = dict(height=kw["height"])
    if allowed is not None:
        rkw["filter_nodes"] = [i for i in range(len(walls)) if not allowed[i]]
    for dt in ("float64", "float32"):
        w = R.power_map_value_and_grads(walls, tx, Xc, Yc, dtype=dt, grid_role=role, **rkw)
        print(f"    ref.py reverse mode {dt}: value {float(w['value'][0, 0])!r} grad {np.asarray(w['grad_rx'][0, 0], np.float64)}")
    _, g0, k0 = CO.power_map_grad(walls, tx, Xc, Yc, with_kink=True, **dict(okw, prune=0))
    print(f"    C-dual plain (prune 0): {g0[0, 0]} kink {bool(k0[0, 0])}")
    up = lambda a: np.nextafter(np.asarray(a, F), F(np.inf))  # noqa: E731
    for name, (t2, X2, Y2) in {"tx+1ulp": (up(tx), Xc, Yc), "cell+1ulp": (tx, up(Xc), up(Yc))}.items():
        _, g2 = CO.power_map_grad(walls, t2, X2, Y2, **okw)
        print(f"    C-dual {name}: {g2[0, 0]}")
    if got is not None:
        for s in (False, True):
            print(f"    GPU {'exhaustive' if s else 'culled'}: value {got[s]['value'][r, c]!r} grad {got[s]['grad_rx'][r, c].astype(np.float64)}")
