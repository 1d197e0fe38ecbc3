#!/usr/bin/env python3
"""Diagnostic: value+grad sweeps (default, exhaustive) against the C gradient oracle on the lattice scenes of
tests/test_gpu_grad.py::test_lattice_scenes_against_the_c_gradient_oracle.   usage: diag_lattice.py [case ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from differt2d_amd import _lib as L
from differt2d_amd.engine import Context
from oracle import c_oracle as CO
F = np.float32
want_cases = [int(a) for a in sys.argv[1:]] or list(range(24))
rng = np.random.default_rng(5)
with Context(0) as c:
    for case in range(24):
        n = int(rng.integers(3, 9))
        walls = (np.round(rng.random((n, 2, 2)) * 4) / 4).astype(F)
        walls[(walls[:, 0] == walls[:, 1]).all(-1)] += F(0.125)
        tx = (np.round(rng.random(2) * 8) / 8).astype(F)
        xs = np.linspace(0, 1, int(rng.integers(9, 34))).astype(F)
        X, Y = np.meshgrid(xs, xs[: int(rng.integers(5, xs.size + 1))])
        approx, function = [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")][case % 3]
        fun = ["received_power", "one", "length", "length_squared"][(case // 3) % 4]
        role = "tx" if case % 2 else "rx"
        kw = dict(min_order=0, max_order=2, approx=approx, function=function, fun=fun, alpha=float(rng.choice([100.0, 16.0])))
        if case not in want_cases:
            continue
        value, grad, gabs, kink, amp = CO.power_map_grad(walls, tx, X, Y, grid_role=role, with_gabs=True, with_kink=True, with_amp=True, **kw)
        up = lambda a: np.nextafter(np.asarray(a, F), F(np.inf))
        stable = np.ones(X.shape, bool)
        for tx2, X2, Y2 in ((up(tx), X, Y), (tx, up(X), up(Y))):
            v2, g2 = CO.power_map_grad(walls, tx2, X2, Y2, grid_role=role, **kw)
            with np.errstate(invalid="ignore"):
                stable &= np.abs(v2 - value) <= 1e-3 * np.abs(value) + 1e-9
                stable &= (np.abs(g2 - grad) <= 1e-2 * gabs[..., None] + 1e-9).all(-1)
        c.set_scene(walls)
        for strict in (False, True):
            got = c.value_and_grads(tx, X, Y, strict_nan=strict, grid_role=L.GRID_TX if role == "tx" else L.GRID_RX, **kw)
            g = got["grad_rx"].astype(np.float64)
            nan_eq = np.array_equal(np.isnan(g), np.isnan(grad))
            fin = np.isfinite(grad) & np.isfinite(g) & ~kink[..., None] & stable[..., None]
            bar = 1e-5 * gabs[..., None] + 1e-5 * np.abs(grad) + 1e-6
            r = np.where(fin, np.abs(g - grad) / bar, 0.0)
            bad = np.argwhere(r.max(-1) > 1.0)
            print(f"case {case} {'strict' if strict else 'default'} {role} {kw} walls {n} grid {X.shape}: NaN equal {nan_eq} ({int(np.isnan(g).any(-1).sum())} / {int(np.isnan(grad).any(-1).sum())}), "
                  f"value equal {np.array_equal(got['value'], value, equal_nan=True)}, {len(bad)} of {fin.all(-1).sum()} cells beyond the bar (kinks {int(kink.sum())}, unstable {int((~stable).sum())}), worst {r.max():.1f} x")
            for b in bad[:4]:
                b = tuple(b)
                print("    cell", b, "xy", X[b], Y[b], "gpu", g[b], "oracle", grad[b], "gabs", gabs[b], "value", value[b], "amp", amp[b])
