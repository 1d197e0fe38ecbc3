#!/usr/bin/env python3
"""Diagnostic: rows of cfg3's value+grad sweep against the C gradient oracle (tests/test_gpu_grad.py::test_cfg3_rows_...): offenders.
usage: diag_rows.py role mode [all]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_scene
from differt2d_amd import _lib as L
from differt2d_amd.engine import Context
from oracle import c_oracle as CO
F = np.float32
role, mode = sys.argv[1], sys.argv[2]
kw = {"hard": dict(approx=False), "hsig": dict(approx=True, function="hard_sigmoid"), "sigmoid": dict(approx=True, function="sigmoid")}[mode]
tx, walls = random_scene(50, seed=1234)
x = np.linspace(0.0, 1.0, 1024).astype(F)
X, Y = np.meshgrid(x, x)
i0 = min(max(int(tx[1] * 1023) - 32, 0), 1024 - 64)
rows = np.unique(np.concatenate([np.arange(i0, i0 + 64), np.arange(0, 1024, 16)]))
if len(sys.argv) > 3 and sys.argv[3] == "all":
    rows = np.arange(1024)
ctx = Context(0)
c = ctx
c.set_scene(walls)
got = c.value_and_grads(tx, X, Y, min_order=0, max_order=2, grid_role=L.GRID_TX if role == "tx" else L.GRID_RX, **kw)
value, grad, gabs, kink = CO.power_map_grad(walls, tx, X[rows], Y[rows], min_order=0, max_order=2, prune=1, grid_role=role, with_gabs=True, with_kink=True, **kw)
g = got["grad_rx"][rows].astype(np.float64)
print("values equal", np.array_equal(got["value"][rows], value), "NaN equal", np.array_equal(np.isnan(g), np.isnan(grad)))
fin = ~np.isnan(grad)
err = np.abs(g - grad)
rowscale = np.nanmax(np.abs(grad), axis=(1, 2), keepdims=True)
bar = 1e-5 * gabs[..., None] + 1e-5 * np.abs(grad) + 1e-6 * rowscale
r = np.where(fin, err / bar, 0.0)
bad = np.argwhere(r.max(-1) > 1)
print(f"{role} {mode}: {rows.size} rows, {int(np.isnan(grad).any(-1).sum())} NaN cells, {int((gabs > 0).sum())} cells with a path")
print(len(bad), "cells beyond the bar; worst", r.max(), "; kink cells among them", int(kink[tuple(bad.T)].sum()) if len(bad) else 0)
for b in bad[:12]:
    b = tuple(b)
    print("  row", rows[b[0]], "col", b[1], "gpu", g[b], "oracle", grad[b], "gabs", gabs[b], "rowscale", rowscale[b[0], 0, 0], "kink", kink[b], "value", value[b])
    # the cell alone: the exhaustive kernel, the culled kernel, the plain oracle (prune = 0, every fold followed)
    r_, c_ = rows[b[0]], b[1]
    Xc, Yc = X[r_:r_ + 1, c_:c_ + 1], Y[r_:r_ + 1, c_:c_ + 1]
    rk = dict(min_order=0, max_order=2, grid_role=L.GRID_TX if role == "tx" else L.GRID_RX, **kw)
    ex = ctx.value_and_grads(tx, Xc, Yc, strict_nan=True, **rk)["grad_rx"][0, 0]
    cu = ctx.value_and_grads(tx, Xc, Yc, strict_nan=False, **rk)["grad_rx"][0, 0]
    _, g00, k00 = CO.power_map_grad(walls, tx, Xc, Yc, min_order=0, max_order=2, prune=0, grid_role=role, with_kink=True, **kw)
    print("      alone: exhaustive", ex, "culled", cu, "plain oracle", g00[0, 0], "kink", bool(k00[0, 0]))
    for nm, (t2, X2, Y2) in {"tx+1ulp": (np.nextafter(tx, F(np.inf)), Xc, Yc), "cell+1ulp": (tx, np.nextafter(Xc, F(np.inf)), np.nextafter(Yc, F(np.inf))),
                             "cell-1ulp": (tx, np.nextafter(Xc, F(-np.inf)), np.nextafter(Yc, F(-np.inf)))}.items():
        _, g2 = CO.power_map_grad(walls, t2, X2, Y2, min_order=0, max_order=2, prune=1, grid_role=role, **kw)
        print("      oracle", nm, g2[0, 0])
v0, g0, ga0, k0 = CO.power_map_grad(walls, tx, X[rows[30:34]], Y[rows[30:34]], min_order=0, max_order=2, prune=0, grid_role=role, with_gabs=True, with_kink=True, **kw)
d_or = np.nan_to_num(np.abs(g0 - grad[30:34])); d_gpu = np.nan_to_num(np.abs(g0 - g[30:34]))
b = bar[30:34]
print("unpruned oracle vs pruned: cells beyond the bar", int((d_or > b).any(-1).sum()), "; vs GPU", int((d_gpu > b).any(-1).sum()), "; kink cells in these rows", int(k0.sum()))
for c in np.argwhere((d_gpu > b).any(-1))[:10]:
    c = tuple(c)
    print("  row", rows[30 + c[0]], "col", c[1], "unpruned", g0[c], "pruned", grad[30 + c[0], c[1]], "gpu", g[30 + c[0], c[1]], "gabs", ga0[c], "kink", k0[c], "bar", b[c])
