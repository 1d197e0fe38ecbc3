#!/usr/bin/env python3
"""How many cells of configs[2]'s gradient map does the oracle's shortcut (prune = 1: stop a candidate's occlusion fold at the first
occluder saturated to exactly 1 -- what the culled AND the exhaustive kernels' adjoint do) change against the plain oracle
(prune = 0: every fold of every candidate, JAX's tie rule at every minimum / maximum, relu6's kinks included)?  CPU only (the GPU's
gradients equal the pruned oracle's on all 1 024 rows up to the cells scripts/diag_rows.py lists: profiles/r06_parity_runs.txt).

    python scripts/plain_vs_pruned_oracle.py role mode [row_step] [nthreads]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_scene  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402

F = np.float32
role, mode = sys.argv[1], sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nthreads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
kw = {"hard": dict(approx=False), "hsig": dict(approx=True, function="hard_sigmoid"), "sigmoid": dict(approx=True, function="sigmoid")}[mode]
tx, walls = random_scene(50, seed=1234)
x = np.linspace(0.0, 1.0, 1024).astype(F)
X, Y = np.meshgrid(x, x)
rows = np.arange(0, 1024, step)
t0 = time.time()
tot = dict(cells=0, kink=0, over=0, over_not_kink=0, nan_diff=0, value_diff=0)
worst = []
for b in range(0, rows.size, 16):
    r = rows[b:b + 16]
    v1, g1, ga = CO.power_map_grad(walls, tx, X[r], Y[r], min_order=0, max_order=2, prune=1, grid_role=role, with_gabs=True, nthreads=nthreads, **kw)
    v0, g0, k0 = CO.power_map_grad(walls, tx, X[r], Y[r], min_order=0, max_order=2, prune=0, grid_role=role, with_kink=True, nthreads=nthreads, **kw)
    rowscale = np.nanmax(np.abs(g1), axis=(1, 2), keepdims=True)
    bar = 1e-5 * ga[..., None] + 1e-5 * np.abs(g1) + 1e-6 * rowscale
    with np.errstate(invalid="ignore"):
        over = ((np.abs(g0 - g1) > bar) & ~np.isnan(g0)).any(-1)
    tot["cells"] += over.size
    tot["kink"] += int(k0.sum())
    tot["over"] += int(over.sum())
    tot["over_not_kink"] += int((over & ~k0).sum())
    tot["nan_diff"] += int((np.isnan(g0) != np.isnan(g1)).any(-1).sum())
    tot["value_diff"] += int((~((v0 == v1) | (np.isnan(v0) & np.isnan(v1)))).sum())
    for i, j in np.argwhere(over):
        worst.append((float(np.abs(g0[i, j] - g1[i, j]).max() / rowscale[i, 0, 0]), int(r[i]), int(j), g0[i, j].tolist(), g1[i, j].tolist(), float(ga[i, j])))
    print(f"rows {r[0]}..{r[-1]}: {tot}  ({time.time() - t0:.0f} s)", flush=True)
worst.sort(reverse=True)
print(f"{role} {mode}: {tot}")
print("largest |plain - pruned| / (largest gradient of the row), row, col, plain, pruned, cell's gradient scale:")
for w in worst[:12]:
    print("  ", w)
