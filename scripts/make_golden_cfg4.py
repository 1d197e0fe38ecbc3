#!/usr/bin/env python3
"""Generates tests/golden/cfg4_samples_{hard,hsig}.npz: BASELINE.json configs[3] at FULL size (200 random walls, NumPy
seed 1234, orders 0..3 = 7 960 201 candidates per cell) on a sample of cells of the 2048 x 2048 grid, with the C oracle
(oracle/d2d_oracle.c, result-preserving pruning on).  About 10 s of CPU per cell and core, hence the sample.

Run from the repo root:  python scripts/make_golden_cfg4.py [n_cells]
"""

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import random_scene  # noqa: E402
from oracle import c_oracle  # noqa: E402

F = np.float32
n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tx, walls = random_scene(200, seed=1234)
x = np.linspace(0.0, 1.0, 2048).astype(F)
rng = np.random.default_rng(4)
# an eighth of the cells in the 8 x 8 patches around the transmitter (the patch holding it and its neighbours), three
# eighths within 60 cells of it (where high-order paths survive), half anywhere
itx, jtx = int(round(float(tx[1]) * 2047)), int(round(float(tx[0]) * 2047))
n_adj, n_near = n_cells // 8, 3 * n_cells // 8
adj = np.stack([np.clip(itx + rng.integers(-12, 13, n_adj), 0, 2047),
                np.clip(jtx + rng.integers(-12, 13, n_adj), 0, 2047)], 1)
near = np.stack([np.clip(itx + rng.integers(-60, 61, n_near), 0, 2047),
                 np.clip(jtx + rng.integers(-60, 61, n_near), 0, 2047)], 1)
far = rng.integers(0, 2048, (n_cells - n_adj - n_near, 2))
ij = np.concatenate([adj, near, far]).astype(np.int32)
X, Y = x[ij[:, 1]], x[ij[:, 0]]
for name, mode in (("hard", dict(approx=False)), ("hsig", dict(approx=True, function="hard_sigmoid"))):
    t = time.time()
    per_order = [c_oracle.power_map(walls, tx, X, Y, min_order=k, max_order=k, prune=True, **mode) for k in range(4)]
    total = c_oracle.power_map(walls, tx, X, Y, min_order=0, max_order=3, prune=True, **mode)
    path = os.path.join(ROOT, "tests", "golden", f"cfg4_samples_{name}.npz")
    np.savez_compressed(path, ij=ij, per_order=np.stack(per_order), total=total)
    print(path, f"{time.time() - t:.0f}s", "nonzero per order:", [(p != 0).sum() for p in per_order])
