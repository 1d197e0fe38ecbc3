#!/usr/bin/env python3
"""Runs BASELINE.json configs[3] at full size on the GPU (200 walls, 2048^2 cells, orders 0..3) and reports the time;
with tests/golden/cfg4_samples_*.npz present, also compares the sampled cells bit for bit."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_scene  # noqa: E402
from differt2d_amd.engine import Context  # noqa: E402

F = np.float32
g = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
max_order = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tx, walls = random_scene(200, seed=1234)
x = np.linspace(0.0, 1.0, 2048).astype(F)[:g]
X, Y = np.meshgrid(x, x)
ctx = Context(0)
ctx.set_scene(walls)
for name, mode in (("hard", dict(approx=False)), ("hsig", dict(approx=True, function="hard_sigmoid"))):
    t = time.time()
    out = ctx.power_map(tx, X, Y, min_order=0, max_order=max_order, **mode)
    dt = time.time() - t
    C = sum(200 * 199 ** (k - 1) if k else 1 for k in range(max_order + 1))
    print(f"{name}: {g}x{g} K<={max_order}: {dt:.2f} s  -> {C * g * g / dt:.3e} cand/s  sum={out.astype(np.float64).sum():.6f}", flush=True)
    path = os.path.join(ROOT, "tests", "golden", f"cfg4_samples_{name}.npz")
    if os.path.exists(path) and g == 2048 and max_order == 3:
        z = np.load(path)
        ij = z["ij"]
        got = out[ij[:, 0], ij[:, 1]]
        print("  sampled cells bit-equal:", np.array_equal(got, z["total"]), float(np.abs(got - z["total"]).max()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", f"cfg4_{name}_{g}_{max_order}.npy"), out[::8, ::8])
