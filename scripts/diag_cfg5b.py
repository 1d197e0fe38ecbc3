#!/usr/bin/env python3
"""Diagnostic: which candidate / after how many Adam steps a cfg5 cell's gradient turns non-finite on the GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from differt2d_amd.engine import default_context  # noqa: E402

F = np.float32
z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz"))
xys, kind, phi, tx = z["xys"], z["kind"], z["phi"], z["tx"]
th = z["theta0"]
x = np.linspace(0.0, 1.0, 300).astype(F)
ctx = default_context()
ctx.set_scene(xys, kind, phi)
cells = [(182, 296), (182, 293), (182, 297), (0, 150), (150, 0), (100, 141), (299, 10), (10, 299)]
for (i, j) in cells:
    print("cell", i, j, "x", x[j], "y", x[i])
    for obj in range(7):
        mask = np.zeros(7, np.uint8)
        mask[obj] = 1
        ctx.set_candidate_mask(mask)
        t0 = np.array([th[obj] if np.isfinite(th[obj]) else 0.0, 0, 0, 0], F)
        ctx.set_theta0([t0])
        line = []
        for steps in (1, 10, 100, 300, 600, 1000):
            r = ctx.value_and_grads(tx, x[j][None, None], x[i][None, None], min_order=1, max_order=1, approx=True, solver="min", steps=steps)
            fin = all(np.isfinite(r[k]).all() for k in ("grad_rx", "tx_bar", "walls_bar", "phi_bar"))
            line.append(f"{steps}:{'ok' if fin else 'NAN'} v={float(r['value'][0,0]):.4g} g={r['grad_rx'][0,0,0]:.3g},{r['grad_rx'][0,0,1]:.3g}")
        print("  obj", obj, " | ".join(line))
    ctx.set_candidate_mask(None)
