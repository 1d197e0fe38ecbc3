#!/usr/bin/env python3
"""Diagnostic: stable cfg5 edge cells whose GPU gradient is off the oracle's by more than the bar."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from differt2d_amd.engine import default_context
F = np.float32
z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz"))
e = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_edges.npz"))
xys, kind, phi, tx, steps = z["xys"], z["kind"], z["phi"], z["tx"], int(z["steps"])
theta0 = [np.array([t, 0, 0, 0], F) if np.isfinite(t) else np.zeros(4, F) for t in z["theta0"]]
x = np.linspace(0.0, 1.0, 300).astype(F)
X, Y = np.meshgrid(x, x)
ctx = default_context()
ctx.set_scene(xys, kind, phi); ctx.set_theta0(theta0)
kw = dict(min_order=1, max_order=1, approx=True, solver="min", steps=steps)
for mode in (0, 1):
    ctx.set_option("opt_grad_mode", mode)
    full = ctx.value_and_grads(tx, X, Y, **kw)
    ij, st = e["ij"], e["stable"]
    g = full["grad_rx"][ij[:, 0], ij[:, 1]]
    g64, g32 = e["grad_cell64"], e["grad_cell32"]
    fin = np.isfinite(g64).all(-1)
    gs = np.maximum(np.abs(np.nan_to_num(g64)).max(-1), np.median(np.abs(g64[fin]).max(-1)))[:, None]
    err, ref = np.abs(g - g64) / gs, np.abs(g32 - g64) / gs
    bad = st & (err > np.maximum(1e-5, 2 * ref)).any(-1)
    print("mode", mode, "bad cells", int(bad.sum()))
    for c in np.where(bad)[0]:
        print("  ", ij[c].tolist(), "x", x[ij[c, 1]], "y", x[ij[c, 0]], "gpu", g[c], "o64", g64[c], "o32", g32[c], "err", err[c], "ref", ref[c], "scale", gs[c])
