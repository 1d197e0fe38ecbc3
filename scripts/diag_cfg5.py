#!/usr/bin/env python3
"""Diagnostic: cfg5 (tests/golden/cfg5_samples.npz) -- the GPU's value / per-cell gradient / scene VJP of every sampled
cell on its own (1 x 1 grids), so that a NaN or an outlier in the summed VJP can be traced to its cell.
Writes gpurun_out/diag_cfg5.npz.  Run on the GPU box:  python scripts/diag_cfg5.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from differt2d_amd.engine import default_context  # noqa: E402

F = np.float32
z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz"))
xys, kind, phi, tx, ij, steps = z["xys"], z["kind"], z["phi"], z["tx"], z["ij"], int(z["steps"])
theta0 = [np.array([t, 0, 0, 0], F) if np.isfinite(t) else np.zeros(4, F) for t in z["theta0"]]
x = np.linspace(0.0, 1.0, 300).astype(F)
ctx = default_context()
ctx.set_scene(xys, kind, phi)
ctx.set_theta0(theta0)
kw = dict(min_order=1, max_order=1, approx=True, solver="min", steps=steps)
out = {k: [] for k in ("value", "grad_rx", "tx_bar", "walls_bar", "phi_bar")}
for c in range(len(ij)):
    r = ctx.value_and_grads(tx, x[ij[c, 1]][None, None], x[ij[c, 0]][None, None], **kw)
    for k in out:
        out[k].append(np.asarray(r[k]))
    bad = [k for k in ("tx_bar", "walls_bar", "phi_bar", "grad_rx") if not np.isfinite(r[k]).all()]
    print(c, ij[c].tolist(), "stable" if z["stable"][c] else "unstable", float(r["value"][0, 0]), r["grad_rx"][0, 0].tolist(), "NONFINITE " + ",".join(bad) if bad else "")
X, Y = np.meshgrid(x, x)
full = ctx.value_and_grads(tx, X, Y, **kw)
nanc = np.argwhere(~np.isfinite(full["grad_rx"]).all(-1))
print("full map: non-finite gradient cells", len(nanc), "columns", sorted(set(nanc[:, 1].tolist()))[:20], "x of those", [float(x[c]) for c in sorted(set(nanc[:, 1].tolist()))[:20]])
print("full VJP finite:", {k: bool(np.isfinite(full[k]).all()) for k in ("tx_bar", "walls_bar", "phi_bar")})
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", "diag_cfg5.npz"), **{k: np.stack(v) for k, v in out.items()}, full_value=full["value"],
         full_grad=full["grad_rx"], full_tx_bar=full["tx_bar"], full_walls_bar=full["walls_bar"], full_phi_bar=full["phi_bar"])
