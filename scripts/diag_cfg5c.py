#!/usr/bin/env python3
"""Diagnostic: the reverse-mode kernel's non-finite gradient cells at cfg5 vs tests/golden/cfg5_edges.npz, and which candidate
/ after how many steps they appear."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from differt2d_amd.engine import default_context
F = np.float32
z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz"))
e = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_edges.npz"))
xys, kind, phi, tx, steps = z["xys"], z["kind"], z["phi"], z["tx"], int(z["steps"])
th = z["theta0"]
theta0 = [np.array([t, 0, 0, 0], F) if np.isfinite(t) else np.zeros(4, F) for t in th]
x = np.linspace(0.0, 1.0, 300).astype(F)
X, Y = np.meshgrid(x, x)
ctx = default_context()
ctx.set_scene(xys, kind, phi); ctx.set_theta0(theta0)
kw = dict(min_order=1, max_order=1, approx=True, solver="min", steps=steps)
full = ctx.value_and_grads(tx, X, Y, **kw)
bad = ~np.isfinite(full["grad_rx"]).all(-1)
print("non-finite cells", int(bad.sum()))
for r in range(0, 300, 6):
    print(''.join('#' if bad[r:r+6, c:c+3].any() else '.' for c in range(0, 300, 3)))
ij = e["ij"]
gb = bad[ij[:, 0], ij[:, 1]]
ob = ~e["grad_finite"]
print("edge cells: GPU non-finite", int(gb.sum()), "oracle non-finite", int(ob.sum()), "both", int((gb & ob).sum()))
print("GPU only:", ij[gb & ~ob][:40].tolist())
print("oracle only:", ij[~gb & ob].tolist())
v = full["value"][ij[:, 0], ij[:, 1]]
print("value vs oracle fp32 on edges: max abs diff", np.nanmax(np.abs(v - e["value32"])), "scale", np.abs(e["value32"]).max())
cells = [tuple(c) for c in ij[gb & ~ob][:6].tolist()] + [tuple(c) for c in ij[~gb & ob].tolist()] + [tuple(c) for c in np.argwhere(bad)[::40][:6].tolist()]
for (i, j) in cells:
    print("cell", i, j, "x", x[j], "y", x[i], "oracle grad", e["grad_cell32"][(ij[:, 0] == i) & (ij[:, 1] == j)].tolist())
    for obj in range(7):
        mask = np.zeros(7, np.uint8); mask[obj] = 1
        ctx.set_candidate_mask(mask)
        ctx.set_theta0([np.array([th[obj] if np.isfinite(th[obj]) else 0.0, 0, 0, 0], F)])
        line = []
        for st in (1, 10, 100, 1000):
            r = ctx.value_and_grads(tx, x[j][None, None], x[i][None, None], min_order=1, max_order=1, approx=True, solver="min", steps=st)
            fin = all(np.isfinite(r[k]).all() for k in ("grad_rx", "tx_bar", "walls_bar", "phi_bar"))
            line.append(f"{st}:{'ok' if fin else 'NAN'} v={float(r['value'][0,0]):.4g} g={r['grad_rx'][0,0,0]:.3g},{r['grad_rx'][0,0,1]:.3g}")
        print("  obj", obj, " | ".join(line))
    ctx.set_candidate_mask(None)
