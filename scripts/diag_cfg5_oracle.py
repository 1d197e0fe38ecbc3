#!/usr/bin/env python3
"""Diagnostic companion of scripts/diag_cfg5.py: the oracle's value / per-cell gradient / scene VJP of every sampled cell of
cfg5 on its own (one forward pass over all cells, one backward pass per cell), fp64 and fp32
-> gpurun_out/diag_cfg5_oracle.npz (CPU)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref as R  # noqa: E402

F = np.float32
z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz"))
xys, kind, phi, tx, ij, steps = z["xys"], z["kind"], z["phi"], z["tx"], z["ij"], int(z["steps"])
x = np.linspace(0.0, 1.0, 300).astype(F)
cands = R.all_path_candidates(7, order=1)
theta0 = [np.array([t], F) if np.isfinite(t) else np.zeros(0, F) for t in z["theta0"]]
X, Y = x[ij[:, 1]][None], x[ij[:, 0]][None]
out = {}
for dt in ("float64", "float32"):
    tb = R.TorchBackend(dt, diff_solver=True)
    w = tb.asarray(np.asarray(xys)).clone().requires_grad_(True)
    ph = tb.asarray(np.asarray(phi)).clone().requires_grad_(True)
    t = tb.asarray(np.asarray(tx)).clone().requires_grad_(True)
    gx = tb.asarray(X).clone().requires_grad_(True)
    gy = tb.asarray(Y).clone().requires_grad_(True)
    objs = [R.Obj(int(k), w[j, 0] if int(k) == R.VERTEX else w[j], ph[j]) for j, k in enumerate(kind)]
    Z = R.facc(t, objs, cands, R.vec(gx, gy, tb), solver="min", xp=tb, theta0s=theta0, steps=steps, approx=True)
    acc = {k: [] for k in ("fixed_bar", "xys_bar", "phi_bar")}
    for c in range(len(ij)):
        gw, gp, gt = torch.autograd.grad(Z[0, c], [w, ph, t], retain_graph=True, allow_unused=True)
        acc["xys_bar"].append(gw.numpy().copy())
        acc["phi_bar"].append(gp.numpy().copy())
        acc["fixed_bar"].append(gt.numpy().copy())
        print(dt, c, flush=True)
    ggx, ggy = torch.autograd.grad(Z.sum(), [gx, gy], allow_unused=True)
    tag = dt[-2:]
    out["value" + tag] = Z.detach().numpy()[0]
    out["grad_cell" + tag] = np.stack([ggx.numpy()[0], ggy.numpy()[0]], -1)
    for k in acc:
        out[k + tag] = np.stack(acc[k])
np.savez(os.path.join(ROOT, "gpurun_out", "diag_cfg5_oracle.npz"), **out)
