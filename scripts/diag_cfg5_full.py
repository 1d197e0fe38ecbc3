#!/usr/bin/env python3
"""Diagnostic (GPU box): the cells of configs[4]'s map where the GPU's MinPath sweep and oracle/d2d_oracle_opt.c disagree beyond the
test's bar although the oracle calls them well conditioned (tests/test_gpu_opt.py::test_cfg5_full_map_against_the_c_oracle):
per cell the GPU's and the oracle's values and, per candidate, the solver's final points on both sides."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from differt2d_amd.engine import Context, make_params  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import ref as R  # noqa: E402

F = np.float32
z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz"))
xys, kind, phi, tx, steps = z["xys"], z["kind"], z["phi"], z["tx"], int(z["steps"])
theta0 = [np.array([t, 0, 0, 0], F) if np.isfinite(t) else np.zeros(4, F) for t in z["theta0"]]
th = [np.array([t], F) if np.isfinite(t) else np.zeros(0, F) for t in z["theta0"]]
cands = R.all_path_candidates(7, order=1)
x = np.linspace(0.0, 1.0, 300).astype(F)
X, Y = np.meshgrid(x, x)
kw = dict(min_order=1, max_order=1, approx=True, solver="min", steps=steps)
with Context(0) as ctx:
    ctx.set_scene(xys, kind, phi)
    ctx.set_theta0(theta0)
    got = ctx.power_map(tx, X, Y, **kw)
    cond = CO.opt_conditioning(kind, xys, phi, tx, X, Y, cands, th, steps, solver="min", approx=True)
    stable, v64, scale = cond["stable"], cond["value64"], cond["scale"]
    bar = np.maximum(1e-5 * scale + 1e-5 * np.abs(v64), 2.0 * cond["dist"])
    err = np.abs(got - v64)
    bad = np.argwhere(stable & (err > bar))
    print(f"{int(stable.sum())} stable cells, {len(bad)} beyond the bar; scale {scale:.4f}")
    for r, c in bad[:40]:
        Xc, Yc = X[r:r + 1, c:c + 1], Y[r:r + 1, c:c + 1]
        v32, p32, l32 = CO.opt_power_map(kind, xys, phi, tx, Xc, Yc, cands, th, with_paths=True, solver="min", steps=steps, approx=True)
        w64, p64, l64 = CO.opt_power_map(kind, xys, phi, tx, Xc, Yc, cands, th, with_paths=True, solver="min", steps=steps, approx=True, dtype="float64")
        tr = ctx.trace_paths(make_params(max_order=4, approx=True, solver="min", steps=steps), np.asarray(tx, F)[None], np.stack([Xc[0], Yc[0]], -1), cands, theta0=theta0)
        print(f"cell ({r}, {c}) = ({X[r, c]:.6f}, {Y[r, c]:.6f}): GPU {got[r, c]!r} oracle32 {float(v32[0, 0])!r} oracle64 {float(w64[0, 0])!r} err/bar {err[r, c] / bar[r, c]:.2f} dist {cond['dist'][r, c]:.3e}")
        for ci, cand in enumerate(cands):
            k = len(cand)
            gp = tr["xys"][0, ci, 1:k + 1].reshape(-1)
            print(f"     cand {cand.tolist()}: GPU pts {gp} valid {tr['valid'][0, ci]:.6g} loss {tr['loss'][0, ci]:.3e} | oracle32 {p32[0, 0, ci, 0, :k].reshape(-1)} loss {l32[0, 0, ci]:.3e} | oracle64 {p64[0, 0, ci, 0, :k].reshape(-1)} loss {l64[0, 0, ci]:.3e}")
