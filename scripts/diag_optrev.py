#!/usr/bin/env python3
"""Diagnostic: reverse mode through the solvers (opt_grad_mode 0) against the forward-tangent kernel (1) and timing at cfg5."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from differt2d_amd.engine import default_context, make_params
from differt2d_amd import _lib as L
import test_gpu_opt as TO
F = np.float32
ctx = default_context()
for solver, steps in (("min", 30), ("min", 200), ("fermat", 60)):
    for approx in (False, True):
        for role in ("rx", "tx"):
            scene, xys, kind, phi, X, Y, cands, theta0 = TO._opt_case(steps, solver, approx, role=role)
            tx = scene.transmitters["tx"].xy
            cot = (np.random.default_rng(5).random(X.shape) + 0.5).astype(F)
            kw = dict(min_order=0, max_order=1, approx=approx, solver=solver, steps=steps, grid_role=L.GRID_TX if role == "tx" else L.GRID_RX)
            out = []
            for mode in (0, 1):
                ctx.set_option("opt_grad_mode", mode)
                out.append(TO._gpu_opt_grads(xys, kind, phi, tx, X, Y, cands, theta0, cot, **kw))
            a, b = out
            line = [f"{solver} {steps} approx={approx} {role}: value equal {np.array_equal(a['value'], b['value'], equal_nan=True)}"]
            for k in ("grad_rx", "tx_bar", "walls_bar", "phi_bar"):
                x, y = np.asarray(a[k], np.float64), np.asarray(b[k], np.float64)
                fin = np.isfinite(x) & np.isfinite(y)
                sc = max(np.abs(y[fin]).max(), 1e-12) if fin.any() else 1.0
                line.append(f"{k}: max|d|/scale {np.abs(x - y)[fin].max() / sc if fin.any() else 0:.2e} nan rev/fwd {int(np.isnan(x).sum())}/{int(np.isnan(y).sum())}")
            print(" | ".join(line), flush=True)
ctx.set_option("opt_grad_mode", 0)
# cfg5 timing + NaN pattern
z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz"))
xys, kind, phi, tx, ij, steps = z["xys"], z["kind"], z["phi"], z["tx"], z["ij"], int(z["steps"])
theta0 = [np.array([t, 0, 0, 0], F) if np.isfinite(t) else np.zeros(4, F) for t in z["theta0"]]
x = np.linspace(0.0, 1.0, 300).astype(F)
X, Y = np.meshgrid(x, x)
ctx.set_scene(xys, kind, phi); ctx.set_theta0(theta0)
kw = dict(min_order=1, max_order=1, approx=True, solver="min", steps=steps)
for mode in (0, 1):
    ctx.set_option("opt_grad_mode", mode)
    ctx.value_and_grads(tx, X, Y, **kw)
    t0 = time.perf_counter(); full = ctx.value_and_grads(tx, X, Y, **kw); dt = time.perf_counter() - t0
    ctx.set_option("time_kernel", 1); ctx.set_grid(X, Y); ctx.launch_vg(make_params(**kw), tx, scene_vjp=True); kms = ctx.last_kernel_ms(); ctx.set_option("time_kernel", 0)
    bad = ~np.isfinite(full["grad_rx"]).all(-1)
    print(f"cfg5 mode {mode}: {dt*1e3:.1f} ms per call (kernel {kms:.2f} ms); non-finite gradient cells {int(bad.sum())}; VJP finite {[bool(np.isfinite(full[k]).all()) for k in ('tx_bar','walls_bar','phi_bar')]}")
    g = full["grad_rx"][ij[:, 0], ij[:, 1]]
    g64, g32 = z["grad_cell64"][0], z["grad_cell32"][0]
    gs = np.maximum(np.abs(g64).max(-1), np.median(np.abs(g64).max(-1)))[:, None]
    err, ref = np.abs(g - g64) / gs, np.abs(g32 - g64) / gs
    st = z["stable"]
    print("   sampled cells: NaN", int(np.isnan(g).any(-1).sum()), "stable err max", np.nanmax(err[st]), "median", np.nanmedian(err[st]), "| oracle fp32 vs fp64 max", ref[st].max(), "median", np.median(ref[st]))
    sub = ctx.value_and_grads(tx, x[ij[st, 1]][None], x[ij[st, 0]][None], **kw)
    for k_got, k_want in (("tx_bar", "fixed_bar"), ("walls_bar", "xys_bar"), ("phi_bar", "phi_bar")):
        a_, b64, b32 = np.asarray(sub[k_got], np.float64), z[k_want + "64"], z[k_want + "32"]
        s_ = max(float(np.abs(b64).max()), 1e-12)
        print(f"   {k_got}: max err/scale {np.nanmax(np.abs(a_ - b64)) / s_:.2e} (oracle fp32 vs fp64 {np.nanmax(np.abs(b32 - b64)) / s_:.2e}) nan {int(np.isnan(a_).sum())}")
