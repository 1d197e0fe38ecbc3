#!/usr/bin/env python3
"""One workload of the non-headline kernels, launched N times (for rocprofv3; scripts/profile_kernels.sh).

usage: kernel_lab.py <case> [launches]
  cfg3      value + gradient + scene VJP of cfg2's sweep            power_fwd_kernel<MODE, false, 2, GRADK = true, LISTED = true, 1>
  cfg3_hsig the same in hard_sigmoid validity; cfg3_txg / cfg3_txg_hsig: the cells are transmitters (env D2D_NAN_SCAN = 0 / 1 / 2: the NaN scan)
  txg       cfg2-sized TX grid (cells are transmitters)            power_fwd_txg_kernel
  cfg4      200 walls, 2048^2, orders 0..3, hard                   power_fwd_kernel<0, false, 3, false, true, 4>
  sigmoid   cfg2 in sigmoid validity                               power_fwd_kernel<2, ...>
  cfg5      RIS scene, 300^2, MinPath 1000 steps, value+grad+VJP   power_opt_rev_kernel<1>
  cfg5_fwd  ... forward values only                                power_opt_cand_kernel
  cfg5_tan  ... gradients by forward tangents (opt_grad_mode 1)    power_opt_grad_kernel
Prints the mean sweep-kernel time (HIP events) and the mean wall time per launch."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import workload  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

F = np.float32
case = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
with Context(0) as ctx:
    if case.startswith("cfg5"):
        z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz"))
        ctx.set_scene(z["xys"], z["kind"], z["phi"])
        ctx.set_theta0([np.array([t, 0, 0, 0], F) if np.isfinite(t) else np.zeros(4, F) for t in z["theta0"]])
        x = np.linspace(0.0, 1.0, 300).astype(F)
        X, Y = np.meshgrid(x, x)
        tx = z["tx"]
        p = make_params(min_order=1, max_order=1, approx=True, solver="min", steps=int(z["steps"]))
        ctx.set_grid(X, Y)
        if case == "cfg5_tan":
            ctx.set_option("opt_grad_mode", 1)
        launch = (lambda: ctx.launch(p, tx)) if case == "cfg5_fwd" else (lambda: ctx.launch_vg(p, tx, scene_vjp=True))
    else:
        walls_n, grid, order = (200, 2048, 3) if case == "cfg4" else (50, 1024, 2)
        tx, walls, X, Y = workload(walls_n, grid)
        ctx.set_scene(walls)
        ctx.set_grid(X, Y)
        kw = dict(min_order=0, max_order=order)
        if case in ("cfg3_hsig",):
            kw.update(approx=True)
        if case == "sigmoid":
            kw.update(approx=True, function="sigmoid")
        if case in ("txg", "cfg3_txg", "cfg3_txg_hsig"):
            kw.update(grid_role=L.GRID_TX)
        if case == "cfg3_txg_hsig":
            kw.update(approx=True)
        if os.environ.get("D2D_NAN_SCAN"):  # 0 off, 1 two levels (default), 2 one wave per patch
            ctx.set_option("nan_scan", int(os.environ["D2D_NAN_SCAN"]))
        p = make_params(**kw)
        launch = (lambda: ctx.launch_vg(p, tx, scene_vjp=True)) if case.startswith("cfg3") else (lambda: ctx.launch(p, tx))
    ctx.set_option("time_kernel", 1)
    for _ in range(4):
        launch()
    ctx.synchronize()
    km, t0 = [], time.perf_counter()
    for _ in range(n):
        launch()
        try:
            km.append(ctx.last_kernel_ms())
        except Exception:  # noqa: BLE001 -- (the forward optimiser sweep has no timed kernel)
            km.append(float("nan"))
    ctx.synchronize()
    print(f"{case}: sweep kernel {np.mean(km):.4f} ms (min {np.min(km):.4f}), wall {(time.perf_counter() - t0) / n * 1e3:.4f} ms per launch incl. the wait, {n} launches")
