#!/usr/bin/env python3
"""Per-patch time distribution of the forward sweep on the bench workload (GPU box, instrumented build)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tx, walls, X, Y = workload(grid=g)
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    for approx in (False, True):
        cyc = ctx.wave_cycles(make_params(max_order=2, approx=approx), tx).astype(np.float64)
        tot = cyc.sum(); srt = np.sort(cyc.ravel())[::-1]
        pc = np.percentile(cyc, [50, 90, 99, 99.9])
        print(f"approx={approx} grid {g}: patches {cyc.size}, mean {cyc.mean():.0f}, p50/p90/p99/p99.9 {pc.round()}, max {cyc.max():.0f} ticks;"
              f" top 1% hold {srt[:cyc.size//100].sum()/tot:.1%}; sum/1024 SIMDs = {tot/1024:.0f}")
        for k in range(5):
            iy, ix = np.unravel_index(np.argsort(cyc.ravel())[-1 - k], cyc.shape)
            x, y = X[iy*8, ix*8], Y[iy*8, ix*8]
            print(f"   #{k} patch ({iy},{ix}) xy ({x:.3f},{y:.3f}) dist to tx {np.hypot(x-tx[0], y-tx[1]):.3f}  ticks {cyc[iy, ix]:.0f}")
        np.save(os.path.join("gpurun_out", f"patch_cycles_{g}_{int(approx)}.npy"), cyc)
