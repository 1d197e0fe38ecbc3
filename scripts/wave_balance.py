#!/usr/bin/env python3
"""Per-wave time distribution of the forward sweep on the bench workload (GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
tx, walls, X, Y = workload()
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    for approx in (False, True):
        cyc = ctx.wave_cycles(make_params(max_order=2, approx=approx), tx).astype(np.float64)
        tot = cyc.sum(); srt = np.sort(cyc.ravel())[::-1]
        print(f"approx={approx}: waves {cyc.size}, mean {cyc.mean():.0f}, median {np.median(cyc):.0f}, p99 {np.percentile(cyc,99):.0f}, max {cyc.max():.0f} ticks;"
              f" top 1% of waves hold {srt[:cyc.size//100].sum()/tot:.1%} of the time; sum/1024 SIMDs/8 = {tot/1024/8:.0f} vs max {cyc.max():.0f}")
        iy, ix = np.unravel_index(np.argmax(cyc), cyc.shape)
        print("   slowest patch at cell", iy*8, ix*8, "xy", X[iy*8, ix*8], Y[iy*8, ix*8], " tx", tx)
