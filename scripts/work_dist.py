#!/usr/bin/env python3
"""Distribution of the per-patch work counter (units of ~25 wave-instructions) of the forward sweep, bench workload."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
approx = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
tx, walls, X, Y = workload(grid=g)
with Context(0) as ctx:
    ctx.set_scene(walls); ctx.set_grid(X, Y)
    ctx.set_option("heavy_split", 0)
    p = make_params(max_order=2, approx=approx)
    for _ in range(2): ctx.launch(p, tx)
    ctx.synchronize()
    n = (g // 8) ** 2
    w = ctx.debug_get_work(n).astype(np.float64)
    np.save(f"gpurun_out/work_{g}_{int(approx)}.npy", w)
    srt = np.sort(w)[::-1]
    print(f"grid {g} approx {approx}: patches {n}, mean {w.mean():.1f}, p50 {np.percentile(w,50):.0f} p90 {np.percentile(w,90):.0f} p99 {np.percentile(w,99):.0f} max {w.max():.0f}")
    for f in (1.5, 2, 3, 4, 6, 8):
        print(f"   patches above {f} x mean: {(w > f * w.mean()).sum()}  (hold {w[w > f*w.mean()].sum()/w.sum():.1%} of the work)")
    print("   sum / 1024 SIMDs:", w.sum() / 1024, " top-10:", srt[:10])
