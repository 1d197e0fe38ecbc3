#!/usr/bin/env python3
"""The instrumented build's map against the product sweep's, every validity mode, cfg2 and a small grid (same bits expected)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from bench import workload  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

bad = 0
for g in (1024, 96):
    tx, walls, X, Y = workload(50, g)
    with Context(0) as ctx:
        ctx.set_scene(walls)
        ctx.set_grid(X, Y)
        for name, kw in (("hard", {}), ("hsig", dict(approx=True)), ("sig", dict(approx=True, function="sigmoid"))):
            for fun in ("received_power", "one"):
                p = make_params(min_order=0, max_order=2, fun=fun, **kw)
                for _ in range(3):
                    ctx.launch(p, tx)
                Z = ctx.get_map().copy()
                ctx.launch_stats(p, tx)
                Zs = ctx.get_map().copy()
                d = np.argwhere(Z != Zs)
                bad += len(d) > 0
                print(f"{g}^2 {name} {fun}: instrumented vs product: {len(d)} cells differ", [(int(r), int(c), float(Z[r, c]), float(Zs[r, c])) for r, c in d[:3]])
print("stats_cmp:", "OK" if not bad else f"{bad} FAILURES")
