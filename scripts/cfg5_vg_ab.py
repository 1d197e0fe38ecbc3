#!/usr/bin/env python3
"""cfg5 legs of bench.py on their own, twice (A/B builds through D2D_LIB: scripts/ab_build.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from differt2d_amd.engine import Context  # noqa: E402

import time  # noqa: E402

with Context(0) as ctx:
    def timed(fn, n_steps, n_warmup):
        for _ in range(n_warmup):
            fn()
        ctx.synchronize()
        t0 = time.perf_counter()
        ctx.timer_begin()
        for _ in range(n_steps):
            fn()
        stream_ms = ctx.timer_end()
        return time.perf_counter() - t0, stream_ms / n_steps

    for rep in range(2):
        out = bench.cfg5_leg(ctx, timed)
        print({k: round(v["ms_per_step"], 3) for k, v in out.items() if isinstance(v, dict)}, flush=True)
