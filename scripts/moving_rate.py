#!/usr/bin/env python3
"""Moving transmitter, back-to-back launches (no per-step synchronisation): wall time per step with the pipelined
preparation (work history three launches old) and without (one launch old)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload, moving_transmitters
from differt2d_amd.engine import Context, make_params
tx0, walls, X, Y = workload(grid=1024)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
settings = sys.argv[2:] or ["-"]  # e.g. sched_key_mode=1,heavy_split=0
txs = moving_transmitters(tx0, n + 5)
for approx, pipe, setting in [(a, q, s) for a in (False, True) for q in (1, 0) for s in settings]:
    if True:
        with Context(0) as ctx:
            ctx.set_scene(walls); ctx.set_grid(X, Y); ctx.set_option("pipeline", pipe)
            if setting != "-":
                for kv in setting.split(","):
                    k, v = kv.split("="); ctx.set_option(k, int(v))
            p = make_params(max_order=2, approx=approx)
            for t in txs[:5]:
                ctx.launch(p, t)
            ctx.synchronize()
            t0 = time.perf_counter()
            for t in txs[5:]:
                ctx.launch(p, t)
            ctx.synchronize()
            print(f"approx={approx} pipeline={pipe} {setting}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms per step", flush=True)
