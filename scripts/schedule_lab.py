#!/usr/bin/env python3
"""What is the best order to start the patches in?  Times the sweep kernel of the bench workload under several
schedules (diagnostic ABI: d2d_debug_set_schedule)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params

g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
approx = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
tx, walls, X, Y = workload(grid=g)
T = (g // 8) ** 2
rng = np.random.default_rng(0)
with Context(0) as ctx:
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    p = make_params(max_order=2, approx=approx)
    ctx.set_option("split_max_tiles", 0)
    ctx.set_option("time_kernel", 1)

    def t(label, order=None, reps=15):
        ctx.debug_set_schedule(order)
        ks = []
        for _ in range(reps):
            ctx.launch(p, tx)
            ks.append(ctx.last_kernel_ms())
        print(f"{label:48s} kernel {np.median(ks):.4f} ms (min {min(ks):.4f})", flush=True)

    t("built-in (device counting sort)")
    order, key = ctx.debug_get_schedule(T)
    cyc = ctx.wave_cycles(p, tx).astype(np.float64).ravel()   # measured cost per patch (instrumented build)
    ident = np.arange(T, dtype=np.int32)
    t("identity", ident)
    t("built-in order, replayed", order)
    stable = np.argsort(-key.astype(np.int64), kind="stable").astype(np.int32)
    t("key desc, ascending patch index inside a key", stable)
    rnd = rng.permutation(T)
    t("key desc, random inside a key", rnd[np.argsort(-key[rnd].astype(np.int64), kind="stable")].astype(np.int32))
    t("random", rnd.astype(np.int32))
    t("measured cost desc (oracle schedule)", np.argsort(-cyc, kind="stable").astype(np.int32))
    # interleave: dearest patches spread evenly over the launch instead of all first
    lpt = np.argsort(-cyc, kind="stable")
    half = T // 2
    t("measured cost: dear half first, each half ascending index", np.concatenate([np.sort(lpt[:half]), np.sort(lpt[half:])]).astype(np.int32))
    # row-major blocks of 8 patches by XCD: patch b -> XCD b % 8, keep spatial neighbours on one XCD
    t("key desc, inside a key 8-strided (neighbours share an XCD)", stable.reshape(-1, 8).T.reshape(-1).astype(np.int32) if T % 8 == 0 else stable)
    ctx.debug_set_schedule(None)
    np.savez(os.path.join("gpurun_out", f"schedule_lab_{g}_{int(approx)}.npz"), key=key, cyc=cyc, order=order, tx=tx)
