#!/usr/bin/env python3
"""A/B of the "unpiped_max_tiles" option (small launches prepare on the sweep's own stream): the reference-harness workload
(basic_scene, orders 0..1, TX grid) and cfg2's scene (orders 0..2, RX grid) on small grids -- API call, launch -> synchronise,
and back-to-back launches (200 without a wait), microseconds; maps must be identical.
usage: python scripts/small_launch_ab.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_scene, unit_grid  # noqa: E402
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402
from differt2d_amd.scene import Scene  # noqa: E402


def timeit(fn, n=600, warm=30):
    for _ in range(warm):
        fn()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best


scene = Scene.basic_scene()
walls7 = np.stack([np.asarray(o.xys, np.float32) for o in scene.objects])
rx = np.asarray(next(iter(scene.receivers.values())).xy, np.float32)
tx50, walls50 = random_scene(50, seed=1234)
cases = [("basic_scene TX grid orders 0..1", walls7, rx, dict(min_order=0, max_order=1, grid_role=L.GRID_TX), (5, 25, 50, 128)),
         ("cfg2 scene RX grid orders 0..2", walls50, tx50, dict(min_order=0, max_order=2), (32, 64, 128, 256, 512))]
ref = {}
for limit in (0, 64, 256, 1024, 4096):
    with Context(0) as ctx:
        ctx.set_option("unpiped_max_tiles", limit)
        for name, walls, fixed, kw, sizes in cases:
            ctx.set_scene(walls)
            for g in sizes:
                X, Y = unit_grid(g)
                ctx.set_grid(X, Y)
                p = make_params(**kw)
                ctx.launch(p, fixed)
                m = ctx.get_map()
                key = (name, g)
                if key in ref:
                    assert np.array_equal(m, ref[key], equal_nan=True), (limit, key)
                ref[key] = m
                t_sync = timeit(lambda: (ctx.launch(p, fixed), ctx.synchronize()))
                t_get = timeit(lambda: (ctx.launch(p, fixed), ctx.get_map()))

                def b2b():
                    for _ in range(50):
                        ctx.launch(p, fixed)
                    ctx.synchronize()
                t_b2b = timeit(b2b, n=20, warm=3) / 50
                tiles = ((g + 7) // 8) ** 2
                print(f"limit {limit:5d} | {name} {g:4d}^2 ({tiles:5d} patches): launch+sync {t_sync:6.1f} us, launch+get_map {t_get:6.1f} us, back-to-back {t_b2b:6.1f} us per launch", flush=True)
