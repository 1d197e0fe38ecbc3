#!/usr/bin/env python3
"""Where does a small call of the reference's entry point spend its 80 - 90 us?  (reference tests/benchmarks/test_scene.py's workload:
basic_scene, scene.grid(50), accumulate_on_transmitters_grid_over_paths, orders 0..1.)  cProfile over N calls, plus the bare
launch -> synchronise -> download sequence on the engine for the same problem.
usage: python scripts/small_call_profile.py [n_calls]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import make_params  # noqa: E402
from differt2d_amd.random import PRNGKey  # noqa: E402
from differt2d_amd.scene import Scene  # noqa: E402
from differt2d_amd.utils import received_power  # noqa: E402

n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scene = Scene.basic_scene()
key = PRNGKey(1234)
X, Y = scene.grid(50)
call = lambda: scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=False, key=key)  # noqa: E731
for _ in range(20):
    call()
t0 = time.perf_counter()
for _ in range(n_calls):
    call()
print(f"API call: {(time.perf_counter() - t0) / n_calls * 1e6:.1f} us")
ctx = scene._ctx()
rx = np.asarray(next(iter(scene.receivers.values())).xy, np.float32)
p = make_params(min_order=0, max_order=1, approx=False, grid_role=L.GRID_TX)
for name, fn in (("launch + synchronize", lambda: (ctx.launch(p, rx), ctx.synchronize())), ("launch + get_map", lambda: (ctx.launch(p, rx), ctx.get_map())),
                 ("get_map alone", lambda: ctx.get_map())):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(n_calls):
        fn()
    print(f"{name}: {(time.perf_counter() - t0) / n_calls * 1e6:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(n_calls):
    call()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
