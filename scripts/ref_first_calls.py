#!/usr/bin/env python3
"""What does the FIRST call of the reference's benchmark workload cost, call by call?  (reference tests/benchmarks/test_scene.py:
basic_scene, scene.grid(n), accumulate_on_transmitters_grid_over_paths, orders 0..1.)  A user of the reference's API makes one
call; bench.py's `reference_harness` leg showed a 95 ms first call for (n = 25, approx = True) in the middle of the sequence.

    python scripts/ref_first_calls.py [fresh]      # fresh: a new Scene (= a new context) per (n, approx)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from differt2d_amd.random import PRNGKey  # noqa: E402
from differt2d_amd.scene import Scene  # noqa: E402
from differt2d_amd.utils import received_power  # noqa: E402

fresh = len(sys.argv) > 1 and sys.argv[1] == "fresh"
key = PRNGKey(1234)
scene = Scene.basic_scene()
for n in (5, 25, 50):
    for approx in (False, True):
        if fresh:
            scene = Scene.basic_scene()
        X, Y = scene.grid(n)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=approx, key=key)
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"n={n} approx={approx}: calls 1..4 ms: " + " ".join(f"{t:.3f}" for t in ts), flush=True)
