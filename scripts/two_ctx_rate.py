#!/usr/bin/env python3
"""Feasibility: two contexts on one device, launches alternating between them -- what two complete pipelines (preparation + sweep
on streams of their own, consecutive sweeps free to overlap head to tail) would give per map, against one context's rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import workload
from differt2d_amd.engine import Context, make_params
tx, walls, X, Y = workload()
approx = len(sys.argv) > 1 and sys.argv[1] == "1"
p = make_params(min_order=0, max_order=2, approx=approx)
ctxs = [Context(0) for _ in range(3)]
for c in ctxs:
    c.set_scene(walls); c.set_grid(X, Y)
    for _ in range(5):
        c.launch(p, tx)
    c.synchronize()
for n_ctx in (1, 2, 3, 1, 2):
    use = ctxs[:n_ctx]
    K = 300
    for c in use:
        c.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        use[i % n_ctx].launch(p, tx)
    for c in use:
        c.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{n_ctx} context(s): {dt*1e3:.4f} ms per map")
ref = ctxs[0].get_map()
print("maps equal:", all(np.array_equal(c.get_map(), ref) for c in ctxs))
