#!/usr/bin/env python3
"""BASELINE.json configs[4]-like: square scene + RIS + its two vertices, 300^2 grid, order 1, MinPath / FermatPath with
1000 Adam steps: end-to-end time of Scene.accumulate_on_receivers_grid_over_paths."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import differt2d_amd.geometry as G
from differt2d_amd.engine import default_context
from differt2d_amd.geometry import RIS
from differt2d_amd.scene import Scene
from differt2d_amd.utils import received_power
scene = Scene.square_scene()
ris = RIS(xys=[[0.5, 0.3], [0.5, 0.7]], phi=np.pi / 4)
scene = scene.add_objects(ris, *ris.get_vertices())
X, Y = scene.grid(n=300)
for name in ("MinPath", "FermatPath"):
    for par in (1, 0):
        default_context().set_option("opt_parallel", par)
        kw = dict(fun=received_power, path_cls=getattr(G, name), order=1, reduce_all=True, approx=True,
                  path_cls_kwargs={"steps": 1000}, key=1234)
        scene.accumulate_on_receivers_grid_over_paths(X, Y, **kw)
        t = time.perf_counter()
        for _ in range(3):
            Z = scene.accumulate_on_receivers_grid_over_paths(X, Y, **kw)
        print(f"{name} candidates {'side by side' if par else 'one after the other'}: {(time.perf_counter() - t) / 3 * 1e3:.2f} ms  (sum {float(Z.sum()):.4f})", flush=True)
default_context().set_option("opt_parallel", 1)
