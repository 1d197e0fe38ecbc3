#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the oracle (oracle/ref.py): forward maps with the NumPy fp32
restatement, gradients with the same op chain under torch autograd in fp64 (JAX-compatible
where/min/max/logistic semantics).  The reference itself (JAX) cannot be imported here, so these
are frozen ORACLE outputs: CPU tests check the oracle against them (drift), GPU tests check HIP.

Run from the repo root:  python scripts/make_golden.py
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import random_scene, unit_grid  # noqa: E402
from oracle import ref as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
F = np.float32

CASES = {
    # name: (walls, tx, X, Y, kwargs)
    "square_o1": lambda: (R.square_scene_walls(), np.array([0.2, 0.2], F), *unit_grid(9, 7), dict(min_order=0, max_order=1)),
    "obstacle_o2": lambda: (R.square_scene_with_obstacle_walls(), np.array([0.2, 0.2], F), *unit_grid(10, 8),
                            dict(min_order=0, max_order=2)),
    "random7_o2": lambda: (*random_scene(7, seed=3)[::-1], *unit_grid(11, 9), dict(min_order=0, max_order=2)),
    "random5_o3_patch": lambda: (*random_scene(5, seed=9)[::-1], *unit_grid(7, 6),
                                 dict(min_order=1, max_order=3, patch=0.02, alpha=50.0, tol=0.05, r_coef=0.3, height=0.25)),
}
MODES = {"hard": dict(approx=False), "hsig": dict(approx=True, function="hard_sigmoid"),
         "sig": dict(approx=True, function="sigmoid")}

for cname, make in CASES.items():
    walls, tx, X, Y, kw = make()
    for mname, mode in MODES.items():
        k = dict(kw)
        fun_kw = {q: k.pop(q) for q in ("r_coef", "height") if q in k}
        value = R.power_map(walls, tx, X, Y, fun_kwargs=fun_kw, **k, **mode)
        g = R.power_map_value_and_grads(walls, tx, X, Y, dtype="float64", fun_kwargs=fun_kw, **k, **mode)
        rng = np.random.default_rng(5)
        cot = rng.random(X.shape, dtype=F) + F(0.5)
        gc = R.power_map_value_and_grads(walls, tx, X, Y, cotangent=cot, dtype="float64", fun_kwargs=fun_kw, **k, **mode)
        path = os.path.join(OUT, f"{cname}_{mname}.npz")
        np.savez_compressed(
            path, walls=walls, tx=tx, X=X, Y=Y, value=value.astype(F), value64=g["value"], grad_rx=g["grad_rx"],
            tx_bar=g["tx_bar"], walls_bar=g["walls_bar"], cot=cot, tx_bar_cot=gc["tx_bar"], walls_bar_cot=gc["walls_bar"],
            kwargs=np.array(repr({**kw, **mode})),
        )
        print(path, value.shape, float(np.abs(g["grad_rx"]).max()))
