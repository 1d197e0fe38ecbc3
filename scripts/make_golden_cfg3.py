#!/usr/bin/env python3
"""Generates tests/golden/cfg3_grad_{rx,tx}_{hard,hsig}.npz: BASELINE.json configs[2] (the 50-wall scene of configs[1],
value + gradient) on 8 x 8 blocks of the full 1024 x 1024 grid, with reverse-mode autodiff of the oracle's op chain
(oracle/ref.py under torch.autograd, float64, JAX-compatible where / min / max / logistic semantics).

Blocks (aligned to the kernel's 8 x 8 patches, so that the GPU sweeps them exactly as it does inside the full map): the
patch holding the transmitter and two of its neighbours, patches crossed by walls (wall mid points and end points),
and random patches.  Per block set: value, per-cell gradient w.r.t. the cell, and the VJP w.r.t. the fixed end point
and every wall end point with cotangent = 1 on the selected cells (0 elsewhere).

"rx": the grid cells are receivers (scene.py:1803-1953, grad w.r.t. rx, scene.py:1920-1923);
"tx": the grid cells are transmitters (scene.py:1489-1648, grad w.r.t. tx, scene.py:1617-1620), fewer blocks.

Run from the repo root (a few minutes per file on 8 cores; oracle/ref.py's candidate-batched evaluation):  python scripts/make_golden_cfg3.py
"""

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import random_scene  # noqa: E402
from oracle import ref as R  # noqa: E402

F = np.float32
GRID = 1024


def pick_blocks(tx, walls, n_random, rng):
    """Top-left (row, col) of 8 x 8 blocks, aligned to multiples of 8."""
    def patch_of(p):
        j, i = int(round(float(p[0]) * (GRID - 1))), int(round(float(p[1]) * (GRID - 1)))
        return (min(i, GRID - 1) // 8 * 8, min(j, GRID - 1) // 8 * 8)

    blocks = []
    pt = patch_of(tx)
    blocks += [pt, (pt[0], min(pt[1] + 8, GRID - 8)), (max(pt[0] - 8, 0), pt[1])]  # the transmitter's patch + 2 neighbours
    for w in (3, 17, 31, 44):  # patches crossed by walls: mid points and one end point
        blocks.append(patch_of(0.5 * (walls[w, 0] + walls[w, 1])))
    blocks.append(patch_of(walls[8, 0]))
    while len(blocks) < 8 + n_random:
        b = (int(rng.integers(0, GRID // 8)) * 8, int(rng.integers(0, GRID // 8)) * 8)
        if b not in blocks:
            blocks.append(b)
    out = []
    for b in blocks:
        if b not in out:
            out.append(b)
    return np.array(out, np.int32)


def main():
    tx, walls = random_scene(50, seed=1234)
    x = np.linspace(0.0, 1.0, GRID).astype(F)
    rng = np.random.default_rng(7)
    for role, n_random in (("rx", 4), ("tx", 0)):
        blocks = pick_blocks(tx, walls, n_random, rng)
        if role == "tx":
            blocks = blocks[[0, 3, 5, 7]]
        ii = (blocks[:, 0, None, None] + np.arange(8)[None, :, None]) + np.zeros((1, 1, 8), np.int64)
        jj = (blocks[:, 1, None, None] + np.arange(8)[None, None, :]) + np.zeros((1, 8, 1), np.int64)
        X, Y = x[jj], x[ii]  # (B, 8, 8)
        for name, mode in (("hard", dict(approx=False)), ("hsig", dict(approx=True, function="hard_sigmoid"))):
            t = time.time()
            kw = dict(min_order=0, max_order=2, grid_role=role, chunk=128, **mode)
            g = R.power_map_value_and_grads_batched(walls, tx, X, Y, dtype="float64", **kw)
            # the same chain in fp32: the value map (bit-comparable with the GPU) and the positions of the reference's
            # autodiff NaN artefacts (un == 0 exactly / zero-length segments happen where the FP32 chain hits them)
            g32 = R.power_map_value_and_grads_batched(walls, tx, X, Y, dtype="float32", **kw)
            v32 = R.power_map_batched(walls, tx, X, Y, min_order=0, max_order=2, grid_role=role, **mode)
            path = os.path.join(ROOT, "tests", "golden", f"cfg3_grad_{role}_{name}.npz")
            np.savez_compressed(path, blocks=blocks, value=v32.astype(F), value64=g["value"], grad=g["grad_rx"],
                                fixed_bar=g["tx_bar"], walls_bar=g["walls_bar"], grad32=g32["grad_rx"].astype(F),
                                fixed_bar32=g32["tx_bar"].astype(F), walls_bar32=g32["walls_bar"].astype(F))
            nan_cells = int(np.isnan(g32["grad_rx"]).any(-1).sum())
            print(path, f"{time.time() - t:.0f}s", "cells:", X.size, "lit:", int((v32 != 0).sum()), "NaN-gradient cells:", nan_cells,
                  "max|grad|:", float(np.nanmax(np.abs(g["grad_rx"]))), flush=True)


if __name__ == "__main__":
    main()
