#!/usr/bin/env python3
"""Generates tests/golden/cfg2_fullmap_crc.npz: BASELINE.json configs[1] at FULL size (50 random walls, NumPy seed 1234,
1024 x 1024 grid over the unit square, orders 0..2 = 2 501 candidates per cell), every cell, with the C oracle
(oracle/d2d_oracle.c, result-preserving pruning on).  A full fp32 map is 4 MB, so the fixture holds one CRC-32 per grid
row of every map (and the SHA-256 of the whole map): a GPU map whose 1024 row CRCs all match equals the oracle's map bit
for bit, and a mismatch names the rows to recompute live.

Maps: the grid as receivers (scene.py:1803-1953) and as transmitters (scene.py:1489-1648), hard and hard_sigmoid
validity, received power and -- from the same oracle pass -- the valid-path count map (fun = 1).

Run from the repo root (about 3 min per map on 8 cores):  python scripts/make_golden_fullmap.py
"""

import hashlib
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import random_scene  # noqa: E402
from oracle import c_oracle  # noqa: E402

F = np.float32


def row_crcs(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=F)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(a.shape[0])], dtype=np.uint32)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype=F).tobytes()).hexdigest()


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    tx, walls = random_scene(50, seed=1234)
    x = np.linspace(0.0, 1.0, grid).astype(F)
    X, Y = np.meshgrid(x, x)
    out = {"grid": np.int32(grid)}
    for role in ("rx", "tx"):
        for name, mode in (("hard", dict(approx=False)), ("hsig", dict(approx=True, function="hard_sigmoid"))):
            t = time.time()
            power, count = c_oracle.power_and_count_maps(walls, tx, X, Y, min_order=0, max_order=2, prune=True,
                                                         grid_role=role, **mode)
            for what, a in (("power", power), ("count", count)):
                key = f"{role}_{name}_{what}"
                out[key + "_crc"] = row_crcs(a)
                out[key + "_sha256"] = np.array(sha(a))
                out[key + "_sum"] = np.float64(a.astype(np.float64).sum())
                out[key + "_nonzero"] = np.int64((a != 0).sum())
            print(role, name, f"{time.time() - t:.0f}s", "lit cells:", int((power != 0).sum()), "max count:", float(count.max()),
                  flush=True)
    path = os.path.join(ROOT, "tests", "golden", "cfg2_fullmap_crc.npz" if grid == 1024 else f"cfg2_fullmap_crc_{grid}.npz")
    np.savez_compressed(path, **out)
    print(path)


if __name__ == "__main__":
    main()
