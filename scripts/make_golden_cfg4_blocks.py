#!/usr/bin/env python3
"""Generates tests/golden/cfg4_blocks.npz: BASELINE.json configs[3] at FULL size (200 random walls, NumPy seed 1234, 2048 x 2048
receivers, orders 0..3 = 7 960 201 candidates per cell) on six CONTIGUOUS blocks of 64 x 64 cells, every cell evaluated by the
C oracle (oracle/d2d_oracle.c, prune level 2: per cell and candidate, exact -- tests/test_oracle_c.py), hard and hard_sigmoid
validity: the block that holds the transmitter (where the region candidate lists are longest), two of its neighbours (one
aligned to the regions of the lists, one straddling four of them), a block centred on a wall, the corner block at the origin,
and one drawn at random.  24 576 cells per mode, 0.75 - 2 core-seconds each.

Run from the repo root:  python scripts/make_golden_cfg4_blocks.py [threads]
"""

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import random_scene  # noqa: E402
from oracle import c_oracle  # noqa: E402

F = np.float32
B = 64
G = 2048


def blocks_of(tx, walls):
    itx, jtx = int(round(float(tx[1]) * (G - 1))), int(round(float(tx[0]) * (G - 1)))
    mid = 0.5 * (walls[0, 0] + walls[0, 1])
    iw, jw = int(round(float(mid[1]) * (G - 1))), int(round(float(mid[0]) * (G - 1)))
    rng = np.random.default_rng(44)
    clamp = lambda v: int(min(max(v, 0), G - B))
    bi, bj = itx // B * B, jtx // B * B
    return np.array([
        [bi, bj],                                  # the transmitter's block (aligned to the 8 x 8-patch top regions of 16 patches)
        [bi, clamp(bj - B)],                       # its neighbour in x: the shadow boundaries of the walls around the transmitter
        [clamp(bi - B - 24), clamp(bj + B // 2 + 8)],  # diagonal neighbour, unaligned on purpose (straddles 4 top regions)
        [clamp(iw - B // 2), clamp(jw - B // 2)],  # centred on the middle of wall 0 (all in shadow: every cell exactly 0)
        [0, 0],                                    # the corner at the origin
        [clamp(int(rng.integers(0, G - B))), clamp(int(rng.integers(0, G - B)))],
    ], np.int32)


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    tx, walls = random_scene(200, seed=1234)
    x = np.linspace(0.0, 1.0, G).astype(F)
    blocks = blocks_of(tx, walls)
    out = dict(blocks=blocks, block_size=np.int32(B), grid=np.int32(G))
    for name, mode in (("hard", dict(approx=False)), ("hsig", dict(approx=True, function="hard_sigmoid"))):
        maps = []
        for (i0, j0) in blocks:
            t = time.time()
            X, Y = np.meshgrid(x[j0 : j0 + B], x[i0 : i0 + B])
            m = c_oracle.power_map(walls, tx, X, Y, min_order=0, max_order=3, prune=2, nthreads=threads, **mode)
            maps.append(m)
            print(name, (int(i0), int(j0)), f"{time.time() - t:.0f} s", "non-zero cells", int((m != 0).sum()), "max", float(np.nanmax(m)), flush=True)
        out[name] = np.stack(maps)
    path = os.path.join(ROOT, "tests", "golden", "cfg4_blocks.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
