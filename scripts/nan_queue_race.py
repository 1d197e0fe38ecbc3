#!/usr/bin/env python3
"""Does a region's probe queue ever hand out a slot nobody wrote?  (the SIGABRT of the driver's GPU suite.)

Runs the case that aborted -- 50 walls, 128 x 128 cells, orders 0..2, value+grad with the scene VJP -- and coarser / more crowded
ones N times with the NaN scan's counters on, and prints, per case: launches, probes a wave made itself because the queue was full
(`self_probes` > 0 = the queue DID overflow), queue items refused because they named no patch / object (`bad_items`: with the
sentinel-filled probe build -DD2D_NAN_QUEUE_R5 a non-zero count is round 5's reserve-and-roll-back race caught in the act; the
product build must print 0), and whether the flags equal the exhaustive kernel's.

    python scripts/nan_queue_race.py [repeats]            # D2D_LIB selects the build
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import random_scene, unit_grid  # noqa: E402

from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    cases = [("50 walls 128^2", 50, 128, 1234), ("50 walls 32^2", 50, 32, 1234), ("120 walls 64^2", 120, 64, 7), ("200 walls 16^2", 200, 16, 9)]
    total_bad = 0
    with Context(0) as c:
        c.set_option("nan_scan_stats", 1)
        for name, nw, g, seed in cases:
            tx, walls = random_scene(nw, seed=seed)
            X, Y = unit_grid(g)
            c.set_scene(walls)
            for role, rname in ((L.GRID_RX, "rx"), (L.GRID_TX, "tx")):
                for approx in (False, True):
                    kw = dict(min_order=0, max_order=2, approx=approx, grid_role=role)
                    want = c.value_and_grads(tx, X, Y, strict_nan=True, **kw)
                    nan_w = np.isnan(want["grad_rx"])
                    selfp = bad = mism = 0
                    for _ in range(reps):
                        got = c.value_and_grads(tx, X, Y, strict_nan=False, **kw)
                        st = c.debug_nan_scan()
                        selfp += st["self_probes"]
                        bad += st["bad_items"]
                        mism += int(not np.array_equal(np.isnan(got["grad_rx"]), nan_w)) + int(
                            not np.array_equal(np.isnan(got["walls_bar"]), np.isnan(want["walls_bar"])))
                    total_bad += bad + mism
                    print(f"{name} {rname} approx={int(approx)}: launches {reps}, self_probes {selfp}, bad_items {bad}, "
                          f"launches whose NaN flags differ from the exhaustive kernel's {mism}, NaN cells {int(nan_w.any(-1).sum())}", flush=True)
    print("TOTAL bad_items + flag mismatches:", total_bad)
    return 0


if __name__ == "__main__":
    sys.exit(main())
