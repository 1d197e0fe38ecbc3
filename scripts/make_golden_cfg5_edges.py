#!/usr/bin/env python3
"""Generates tests/golden/cfg5_edges.npz: BASELINE.json configs[4] (see make_golden_cfg5.py) on the cells where the reference's
own reverse-mode chain yields NaN gradients -- receivers lying exactly on an object's supporting line (the outer rows and
columns of scene.grid(n=300) sit on the square's walls) -- and on their neighbours: the fp32 autodiff oracle's value, per-cell
gradient and which of its entries are finite.  The GPU's NaN positions must coincide with these.

Cells: rows 0, 1, 298, 299 and columns 0, 1, 298, 299 (every third cell), plus rows 90 and 210 (they cross the RIS's end
points' height) every third cell.   Run from the repo root (about 20 minutes):  python scripts/make_golden_cfg5_edges.py
"""

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

from make_golden_cfg5 import scene_tables, solver_agreement  # noqa: E402
from oracle import ref as R  # noqa: E402

F = np.float32


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    z = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz"))
    xys, kind, phi = scene_tables()
    tx = np.array([0.2, 0.2], F)
    x = np.linspace(0.0, 1.0, 300).astype(F)
    cands = R.all_path_candidates(7, order=1)
    theta0 = [np.array([t], F) if np.isfinite(t) else np.zeros(0, F) for t in z["theta0"]]  # the same starts as cfg5_samples
    cells = set()
    for r in (0, 1, 298, 299, 90, 210):
        cells |= {(r, c) for c in range(0, 300, 3)}
    for c in (0, 1, 298, 299):
        cells |= {(r, c) for r in range(0, 300, 3)}
    ij = np.array(sorted(cells), np.int32)
    out = dict(ij=ij, steps=np.int32(steps))
    B = 256
    t0 = time.time()
    for dt, tag in (("float32", "32"), ("float64", "64")):
        vals, grads = [], []
        for b0 in range(0, len(ij), B):
            sub = ij[b0 : b0 + B]
            g = R.opt_value_and_grads(kind, xys, phi, tx, x[sub[:, 1]][None], x[sub[:, 0]][None], cands, theta0, solver="min", steps=steps,
                                      dtype=dt, approx=True)
            vals.append(np.asarray(g["value"])[0])
            grads.append(np.asarray(g["grad_cell"])[0])
            print(f"{dt}: {b0 + len(sub)} of {len(ij)} cells, {time.time() - t0:.0f} s", flush=True)
        out["value" + tag] = np.concatenate(vals)
        out["grad_cell" + tag] = np.concatenate(grads)
    # Conditioning, from the oracle alone: the same fp32 chain with its inputs nudged by an ulp (the receiver, the transmitter,
    # the starting points).  A cell whose result moves under such a nudge amplifies round-off -- any two fp32 evaluations of
    # the reference's formulas (XLA's, torch's, a hand-derived gradient) may land on different solutions there.
    nudged = []
    up = lambda a: np.nextafter(np.asarray(a, F), F(np.inf)).astype(F)
    dn = lambda a: np.nextafter(np.asarray(a, F), F(-np.inf)).astype(F)
    for label, (xs, txn, th0) in (("rx+", (up(x), tx, theta0)), ("tx-, theta0+", (x, dn(tx), [up(t) for t in theta0]))):
        vals, grads = [], []
        for b0 in range(0, len(ij), B):
            sub = ij[b0 : b0 + B]
            g = R.opt_value_and_grads(kind, xys, phi, txn, xs[sub[:, 1]][None], xs[sub[:, 0]][None], cands, th0, solver="min", steps=steps,
                                      dtype="float32", approx=True)
            vals.append(np.asarray(g["value"])[0])
            grads.append(np.asarray(g["grad_cell"])[0])
            print(f"nudged ({label}): {b0 + len(sub)} of {len(ij)} cells, {time.time() - t0:.0f} s", flush=True)
        nudged.append((np.concatenate(vals), np.concatenate(grads)))
    out["grad_finite"] = np.isfinite(out["grad_cell32"]).all(-1)
    # `stable`: the oracle's own fp32 run AND its two nudged fp32 runs agree with its fp64 run (value within 2e-3, gradient
    # within 1e-2 of the cell's gradient scale) -- elsewhere the derivative through 1000 Adam steps is ill-conditioned and the
    # reference's fp32 result is noise (next to the corners it overflows: gradients of 1e30 and NaN in neighbouring cells)
    v64, v32, g64, g32 = out["value64"], out["value32"], out["grad_cell64"], out["grad_cell32"]
    fin64 = np.isfinite(g64).all(-1)
    gscale = np.maximum(np.abs(np.nan_to_num(g64)).max(-1), np.median(np.abs(g64[fin64]).max(-1)))
    with np.errstate(invalid="ignore"):
        out["stable"] = (np.isclose(v32, v64, rtol=2e-3, atol=2e-3 * np.abs(v64).max()) & out["grad_finite"] & fin64
                         & (np.abs(g32 - g64).max(-1) <= 1e-2 * gscale))
    for vn, gn in nudged:
        with np.errstate(invalid="ignore"):
            out["stable"] &= (np.isclose(vn, v64, rtol=2e-3, atol=2e-3 * np.abs(v64).max()) & np.isfinite(gn).all(-1)
                              & (np.abs(gn - g64).max(-1) <= 1e-2 * gscale))
    agree = solver_agreement(kind, xys, phi, tx, x[ij[:, 1]][None], x[ij[:, 0]][None], cands, theta0, steps)[0]
    print("cells whose solvers agree (fp32, nudged fp32, fp64):", int(agree.sum()), "-- of the", int(out["stable"].sum()),
          "value/gradient-stable ones:", int((agree & out["stable"]).sum()), flush=True)
    out["stable"] &= agree
    print("cells:", len(ij), "with a non-finite fp32 gradient:", int((~out["grad_finite"]).sum()), "non-finite in fp64:", int((~fin64).sum()),
          "stable:", int(out["stable"].sum()))
    path = os.path.join(ROOT, "tests", "golden", "cfg5_edges.npz")
    np.savez_compressed(path, **out)
    print(path)


if __name__ == "__main__":
    main()
