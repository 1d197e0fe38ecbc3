#!/usr/bin/env python3
"""Generates tests/golden/cfg5_samples.npz: BASELINE.json configs[4] -- Scene.square_scene() + RIS([[0.5, 0.3], [0.5, 0.7]],
phi = pi / 4) (examples/plot_ris_power_map.py:38-43) + the RIS's two end points as diffraction Vertex objects, 300 x 300
grid (scene.grid(n=300)), order 1 (7 candidates), MinPath with 1000 Adam steps, hard_sigmoid validity -- on a sample of
cells: values and reverse-mode gradients THROUGH the solver (oracle/ref.py: opt_value_and_grads, torch double backward)
in fp64, and the same chain in fp32 (what fp32 round-off alone does to every entry after 1000 sequential steps).

theta0 (one start per candidate, shared by all cells as in the reference, scene.py:1887-1890) is drawn from NumPy's
default_rng(1234) -- jax.random cannot be reproduced here -- and stored in the fixture.

Run from the repo root (a few minutes):  python scripts/make_golden_cfg5.py [n_cells]
"""

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref as R  # noqa: E402

F = np.float32


def scene_tables():
    walls = R.square_scene_walls()
    ris = np.array([[0.5, 0.3], [0.5, 0.7]], F)
    xys = np.concatenate([walls, ris[None], np.stack([ris[0], ris[0]])[None], np.stack([ris[1], ris[1]])[None]]).astype(F)
    kind = np.array([R.WALL] * 4 + [R.RIS, R.VERTEX, R.VERTEX], np.uint8)
    phi = np.full(7, np.pi / 4, F)
    return xys, kind, phi


def solver_agreement(kind, xys, phi, tx, X, Y, cands, theta0, steps, solver="min", tol=2e-5):
    """Cells where the SOLVER of every candidate follows the same trajectory in the oracle's fp32 run, in its fp32 run from
    inputs nudged by one ulp, and in its fp64 run: interaction points within `tol` after 30, 100, 300 and all `steps`
    iterations.  Agreement of the summed value alone can be an accident: a candidate whose solver wanders chaotically next to
    a wall usually ends somewhere invalid in every run and contributes 0 each time -- until an evaluation with another
    rounding (XLA's, a hand-derived gradient) happens to end on the valid reflection point.  And agreement of the FINAL points
    alone says little about the gradient: Adam's second moment remembers the transient (0.999 ** 700 = 0.5), so the
    derivative through the loop depends on the whole trajectory -- a grazing reflection whose fp32 and fp64 transients are
    1e-4 apart ends on the same point with gradients that differ in the third digit."""
    up = lambda a: np.nextafter(np.asarray(a, F), F(np.inf)).astype(F)
    ok = np.ones(np.shape(X), bool)
    objs64 = [R.Obj(int(k), np.asarray(xys[j, 0] if int(k) == R.VERTEX else xys[j], np.float64), float(phi[j])) for j, k in enumerate(kind)]
    objs32 = [R.Obj(int(k), np.asarray(xys[j, 0] if int(k) == R.VERTEX else xys[j], F), float(phi[j])) for j, k in enumerate(kind)]
    rx32 = np.stack([X, Y], -1).astype(F)
    tx32 = np.broadcast_to(np.asarray(tx, F), rx32.shape)
    for c, th in zip(cands, theta0):
        if len(th) == 0:
            continue
        for n_steps in sorted({s_ for s_ in (30, 100, 300, steps) if s_ <= steps}):
            ref, _ = R.opt_path(solver, tx32.astype(np.float64), [objs64[int(i)] for i in c], rx32.astype(np.float64), th, n_steps, R.NUMPY64)
            for t_, r_, th_ in ((tx32, rx32, th), (tx32, up(rx32), th), (np.broadcast_to(up(tx), rx32.shape), rx32, up(th))):
                got, _ = R.opt_path(solver, t_, [objs32[int(i)] for i in c], r_, th_, n_steps, R.NUMPY)
                for a, b in zip(got[1:-1], ref[1:-1]):
                    with np.errstate(invalid="ignore"):
                        ok &= np.abs(np.asarray(a, np.float64) - b).max(-1) <= tol
    return ok


def main():
    n_cells = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    xys, kind, phi = scene_tables()
    tx = np.array([0.2, 0.2], F)  # scene.py:836
    x = np.linspace(0.0, 1.0, 300).astype(F)
    rng = np.random.default_rng(1234)
    cands = R.all_path_candidates(7, order=1)
    theta0 = [rng.random(0 if kind[int(c[0])] == R.VERTEX else 1, dtype=F) for c in cands]
    ij = np.stack([rng.integers(1, 299, n_cells), rng.integers(1, 299, n_cells)], 1).astype(np.int32)
    X, Y = x[ij[:, 1]][None], x[ij[:, 0]][None]  # (1, n_cells)
    out = dict(ij=ij, theta0=np.array([t[0] if len(t) else np.nan for t in theta0], F), steps=np.int32(steps), xys=xys, kind=kind, phi=phi,
               tx=tx)
    res = {}
    for dt in ("float64", "float32"):
        t = time.time()
        g = R.opt_value_and_grads(kind, xys, phi, tx, X, Y, cands, theta0, solver="min", steps=steps, dtype=dt, approx=True)
        res[dt] = g
        tag = "64" if dt == "float64" else "32"
        out["value" + tag], out["grad_cell" + tag] = np.asarray(g["value"]), np.asarray(g["grad_cell"])
        print(dt, f"{time.time() - t:.0f}s", "value range", float(g["value"].min()), float(g["value"].max()), "NaN gradient cells",
              int(np.isnan(g["grad_cell"]).any(-1).sum()), flush=True)
    # The derivative through 1000 Adam steps is ill-conditioned in some cells (as nu decays the update's sensitivity to the
    # gradient grows like 1 / (sqrt(nu) + eps)): there the oracle's OWN fp32 run is orders of magnitude away from its fp64
    # run, i.e. the reference's fp32 gradient is noise.  The scene VJP is a sum over cells, so it is taken over the cells
    # where the two runs agree (value within 2e-3, gradient within 1e-2 of the cell's gradient scale): `stable`.
    v64, v32 = res["float64"]["value"][0], res["float32"]["value"][0]
    g64, g32 = res["float64"]["grad_cell"][0], res["float32"]["grad_cell"][0]
    gscale = np.maximum(np.abs(g64).max(-1), np.median(np.abs(g64).max(-1)))
    stable = (np.isclose(v32, v64, rtol=2e-3, atol=2e-3 * np.abs(v64).max()) & np.isfinite(g32).all(-1)
              & (np.abs(g32 - g64).max(-1) <= 1e-2 * gscale))
    agree = solver_agreement(kind, xys, phi, tx, X, Y, cands, theta0, steps)[0]
    print("cells whose solvers agree (fp32, nudged fp32, fp64):", int(agree.sum()), "of", n_cells, "-- of the", int(stable.sum()),
          "value/gradient-stable ones:", int((agree & stable).sum()), flush=True)
    stable &= agree
    out["stable"] = stable
    print("stable cells:", int(stable.sum()), "of", n_cells, flush=True)
    for dt in ("float64", "float32"):
        g = R.opt_value_and_grads(kind, xys, phi, tx, X[:, stable], Y[:, stable], cands, theta0, solver="min", steps=steps, dtype=dt,
                                  approx=True)
        tag = "64" if dt == "float64" else "32"
        for k in ("fixed_bar", "xys_bar", "phi_bar"):
            out[k + tag] = np.asarray(g[k])
    path = os.path.join(ROOT, "tests", "golden", "cfg5_samples.npz")
    np.savez_compressed(path, **out)
    print(path)


if __name__ == "__main__":
    main()
