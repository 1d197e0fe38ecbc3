"""The C oracle's per-cell gradient (oracle/d2d_oracle_grad.c: forward-mode dual numbers, the reference's two reverse-mode
NaN traps stated as rules) against REVERSE-mode autodiff of oracle/ref.py under torch (where / minimum / maximum / logistic
with JAX's semantics): values bit for bit, NaN positions identical, finite entries within 1e-5 of the largest (the bar
north_star sets for value+grad) -- on the committed golden fixtures (fp64 autodiff, incl. their NaN cells), on live random
scenes in every mode / path function / grid role, and on lattice-snapped scenes where exact zeros are common.

CPU only; this is what makes the C oracle usable as the GPU's independent gradient checker at scale
(tests/test_gpu_grad.py::test_cfg3_rows_against_the_c_gradient_oracle)."""

import glob
import os

import numpy as np
import pytest

from conftest import random_scene, unit_grid
from oracle import c_oracle as CO
from oracle import ref as R

F = np.float32
GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith("cfg"))


def _check(grad, want, name, want32=None):
    """grad: C oracle (fp64 tangents of the fp32 chain); want: fp64 autodiff; want32: fp32 autodiff (NaN positions)."""
    nan_ref = np.isnan(want32 if want32 is not None else want)
    assert np.array_equal(np.isnan(grad), nan_ref), f"{name}: NaN positions differ ({int(np.isnan(grad).sum())} vs {int(nan_ref.sum())})"
    fin = ~nan_ref & np.isfinite(want)
    if not fin.any():
        return
    scale = float(np.abs(want[fin]).max())
    err = np.abs(grad[fin] - want[fin])
    bar = 1e-5 * scale + 1e-5 * np.abs(want[fin]) + 1e-7
    if want32 is not None:
        # the tangents are exact derivatives of the fp32 chain: where the reference's own fp32 autodiff sits further than
        # 1e-5 from fp64 (sigmoid at alpha = 100 amplifies every rounding of its argument), twice that distance is the bar
        bar = np.maximum(bar, 2.0 * np.abs(np.asarray(want32, np.float64)[fin] - want[fin]))
    assert (err <= bar).all(), f"{name}: max abs err {err.max():.3e} at scale {scale:.3e} ({int((err > bar).sum())} entries over the bar)"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_fixtures(path):
    d = np.load(path)
    kw = eval(str(d["kwargs"]))  # written by scripts/make_golden.py
    value, grad = CO.power_map_grad(d["walls"], d["tx"], d["X"], d["Y"], **kw)
    if kw.get("function") == "sigmoid":
        np.testing.assert_allclose(value, d["value"], rtol=2e-6, atol=1e-7)
    else:
        assert np.array_equal(value, d["value"])
    _check(grad, np.asarray(d["grad_rx"], np.float64), os.path.basename(path))


@pytest.mark.parametrize("approx,function", [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")])
@pytest.mark.parametrize("role,fun", [("rx", "received_power"), ("rx", "length_squared"), ("rx", "length"), ("rx", "one"),
                                      ("tx", "received_power"), ("tx", "length")])
def test_live_autodiff(approx, function, fun, role):
    tx, walls = random_scene(6, seed=31)
    X, Y = unit_grid(9, 7)
    kw = dict(min_order=0, max_order=2, approx=approx, function=function, fun=fun)
    want = R.power_map_value_and_grads(walls, tx, X, Y, dtype="float64", grid_role=role, **kw)
    want32 = R.power_map_value_and_grads(walls, tx, X, Y, dtype="float32", grid_role=role, **kw)
    value, grad = CO.power_map_grad(walls, tx, X, Y, grid_role=role, **kw)
    assert np.array_equal(value, CO.power_map(walls, tx, X, Y, grid_role=role, **kw), equal_nan=True)
    np.testing.assert_allclose(value, want["value"], rtol=2e-5, atol=1e-5)
    _check(grad, want["grad_rx"], f"{role} {fun}", want32["grad_rx"])


@pytest.mark.parametrize("seed", range(6))
def test_lattice_scenes_nan_positions(seed):
    """Walls snapped to a coarse lattice, cells on walls' lines, a transmitter on a lattice point: un == 0 and zero-length
    segments do occur; the NaN pattern must be reverse mode's (fp32 autodiff of ref.py), in every mode and both roles."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(3, 7))
    walls = (np.round(rng.random((n, 2, 2)) * 4) / 4).astype(F)
    walls[walls[:, 0].tolist() == walls[:, 1].tolist()] += F(0.125)  # (no zero-length walls here: every cell would be NaN)
    tx = (np.round(rng.random(2) * 8) / 8).astype(F)
    x = np.linspace(0, 1, 9).astype(F)
    X, Y = np.meshgrid(x, x[:7])
    role = "tx" if seed % 2 else "rx"
    seen = 0
    for approx, function in ((False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")):
        kw = dict(min_order=0, max_order=2, approx=approx, function=function, alpha=float(rng.choice([100.0, 16.0])))
        want = R.power_map_value_and_grads(walls, tx, X, Y, dtype="float64", grid_role=role, **kw)
        want32 = R.power_map_value_and_grads(walls, tx, X, Y, dtype="float32", grid_role=role, **kw)
        value, grad = CO.power_map_grad(walls, tx, X, Y, grid_role=role, **kw)
        assert np.array_equal(np.isnan(grad), np.isnan(want32["grad_rx"])), (seed, approx, function)
        seen += int(np.isnan(grad).any(-1).sum())
        # finite entries: against fp64 autodiff where that is finite too (a lattice scene sits ON the kinks of min / max,
        # where fp32 and fp64 runs may select different arguments: compare only where they agree to 1e-3)
        fin = np.isfinite(want["grad_rx"]) & np.isfinite(want32["grad_rx"]) & np.isfinite(grad)
        same = fin & (np.abs(want["grad_rx"] - want32["grad_rx"]) <= 1e-3 * (1.0 + np.abs(want["grad_rx"])))
        if same.any():
            scale = float(np.abs(want["grad_rx"][same]).max()) + 1e-30
            assert np.abs(grad[same] - want32["grad_rx"][same]).max() <= 2e-3 * scale + 1e-6
    assert seen > 0 or seed not in (0, 1), "the lattice scenes were meant to hit the NaN rules"


def test_prune_levels_agree():
    """orc_params.prune skips work whose value and tangent are exactly zero: same maps, same gradients, same NaN cells."""
    for seed, lattice in ((5, False), (6, True), (7, True)):
        rng = np.random.default_rng(seed)
        tx, walls = random_scene(12, seed=seed)
        if lattice:
            walls = (np.round(walls * 4) / 4).astype(F)
        X, Y = unit_grid(17, 13)
        for approx in (False, True):
            for role in ("rx", "tx"):
                kw = dict(min_order=0, max_order=2, approx=approx, grid_role=role, alpha=float(rng.choice([100.0, 50.0])))
                v0, g0 = CO.power_map_grad(walls, tx, X, Y, prune=0, **kw)
                v1, g1 = CO.power_map_grad(walls, tx, X, Y, prune=1, **kw)
                assert np.array_equal(v0, v1, equal_nan=True)
                assert np.array_equal(g0, g1, equal_nan=True)
