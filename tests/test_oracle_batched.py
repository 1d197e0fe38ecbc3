"""The candidate-batched evaluation of the oracle (oracle/ref.py: facc_batched and friends, used to generate the
gradient fixtures of the 50-wall scene) against the candidate-by-candidate functions that the reference's known answers
pin: values bit for bit, reverse-mode gradients to fp64 round-off, NaN positions identical."""

import numpy as np
import pytest

from conftest import random_scene, unit_grid
from oracle import ref as R

F = np.float32
MODES = [dict(approx=False), dict(approx=True, function="hard_sigmoid"), dict(approx=True, function="sigmoid")]


@pytest.mark.parametrize("mode", MODES, ids=["hard", "hsig", "sig"])
@pytest.mark.parametrize("role", ["rx", "tx"])
def test_values_bit_identical(mode, role):
    tx, walls = random_scene(7, seed=3)
    X, Y = unit_grid(9, 7)
    kw = dict(min_order=0, max_order=2, grid_role=role, patch=0.01, **mode)
    a = R.power_map(walls, tx, X, Y, **kw)
    b = R.power_map_batched(walls, tx, X, Y, **kw)
    assert a.dtype == b.dtype == np.float32 and np.array_equal(a, b, equal_nan=True)


def test_filter_and_order3():
    tx, walls = random_scene(5, seed=9)
    X, Y = unit_grid(5, 4)
    kw = dict(min_order=1, max_order=3, filter_nodes=(2,), approx=True, alpha=50.0, tol=0.05, fun="length")
    assert np.array_equal(R.power_map(walls, tx, X, Y, **kw), R.power_map_batched(walls, tx, X, Y, **kw), equal_nan=True)


@pytest.mark.parametrize("mode", MODES, ids=["hard", "hsig", "sig"])
@pytest.mark.parametrize("role", ["rx", "tx"])
def test_gradients_match_looped_autodiff(mode, role):
    tx, walls = random_scene(6, seed=17)
    X, Y = unit_grid(7, 5)
    rng = np.random.default_rng(1)
    cot = rng.random(X.shape) + 0.5
    kw = dict(min_order=0, max_order=2, grid_role=role, **mode)
    a = R.power_map_value_and_grads(walls, tx, X, Y, cotangent=cot, dtype="float64", **kw)
    b = R.power_map_value_and_grads_batched(walls, tx, X, Y, cotangent=cot, dtype="float64", chunk=16, **kw)
    for k in ("value", "grad_rx", "tx_bar", "walls_bar"):
        assert np.array_equal(np.isnan(a[k]), np.isnan(b[k])), k
        np.testing.assert_allclose(b[k], a[k], rtol=1e-10, atol=1e-12 * max(1.0, float(np.nanmax(np.abs(a[k])))), err_msg=k)


def test_nan_traps_survive_batching():
    """basic_scene has collinear walls: the reference's own order-2 gradient is NaN in every cell (un == 0 `where` trap)."""
    walls = R.basic_scene_walls()
    tx = np.array([0.1, 0.1], F)
    X, Y = unit_grid(4, 3)
    kw = dict(min_order=2, max_order=2, approx=True)
    a = R.power_map_value_and_grads(walls, tx, X, Y, dtype="float32", **kw)
    b = R.power_map_value_and_grads_batched(walls, tx, X, Y, dtype="float32", **kw)
    assert np.isnan(a["grad_rx"]).any()
    assert np.array_equal(np.isnan(a["grad_rx"]), np.isnan(b["grad_rx"]))
    assert np.array_equal(np.isnan(a["walls_bar"]), np.isnan(b["walls_bar"]))
