"""The C oracle (oracle/d2d_oracle.c, -ffp-contract=off) must equal the NumPy restatement
(oracle/ref.py) bit for bit -- the NumPy one is what the reference's known answers pin."""

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import ref as R

from conftest import random_scene, unit_grid

F = np.float32
MODES = [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")]


@pytest.mark.parametrize("approx,function", MODES)
def test_cfg1_square_scene_64(approx, function):
    # BASELINE.json configs[0]: Scene.square_scene(), 64x64 RX grid, order <= 1
    walls, tx = R.square_scene_walls(), np.array([0.2, 0.2], F)
    X, Y = unit_grid(64)
    a = R.power_map(walls, tx, X, Y, 0, 1, approx=approx, function=function)
    b = CO.power_map(walls, tx, X, Y, min_order=0, max_order=1, approx=approx, function=function)
    if function == "sigmoid":
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)
    else:
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("approx,function", MODES)
@pytest.mark.parametrize("fun", ["received_power", "one", "length_squared"])
def test_random_scene_order2(approx, function, fun):
    tx, walls = random_scene(9, seed=7)
    X, Y = unit_grid(12, 10)
    a = R.power_map(walls, tx, X, Y, 0, 2, approx=approx, function=function, fun=fun)
    b = CO.power_map(walls, tx, X, Y, min_order=0, max_order=2, approx=approx, function=function, fun=fun)
    c = CO.power_map(walls, tx, X, Y, min_order=0, max_order=2, approx=approx, function=function, fun=fun, prune=True)
    assert np.array_equal(b, c, equal_nan=True)  # pruning is exact
    if function == "sigmoid":
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-5)
    else:
        assert np.array_equal(a, b, equal_nan=True)


def test_order3_patch_alpha_filter():
    tx, walls = random_scene(6, seed=3)
    X, Y = unit_grid(7, 5)
    allowed = np.array([1, 1, 0, 1, 1, 1], np.uint8)
    kw = dict(approx=True, function="hard_sigmoid", alpha=50.0, patch=0.01, tol=0.05)
    a = R.power_map(walls, tx, X, Y, 1, 3, filter_nodes=(2,), **kw)
    b = CO.power_map(walls, tx, X, Y, min_order=1, max_order=3, allowed=allowed, **kw)
    assert np.array_equal(a, b, equal_nan=True)


def test_notebook_count_through_c_oracle():
    # 6 valid / 50 invalid (notebook cell 6) through the C restatement
    walls = R.square_scene_with_obstacle_walls()
    valid, fun, idx = CO.eval_candidates(walls, [0.2, 0.2], [0.5, 0.6], order=2, fun="one")
    assert valid.shape == (56,) and int(valid.sum()) == 6
    assert [tuple(r[:2]) for r in idx[valid > 0]] == [(0, 1), (0, 2), (1, 3), (2, 6), (3, 1), (3, 2)]


def test_degenerate_rx_on_walls_and_tx():
    # RX exactly on walls, on wall end points and on the TX: no crash, identical results
    walls, tx = R.square_scene_with_wall_walls(), np.array([0.2, 0.5], F)
    x = np.array([0.0, 0.2, 0.5, 1.0], F)
    X, Y = np.meshgrid(x, np.array([0.0, 0.2, 0.5, 0.8, 1.0], F))
    for approx, function in MODES[:2]:
        a = R.power_map(walls, tx, X, Y, 0, 2, approx=approx, function=function)
        b = CO.power_map(walls, tx, X, Y, min_order=0, max_order=2, approx=approx, function=function)
        assert np.array_equal(a, b, equal_nan=True)
