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
    for level in (1, 2):
        c = CO.power_map(walls, tx, X, Y, min_order=0, max_order=2, approx=approx, function=function, fun=fun, prune=level)
        assert np.array_equal(b, c, equal_nan=True), level  # pruning is exact
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


@pytest.mark.parametrize("approx,function", MODES)
@pytest.mark.parametrize("role", ["rx", "tx"])
def test_prune_levels_are_exact_on_dense_scenes(approx, function, role):
    """Prune level 2 (leave a candidate at its first wall whose on_objects term is exactly 0; stop the occlusion tests at the
    first one that is exactly True / saturated) is what makes the full-size cfg4 blocks affordable
    (scripts/make_golden_cfg4_blocks.py).  Cell for cell the same bits as the literal evaluation (level 0), also with cells
    on walls, on end points and on the transmitter, lattice-snapped and collinear walls, zero-length walls, orders 0..3."""
    rng = np.random.default_rng(11)
    for n, order, snap in ((14, 3, False), (25, 2, False), (12, 3, True), (40, 2, True)):
        tx, walls = random_scene(n, seed=100 + n)
        if snap:
            walls = (np.round(walls * 8) / 8).astype(F)  # shared corners, collinear and zero-length walls
            tx = (np.round(tx * 8) / 8 + F(0.03)).astype(F)
        X, Y = unit_grid(9, 7)
        X, Y = X.copy(), Y.copy()
        X[0, 0], Y[0, 0] = tx                       # a cell on the fixed end point
        X[1, 1], Y[1, 1] = walls[0, 0]              # on a wall's end point
        X[2, 2], Y[2, 2] = 0.5 * (walls[1, 0] + walls[1, 1])  # on a wall
        kw = dict(min_order=0, max_order=order, approx=approx, function=function, grid_role=role, patch=float(rng.choice([0.0, 0.01])))
        ref = CO.power_and_count_maps(walls, tx, X, Y, prune=0, **kw)
        for level in (1, 2):
            got = CO.power_and_count_maps(walls, tx, X, Y, prune=level, **kw)
            assert np.array_equal(ref[0], got[0], equal_nan=True) and np.array_equal(ref[1], got[1], equal_nan=True), (n, order, snap, level)
