"""
Pins the oracle (oracle/ref.py) to every known answer the reference itself records for the
hot path: its unit tests, doctests and the executed notebook.
File:line citations are relative to the DiffeRT2d v0.4.0 checkout.
"""

import numpy as np
import pytest

from oracle import ref as R

F = np.float32


def A(*v):
    return np.array(v, dtype=F)


# ---- doctest scalars -----------------------------------------------------------------


def test_segments_intersect_doctest():
    # differt2d/geometry.py:142-151 and tests/test_geometry.py:101-120
    P = [A(0, 0), A(1, 0), A(0.5, -1), A(0.5, 1)]
    assert R.segments_intersect(*P, approx=True) == F(1.0)
    assert bool(R.segments_intersect(*P, approx=False)) is True
    assert R.segments_intersect(*P, approx=True, function="sigmoid") == F(1.0)
    Q = [A(0, 0), A(1, 0), A(0, 1), A(1, 1)]
    assert R.is_false(R.segments_intersect(*Q, approx=True), True)
    assert R.is_false(R.segments_intersect(*Q, approx=False), False)


def test_path_length_doctest():
    # differt2d/geometry.py:195-197 ; tests/test_geometry.py:123-135 (exactly 4.0)
    assert R.path_length([A(0, 0), A(1, 0), A(1, 1), A(0, 0)]) == F(3.4142137)
    assert R.path_length([A(0, 0), A(1, 0), A(1, 1), A(0, 1), A(0, 0)]) == F(4.0)


def test_normalize_doctest():
    # differt2d/geometry.py:217-225
    v, l = R.normalize(A(1, 1))
    assert np.array_equal(v, A(0.70710677, 0.70710677)) and l == F(1.4142135)
    v, l = R.normalize(A(0, 0))
    assert np.array_equal(v, A(0, 0)) and l == F(1.0)


def test_image_of_doctest():
    # differt2d/geometry.py:663-667
    w = np.array([[0, 0], [1, 0]], dtype=F)
    assert np.array_equal(R.wall_image_of(w, A(0, 1)), A(0, -1))


def test_cartesian_to_parametric_table():
    # tests/test_geometry.py:290-311
    w = np.array([[0, 0], [4, 2]], dtype=F)
    for p, s in [((2, 1), 0.5), ((0, 0), 0.0), ((4, 2), 1.0), ((8, 4), 2.0), ((-4, -2), -1.0)]:
        assert R.wall_cartesian_to_parametric(w, A(*p)) == F(s)


@pytest.mark.parametrize("approx", [True, False])
def test_contains_and_intersects(approx):
    # tests/test_geometry.py:313-342
    w = np.array([[0, 0], [4, 2]], dtype=F)
    assert R.is_true(R.wall_contains_parametric(F(0.5), approx), approx)
    assert R.is_false(R.wall_contains_parametric(F(2.0), approx), approx)
    hit = lambda a, b: R.wall_intersects_cartesian(w, A(*a), A(*b), approx=approx)
    assert R.is_true(hit((0, 2), (4, 0)), approx)
    assert R.is_false(hit((0, 1), (4, 3)), approx)
    assert R.is_false(hit((0, 1), (2, 7)), approx)
    got = hit((0, 1), (0, 0))  # touches the extremity
    assert (got > 0) if approx else bool(got)


def test_evaluate_cartesian_wall_and_ris():
    # tests/test_geometry.py:344-376
    w = np.array([[0, 0], [4, 0]], dtype=F)
    assert abs(R.wall_evaluate_cartesian(w, A(0, 1), A(2, 0), A(4, 1))) < 1e-6
    assert abs(R.wall_evaluate_cartesian(w, A(0, 1), A(2.1, 0), A(4, 1))) > 1e-5
    assert abs(R.ris_evaluate_cartesian(w, F(0.0), A(0, 1), A(2, 0), A(2, 1))) < 1e-6
    assert abs(R.ris_evaluate_cartesian(w, F(0.0), A(0, 1), A(2, 0), A(4, 1))) > 1e-5


def test_received_power():
    # tests/test_utils.py:8-22 : 0.3 / (2*2)
    got = R.received_power([A(0, 0), A(1, 0), A(1, 1)], r_coef=0.3, height=0.0)
    np.testing.assert_allclose(got, 0.3 / 4.0, rtol=1e-6)


# ---- scenes --------------------------------------------------------------------------


def test_scene_shapes():
    # differt2d/scene.py:750-759, 804-813, 856-865, 901-910
    assert R.square_scene_walls().shape == (4, 2, 2)
    assert R.square_scene_with_wall_walls().shape == (5, 2, 2)
    assert R.square_scene_with_obstacle_walls().shape == (8, 2, 2)
    assert R.basic_scene_walls().shape == (7, 2, 2)


# ---- candidate enumeration (differt-core 0.0.31, not in the tree) ----------------------


def test_candidates_order0_and_filter():
    # tests/test_scene.py:372-399
    got = R.all_path_candidates(1, 0, 0)
    assert len(got) == 1 and len(got[0]) == 0
    got = R.all_path_candidates(1, order=0)
    assert len(got) == 1 and len(got[0]) == 0
    got = R.all_path_candidates(6, 0, 2, filter_nodes=(0, 1, 2, 4, 5))
    assert [list(map(int, c)) for c in got] == [[], [3]]
    assert got[0].dtype == np.int32


def test_candidates_recorded_order_n7_k2():
    # docs/source/notebooks/cost20120_helsinki_model.ipynb, cell 20 recorded output:
    # [0,1],[0,2],...,[0,6],[1,0],[1,2],... lexicographic, never [i,i]
    got = [tuple(map(int, c)) for c in R.all_path_candidates(7, order=2)]
    want = [(i, j) for i in range(7) for j in range(7) if i != j]
    assert got == want and len(got) == 42


def test_candidate_counts():
    # c_k = N (N-1)^(k-1): 8*7 = 56 (notebook cell 6); cfg2 of BASELINE.json: 2501
    assert len(R.all_path_candidates(8, order=2)) == 56
    assert len(R.all_path_candidates(50, 0, 2)) == 2501


# ---- image path ----------------------------------------------------------------------


def test_image_path_loss_is_zero_square_scene():
    # tests/test_geometry.py:493-500 : loss <= 1e-13 over all 4 walls of square_scene()
    objs = R.walls_to_objs(R.square_scene_walls())
    pts, loss = R.image_path(A(0.2, 0.2), objs, A(0.5, 0.6))
    assert abs(loss) <= 1e-13
    assert len(pts) == 6


@pytest.mark.parametrize("approx", [True, False])
def test_is_valid_square_scene(approx):
    # tests/test_geometry.py:451-467 (ImagePath column)
    objs = R.walls_to_objs(R.square_scene_walls())
    pts, loss = R.image_path(A(0.2, 0.2), objs, A(0.5, 0.6))
    v = R.is_valid(objs, [0, 1, 2, 3], objs, pts, loss, approx=approx)
    assert R.is_true(v, approx)


def test_notebook_6_valid_50_invalid():
    # docs/source/notebooks/cost20120_helsinki_model.ipynb cell 6 recorded output:
    # "Found 6 valid path candidates, and 50 invalid path candidates" on
    # Scene.square_scene_with_obstacle(), order=2, hard mode
    objs = R.walls_to_objs(R.square_scene_with_obstacle_walls())
    tx, rx = A(0.2, 0.2), A(0.5, 0.6)
    valid = []
    for cand in R.all_path_candidates(8, order=2):
        v, _, _, _ = R.accumulate_candidate(tx, objs, cand, rx, approx=False)
        if bool(v):
            valid.append(tuple(map(int, cand)))
    assert len(valid) == 6
    assert valid == [(0, 1), (0, 2), (1, 3), (2, 6), (3, 1), (3, 2)]


def test_no_object_path_length():
    # tests/test_geometry.py:391-400
    pts, loss = R.image_path(A(0, 1), [], A(2, 1))
    np.testing.assert_allclose(R.path_length(pts), 2.0, rtol=1e-6)


def test_midpoint_path():
    # tests/test_geometry.py:380-389 : base Path on one wall has length 2*sqrt(2)
    w = R.Obj(R.WALL, np.array([[0, 0], [2, 0]], dtype=F))
    pts, _ = R.midpoint_path(A(0, 1), [w], A(2, 1))
    np.testing.assert_allclose(R.path_length(pts), 2 * np.sqrt(2), rtol=1e-6)


# ---- LOS analytics (tests/test_scene.py:443-627) ---------------------------------------


def _los_grid():
    x = np.linspace(-3, 3, 10).astype(F)
    return np.meshgrid(x, x)


def test_los_receivers_grid_values():
    X, Y = _los_grid()
    Z0 = R.power_map(np.zeros((0, 2, 2), F), A(0, 0), X, Y, max_order=1, fun="length_squared")
    Z1 = R.power_map(np.zeros((0, 2, 2), F), A(1, 0), X, Y, max_order=1, fun="length_squared")
    np.testing.assert_allclose(Z0, X**2 + Y**2, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(Z1, (X - 1) ** 2 + Y**2, rtol=1e-6, atol=1e-6)
    assert Z0.shape == X.shape and Z0.dtype == np.float32


def test_los_receivers_grid_gradients():
    X, Y = _los_grid()
    out = R.power_map_value_and_grads(np.zeros((0, 2, 2), F), A(1, 0), X, Y, max_order=1, fun="length_squared")
    want = np.stack([2 * (X - 1), 2 * Y], axis=-1)
    np.testing.assert_allclose(out["grad_rx"], want, rtol=1e-5, atol=1e-5)
    # gradient w.r.t. the transmitter of sum(Z): -sum of per-cell gradients
    np.testing.assert_allclose(out["tx_bar"], -want.reshape(-1, 2).sum(0), rtol=1e-4, atol=1e-3)


def test_pairwise_los_sums():
    # tests/test_scene.py:443-485 : 2, 1, 1, 2 and their sum 6
    txs = [A(0, 0), A(1, 0)]
    rxs = [A(1, 1), A(0, 1)]
    vals = [R.facc(t, [], R.all_path_candidates(0, 0, 1), r, fun="length_squared") for t in txs for r in rxs]
    np.testing.assert_allclose(vals, [2, 1, 1, 2], rtol=1e-6)
    np.testing.assert_allclose(sum(vals), 6.0, rtol=1e-6)


# ---- optimisers / MinPath / FermatPath ------------------------------------------------


def test_adam_convex():
    # tests/test_optimize.py:27-40 (x0 = [1,2,3] -> 0.5, loss 2.0, steps=1000, rtol 1e-3)
    import torch

    tb = R.TorchBackend("float32")

    def vg(x):
        xs = [v.detach().clone().requires_grad_(True) for v in x]
        loss = sum((v - 0.5) * (v - 0.5) for v in xs) + 2.0
        g = torch.autograd.grad(loss, xs)
        return loss.detach(), [gi.detach() for gi in g]

    x, loss = R.adam_minimize(vg, [torch.tensor(v, dtype=torch.float32) for v in (1.0, 2.0, 3.0)], steps=1000, xp=tb)
    np.testing.assert_allclose([float(v) for v in x], [0.5] * 3, rtol=1e-3)
    np.testing.assert_allclose(float(loss), 2.0, rtol=1e-3)


@pytest.mark.parametrize("solver", ["min", "fermat"])
def test_single_reflection_opt_paths(solver):
    # tests/test_geometry.py:503-525 : reflection point (1, 0) within rtol 1e-2, MinPath loss <= 1e-4
    w = R.Obj(R.WALL, np.array([[0, 0], [2, 0]], dtype=F))
    pts, loss = R.opt_path(solver, A(0, 1), [w], A(2, 1), theta0=[0.3], steps=1000)
    np.testing.assert_allclose(pts[1], [1.0, 0.0], rtol=1e-2, atol=1e-2)
    if solver == "min":
        assert abs(float(loss)) <= 1e-4
