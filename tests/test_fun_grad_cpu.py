"""differt2d_amd/fun_grad.py on the host alone: the derivative of a user's path function w.r.t. the path points, from a tape of
its operations and from a derivative the user supplies -- the half of the arbitrary-`fun` gradient (reference scene.py:1892-1923)
that needs no GPU (the other half, the kernels' adjoint, is tests/test_gpu_api.py's)."""

import numpy as np
import pytest

from differt2d_amd.fun_grad import value_and_xys_bar
from differt2d_amd.geometry import Path, Point

F = np.float32
EPS = float(np.finfo(np.float32).eps)


def _fun(tx, rx, path, objects, w=0.3):
    r = path.length()
    dx = rx.xy[..., 0] - tx.xy[..., 0]
    return w * r * r.sqrt() + dx * dx + path.xys[..., -2, 0] * rx.xy[..., 1]


def _analytic(fixed, grid, grid_is_rx, xys, w):
    xys = np.asarray(xys, np.float64)
    v = (xys[..., 1:, :] - xys[..., :-1, :]) + EPS
    ln = np.sqrt((v * v).sum(-1))
    r = ln.sum(-1)
    tx, rx = (fixed, grid) if grid_is_rx else (grid, fixed)
    tx, rx = np.broadcast_to(np.asarray(tx, np.float64), grid.shape), np.broadcast_to(np.asarray(rx, np.float64), grid.shape)
    dx = rx[..., 0] - tx[..., 0]
    val = w * r * np.sqrt(r) + dx * dx + xys[..., -2, 0] * rx[..., 1]
    bar = np.zeros_like(xys)
    u = (1.5 * w * np.sqrt(r))[..., None, None] * v / ln[..., None]
    bar[..., 1:, :] += u
    bar[..., :-1, :] -= u
    bar[..., -2, 0] += rx[..., 1]
    bar[..., 0, :] += np.stack([-2 * dx, np.zeros_like(dx)], -1)   # d / d tx.xy, folded into row 0
    bar[..., -1, :] += np.stack([2 * dx, xys[..., -2, 0]], -1)    # d / d rx.xy, folded into the last row
    return val, bar


@pytest.mark.parametrize("grid_is_rx", [True, False])
@pytest.mark.parametrize("k", [0, 1, 3])
def test_tape_against_the_analytic_derivative(k, grid_is_rx):
    pytest.importorskip("torch")
    rng = np.random.default_rng(k)
    grid = rng.random((5, 7, 2)).astype(F)
    fixed = rng.random(2).astype(F)
    xys = rng.random((5, 7, k + 2, 2)).astype(F)
    # (the end points of the path are the end points handed to fun, as in a traced path)
    a, b = (fixed, grid) if grid_is_rx else (grid, fixed)
    xys[..., 0, :] = a
    xys[..., -1, :] = b
    val, bar = value_and_xys_bar(_fun, fixed, grid, grid_is_rx, xys, np.zeros((5, 7), F), [], (), dict(w=0.25), Point, Path)
    want_val, want_bar = _analytic(fixed, grid, grid_is_rx, xys, 0.25)
    assert val.dtype == F and bar.dtype == F and bar.shape == xys.shape
    np.testing.assert_allclose(val, want_val, rtol=2e-6)
    np.testing.assert_allclose(bar, want_bar, rtol=2e-5, atol=2e-6)


def test_user_supplied_derivative_and_constants():
    grid = np.random.default_rng(1).random((4, 3, 2)).astype(F)
    fixed = np.array([0.2, 0.7], F)
    xys = np.random.default_rng(2).random((4, 3, 3, 2)).astype(F)

    def never(*a, **k):
        raise AssertionError("fun.value_and_grad should have been used")

    never.value_and_grad = lambda tx, rx, path, objects: (np.full(path.xys.shape[:-2], 2.0), np.ones_like(path.xys), np.array([1.0, 0.0]), np.array([0.0, 3.0]))
    val, bar = value_and_xys_bar(never, fixed, grid, True, xys, np.zeros((4, 3), F), [], (), None, Point, Path)
    assert (val == 2.0).all() and (bar[..., 1, :] == 1.0).all()
    assert (bar[..., 0, :] == np.array([2.0, 1.0], F)).all() and (bar[..., -1, :] == np.array([1.0, 4.0], F)).all()
    never.value_and_grad = lambda *a: (1.0,)
    with pytest.raises(TypeError):
        value_and_xys_bar(never, fixed, grid, True, xys, np.zeros((4, 3), F), [], (), None, Point, Path)
    pytest.importorskip("torch")
    # a constant function on the tape route: zero derivative, broadcast value
    val, bar = value_and_xys_bar(lambda tx, rx, path, objects: 3.0, fixed, grid, True, xys, np.zeros((4, 3), F), [], (), None, Point, Path)
    assert (val == 3.0).all() and not bar.any()


def test_a_function_the_tape_cannot_follow_is_refused():
    pytest.importorskip("torch")
    from differt2d_amd import _lib as L

    grid = np.zeros((2, 2, 2), F)
    with pytest.raises(L.D2DUnsupported):
        value_and_xys_bar(lambda tx, rx, path, objects: np.sqrt(path.length()), np.zeros(2, F), grid, True, np.ones((2, 2, 2, 2), F),
                          np.zeros((2, 2), F), [], (), None, Point, Path)
