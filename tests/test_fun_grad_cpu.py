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
    # a constant function on the tape route: zero derivative, broadcast value
    val, bar = value_and_xys_bar(lambda tx, rx, path, objects: 3.0, fixed, grid, True, xys, np.zeros((4, 3), F), [], (), None, Point, Path)
    assert (val == 3.0).all() and not bar.any()


def test_a_function_the_tape_cannot_follow_is_refused():
    from differt2d_amd import _lib as L

    grid = np.zeros((2, 2, 2), F)
    args = (np.zeros(2, F), grid, True, np.ones((2, 2, 2, 2), F), np.zeros((2, 2), F), [], (), None, Point, Path)
    for bad in (lambda tx, rx, path, objects: np.sort(path.length()),            # no rule for np.sort
                lambda tx, rx, path, objects: float(path.length().sum()),        # the derivative would be lost
                lambda tx, rx, path, objects: np.asarray(path.xys)[..., 0, 0],   # idem
                lambda tx, rx, path, objects: path.length() * path.loss):        # the reference differentiates THROUGH loss
        with pytest.raises(L.D2DUnsupported):
            value_and_xys_bar(bad, *args)


def test_numpy_functions_on_the_tape_against_finite_differences():
    """A function written with NumPy calls (which a torch tape could not follow): ufuncs, np.sum / where / stack / linalg.norm /
    minimum / maximum / clip, indexing, broadcasting against constants -- the tape's derivative against central differences of
    the same function in float64."""

    def fun(tx, rx, path, objects, c=0.7):
        seg = path.xys[..., 1:, :] - path.xys[..., :-1, :]
        ln = np.linalg.norm(seg, axis=-1)                                  # [..., k + 1]
        r = np.sum(ln, axis=-1)
        far = np.where(r > 1.0, np.log(r), r - 1.0)
        bend = np.stack([np.sin(path.xys[..., 0, 0]), np.cos(path.xys[..., -1, 1])], axis=-1).sum(-1)
        d = rx.xy - tx.xy
        return (np.exp(-c * r) + far + 0.1 * bend + np.minimum(ln[..., 0], 0.4) + np.maximum(ln[..., -1], 0.2)
                + np.clip(d[..., 0], -0.3, 0.3) ** 2 + np.sqrt(np.abs(d[..., 1]) + 1.0) + np.arctan2(d[..., 1], 2.0 + d[..., 0]) / (1.0 + r * r))

    rng = np.random.default_rng(3)
    for k, grid_is_rx in ((0, True), (2, False), (3, True)):
        grid = rng.random((4, 5, 2))
        fixed = rng.random(2)
        xys = rng.random((4, 5, k + 2, 2)) * 1.5
        a, b = (fixed, grid) if grid_is_rx else (grid, fixed)
        xys[..., 0, :] = a
        xys[..., -1, :] = b
        val, bar = value_and_xys_bar(fun, fixed.astype(F), grid.astype(F), grid_is_rx, xys.astype(F), np.zeros((4, 5), F), [], (), None, Point, Path)

        def f64(x):  # the same function on plain float64 arrays; the end points ARE rows 0 and k + 1 of the path
            tx, rx = Point.__new__(Point), Point.__new__(Point)
            object.__setattr__(tx, "xy", x[..., 0, :])
            object.__setattr__(rx, "xy", x[..., -1, :])
            pth = Path.__new__(Path)
            object.__setattr__(pth, "xys", x)
            return fun(tx, rx, pth, [])

        x0 = xys.astype(F).astype(np.float64)
        np.testing.assert_allclose(val, f64(x0), rtol=3e-6)
        want = np.zeros_like(x0)
        h = 1e-6
        for i in range(k + 2):
            for c in range(2):
                e = np.zeros_like(x0)
                e[..., i, c] = h
                want[..., i, c] = (f64(x0 + e) - f64(x0 - e)) / (2 * h)
        np.testing.assert_allclose(bar, want, rtol=2e-4, atol=2e-5)


def test_jax_conventions_at_ties_and_kinks():
    """jnp.minimum / maximum split a tie evenly, abs'(0) = 0, sqrt'(0) = inf, and where() sends a ZERO cotangent into the branch
    not taken -- which an infinite local derivative there turns into NaN, exactly the reference's own autodiff trap
    (geometry.py:1105); the "double where" idiom is clean, as under JAX."""
    grid = np.zeros((1, 1, 2), F)
    fixed = np.zeros(2, F)
    xys = np.zeros((1, 1, 2, 2), F)
    xys[..., 1, :] = [2.0, 0.0]
    run = lambda f: value_and_xys_bar(f, fixed, grid, False, xys, np.zeros((1, 1), F), [], (), None, Point, Path)[1][0, 0]  # noqa: E731
    x = lambda p: p.xys[..., 1, 0]  # noqa: E731 -- = 2
    y = lambda p: p.xys[..., 1, 1]  # noqa: E731 -- = 0
    assert run(lambda t, r, p, o: np.minimum(x(p), 2.0 * x(p) - 2.0))[1, 0] == 1.5          # tie: (1 + 2) / 2
    assert run(lambda t, r, p, o: np.maximum(x(p) * x(p), 4.0))[1, 0] == 2.0               # tie with a constant: 4 / 2
    assert run(lambda t, r, p, o: abs(y(p)) + x(p))[1].tolist() == [1.0, 0.0]
    assert np.isnan(run(lambda t, r, p, o: np.where(y(p) == 0.0, x(p), x(p) / y(p)))[1]).all()
    assert run(lambda t, r, p, o: np.where(y(p) == 0.0, x(p), x(p) / np.where(y(p) == 0.0, 1.0, y(p))))[1].tolist() == [1.0, 0.0]
    assert np.isinf(run(lambda t, r, p, o: np.sqrt(y(p)))[1, 1])
    assert run(lambda t, r, p, o: x(p) ** 3)[1, 0] == 12.0 and run(lambda t, r, p, o: x(p) ** 0)[1, 0] == 0.0
    assert np.isclose(run(lambda t, r, p, o: 2.0 ** x(p))[1, 0], 4.0 * np.log(2.0))


def test_the_package_imports_and_differentiates_without_torch():
    """no autodiff framework inside differt2d_amd/ -- `import torch` blocked in a fresh interpreter, the
    package imported, a user function differentiated."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        "class Block:\n"
        "    def find_spec(self, name, path=None, target=None):\n"
        "        if name.split('.')[0] in ('torch', 'jax'):\n"
        "            raise ImportError('blocked: ' + name)\n"
        "sys.meta_path.insert(0, Block())\n"
        "import numpy as np\n"
        "import differt2d_amd\n"
        "from differt2d_amd.fun_grad import value_and_xys_bar\n"
        "from differt2d_amd.geometry import Path, Point\n"
        "F = np.float32\n"
        "xys = np.random.default_rng(0).random((3, 4, 3, 2)).astype(F)\n"
        "val, bar = value_and_xys_bar(lambda t, r, p, o: p.length() ** 1.5, xys[0, 0, -1], xys[..., 0, :], False, xys, np.zeros((3, 4), F), [], (), None, Point, Path)\n"
        "assert np.isfinite(bar).all() and np.abs(bar).max() > 0\n"
        "assert not any(m.split('.')[0] in ('torch', 'jax') for m in sys.modules)\n"
        "print('ok')\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]
    src = os.path.join(root, "differt2d_amd")
    for name in os.listdir(src):
        if name.endswith(".py"):
            text = open(os.path.join(src, name)).read()
            assert "import torch" not in text and "import jax" not in text, name
