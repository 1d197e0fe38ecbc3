"""
GPU tests of the Scene / Path API mirror.  They read like the reference's own tests
(tests/test_scene.py, tests/test_geometry.py of DiffeRT2d v0.4.0) with NumPy in place of JAX, and
add oracle comparisons where the reference has no numeric expectation.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    from differt2d_amd import _lib as L

    assert L.device_count() > 0


def _scene_walls(scene):
    return np.stack([o.xys for o in scene.objects]) if scene.objects else np.zeros((0, 2, 2), F)


# ---- reference tests/test_geometry.py -------------------------------------------------------


def test_image_path_loss_is_zero():
    # tests/test_geometry.py:493-500
    from differt2d_amd.geometry import ImagePath
    from differt2d_amd.scene import Scene
    from oracle import ref as R

    scene = Scene.square_scene()
    got = ImagePath.from_tx_objects_rx(scene.transmitters["tx"], scene.objects, scene.receivers["rx"])
    assert abs(float(got.loss)) <= 1e-13
    assert got.xys.shape == (6, 2)
    pts, loss = R.image_path(np.array([0.2, 0.2], F), R.walls_to_objs(R.square_scene_walls()), np.array([0.5, 0.6], F))
    assert np.array_equal(got.xys, np.stack(pts)) and got.loss == loss


def test_image_path_no_object_and_mixed_types():
    # tests/test_geometry.py:391-400, 90-98
    from differt2d_amd.geometry import RIS, ImagePath, Point, Wall

    path = ImagePath.from_tx_objects_rx(Point(xy=[0, 1]), [], Point(xy=[2, 1]))
    np.testing.assert_allclose(path.length(), 2.0, rtol=1e-6)
    with pytest.raises(ValueError):
        ImagePath.from_tx_objects_rx(Point(xy=[0, 1]), [Wall(), RIS()], Point(xy=[2, 1]))


@pytest.mark.parametrize("approx", [True, False])
def test_path_validity_methods(approx):
    # tests/test_geometry.py:402-467
    from differt2d_amd import logic
    from differt2d_amd.geometry import ImagePath, Path, Wall
    from differt2d_amd.scene import Scene

    with logic.enable_approx(approx):
        scene = Scene.random_uniform_scene(key=1234, n_walls=4)  # the reference uses 5; D2D_MAX_ORDER is 4
        path = Path.from_tx_objects_rx(scene.transmitters["tx_0"], scene.objects, scene.receivers["rx_0"])
        np.testing.assert_allclose(F(path.on_objects(scene.objects)), F(logic.true_value()), atol=1e-8)
        got = path.on_objects([Wall(xys=[[10.0, 10.0], [20.0, 20.0]])] * 4)
        np.testing.assert_allclose(F(got), F(logic.false_value()), atol=1e-8)

        scene = Scene.random_uniform_scene(key=1234, n_walls=10)
        path = Path.from_tx_objects_rx(scene.transmitters["tx_0"], scene.objects[:4], scene.receivers["rx_0"])
        got = path.intersects_with_objects(scene.objects, np.arange(4, dtype=np.int32))
        np.testing.assert_allclose(F(got), F(logic.true_value()), atol=1e-8)

        scene = Scene.square_scene()
        cand = np.arange(4, dtype=np.int32)
        path = Path.from_tx_objects_rx(scene.transmitters["tx"], scene.objects, scene.receivers["rx"])
        got = path.intersects_with_objects(scene.objects, cand)
        np.testing.assert_allclose(F(got), F(logic.false_value()), atol=1e-8)

        for cls in (Path, ImagePath):
            p = cls.from_tx_objects_rx(scene.transmitters["tx"], scene.objects, scene.receivers["rx"])
            assert logic.is_true(p.is_valid(scene.objects, cand, scene.get_interacting_objects(cand)))


# ---- reference tests/test_scene.py ------------------------------------------------------------


@pytest.mark.parametrize("min_order,max_order", [(0, 0), (1, 1), (2, 2), (0, 2)])
def test_all_paths_and_valid_paths(min_order, max_order):
    # tests/test_scene.py:401-441
    from differt2d_amd import logic
    from differt2d_amd.scene import Scene

    scene = Scene.square_scene()
    valid_paths = scene.all_valid_paths(approx=False, min_order=min_order, max_order=max_order, key=1234)
    n_valid = 0
    for tx_key, rx_key, got_valid, path, cand in scene.all_paths(min_order=min_order, max_order=max_order, approx=False):
        assert tx_key == "tx" and rx_key == "rx"
        assert min_order <= path.xys.shape[0] - 2 <= max_order and min_order <= len(cand) <= max_order
        expected = path.is_valid(scene.objects, cand, scene.get_interacting_objects(cand), approx=False)
        assert bool(got_valid) == bool(expected)
        if logic.is_true(got_valid, approx=False):
            _, _, got_path, _ = next(valid_paths)
            assert np.array_equal(got_path.xys, path.xys)
            n_valid += 1
    with pytest.raises(StopIteration):
        next(valid_paths)
    assert n_valid >= 1


def test_notebook_valid_count_via_all_paths():
    # docs/source/notebooks/cost20120_helsinki_model.ipynb cell 6: 6 valid, 50 invalid
    from differt2d_amd import logic
    from differt2d_amd.scene import Scene

    scene = Scene.square_scene_with_obstacle()
    flags = [bool(logic.is_true(v, approx=False)) for _, _, v, _, _ in scene.all_paths(min_order=2, max_order=2, approx=False)]
    assert len(flags) == 56 and sum(flags) == 6
    assert len(list(scene.all_valid_paths(order=2))) == 6


def _length_sq(transmitter, receiver, path, interacting_objects):
    # the reference's own local `fun` (tests/test_scene.py:444, 558-560): recognised by what it computes and fused natively
    return path.length() ** 2


def _length_sq_host(transmitter, receiver, path, interacting_objects):
    # an arbitrary python fun (it depends on the interacting objects): goes through the GPU trace + host fun path
    return path.length() ** 2 * (1.0 + 1e-3 * len(interacting_objects))


def test_accumulate_over_paths():
    # tests/test_scene.py:443-485
    from differt2d_amd.geometry import Point
    from differt2d_amd.scene import Scene

    scene = Scene(transmitters={"tx0": Point(xy=[0.0, 0.0]), "tx1": Point(xy=[1.0, 0.0])}, objects=[],
                  receivers={"rx0": Point(xy=[1.0, 1.0]), "rx1": Point(xy=[0.0, 1.0])})
    got = list(scene.accumulate_over_paths(fun=_length_sq, max_order=1, approx=False))
    assert [(a, b) for a, b, _ in got] == [("tx0", "rx0"), ("tx0", "rx1"), ("tx1", "rx0"), ("tx1", "rx1")]
    np.testing.assert_allclose([v for _, _, v in got], [2.0, 1.0, 1.0, 2.0], rtol=1e-6)
    total = scene.accumulate_over_paths(fun=_length_sq, reduce_all=True, max_order=1, approx=False)
    np.testing.assert_allclose(total, 6.0, rtol=1e-6)


@pytest.mark.parametrize("path_cls_name", ["MinPath", "FermatPath"])
def test_accumulate_over_paths_keyed_optimiser_classes_fused_equals_traced(path_cls_name):
    """with a Threefry key the reference hands every (pair, candidate) its own key from a chain of splits
    (scene.py:1204-1219).  The fused route (a recognised `fun`) and the traced route (`_d2d_native = False`) must draw the
    same initial guesses -- the same numbers whichever way `fun` runs -- and an explicit theta0 is shared by all pairs in both;
    no candidates at all: an empty iterator in both (the reference's groupby over no paths)."""
    import differt2d_amd.geometry as G
    from differt2d_amd.geometry import Point
    from differt2d_amd.random import PRNGKey
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power

    path_cls = getattr(G, path_cls_name)
    scene = Scene.square_scene_with_obstacle()
    scene = scene.with_transmitters(tx_a=Point(xy=np.array([0.5, 0.7], F)), tx_b=Point(xy=np.array([0.15, 0.35], F)))
    scene = scene.with_receivers(rx_0=Point(xy=np.array([0.3, 0.1], F)), rx_1=Point(xy=np.array([0.5, 0.1], F)),
                                 rx_2=Point(xy=np.array([0.83, 0.41], F)))
    host_fun = lambda t, r, p, o: received_power(t, r, p, o)  # noqa: E731
    host_fun._d2d_native = False
    # few steps: the solver has NOT converged, so the result depends on the initial guess (what the key decides)
    kw = dict(max_order=2, approx=True, path_cls=path_cls, path_cls_kwargs=dict(steps=7), key=PRNGKey(1234))
    fused = list(scene.accumulate_over_paths(received_power, **kw))
    traced = list(scene.accumulate_over_paths(host_fun, **kw))
    assert [(a, b) for a, b, _ in fused] == [(a, b) for a, b, _ in traced] and len(fused) == 6
    np.testing.assert_allclose([v for *_, v in fused], [v for *_, v in traced], rtol=2e-6, atol=1e-9)
    other = list(scene.accumulate_over_paths(received_power, **dict(kw, key=PRNGKey(1235))))
    assert not np.allclose([v for *_, v in fused], [v for *_, v in other], rtol=1e-4), "the key must matter after 7 steps"
    # pairs of one transmitter must not share their guesses: the chain gives each its own
    values, vjp = scene.accumulate_over_paths_value_and_vjp(received_power, **kw)
    assert [values[(a, b)] for a, b, _ in fused] == [v for *_, v in fused]
    assert all(np.isfinite(vjp["transmitters"][k]).all() for k in ("tx_a", "tx_b"))
    one = Scene(transmitters=scene.transmitters, receivers=scene.receivers, objects=scene.objects[:1])
    assert list(one.accumulate_over_paths(received_power, **dict(kw, order=2))) == []
    assert list(one.accumulate_over_paths(host_fun, **dict(kw, order=2))) == []
    assert one.accumulate_over_paths(received_power, reduce_all=True, **dict(kw, order=2)) == 0.0


@pytest.mark.parametrize("approx", [False, True])
def test_accumulate_over_paths_value_and_vjp_like_plot_power_optimize(approx):
    """examples/plot_power_optimize.py:60-93, 207-226 transcribed: `loss(tx_coords, scene)` = -min over the receivers of
    (accumulated power / P0), and `jax.value_and_grad(loss)` w.r.t. the transmitter's coordinates -- here one forward launch
    over the receivers (a 1 x R grid), the objective's own derivative on the host, one reverse launch -- against reverse-mode
    autodiff of the oracle (torch, fp64) of the same loss; also the fused values against the traced-paths + host-fun route."""
    import torch

    from differt2d_amd.geometry import Point
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import P0, received_power
    from oracle import ref as R

    scene = Scene.square_scene_with_obstacle()
    scene = scene.with_transmitters(tx=Point(xy=np.array([0.5, 0.7], F)))
    scene = scene.with_receivers(rx_0=Point(xy=np.array([0.3, 0.1], F)), rx_1=Point(xy=np.array([0.5, 0.1], F)),
                                 rx_2=Point(xy=np.array([0.83, 0.41], F)))
    kw = dict(max_order=1, approx=approx, alpha=50.0)  # (max_order = 1: the hard mode's zero-gradient zone is avoided, as the example notes)

    def cot_of(values):  # d loss / d value: -1 / P0 for the receiver with the smallest power (jnp.minimum in a left fold)
        worst = min(values, key=lambda k: float(values[k]))
        return {worst: -1.0 / P0}

    values, vjp = scene.accumulate_over_paths_value_and_vjp(received_power, cotangent=cot_of, **kw)
    plain = {(a, b): v for a, b, v in scene.accumulate_over_paths(received_power, **kw)}
    assert list(values) == list(plain) == [("tx", "rx_0"), ("tx", "rx_1"), ("tx", "rx_2")]
    assert all(values[k] == plain[k] for k in plain)
    host_fun = lambda t, r, p, o: received_power(t, r, p, o)  # noqa: E731 -- not fused: GPU trace + host fun
    host_fun._d2d_native = False
    slow = {(a, b): v for a, b, v in scene.accumulate_over_paths(host_fun, **kw)}
    np.testing.assert_allclose([values[k] for k in plain], [slow[k] for k in plain], rtol=2e-6)
    loss = -min(float(v) for v in values.values()) / P0

    # the oracle's loss under torch autodiff
    tb = R.TorchBackend("float64")
    walls = tb.asarray(_scene_walls(scene)).clone().requires_grad_(True)
    tx = tb.asarray(scene.transmitters["tx"].xy).clone().requires_grad_(True)
    rx = np.stack([r.xy for r in scene.receivers.values()])
    Z = R.power_map(walls, tx, rx[None, :, 0], rx[None, :, 1], min_order=0, max_order=1, approx=approx, alpha=50.0, xp=tb)[0]
    want_loss = -(Z / P0).min()
    g_tx, g_walls = torch.autograd.grad(want_loss, [tx, walls], retain_graph=True)
    assert abs(loss - float(want_loss)) <= 2e-6 * abs(float(want_loss))
    got_tx = vjp["transmitters"]["tx"]
    assert np.isfinite(got_tx).all() and np.abs(g_tx.numpy()).max() > 0
    np.testing.assert_allclose(got_tx, g_tx.numpy(), rtol=2e-5, atol=2e-5 * float(np.abs(g_tx.numpy()).max()))
    np.testing.assert_allclose(vjp["objects"], g_walls.numpy(), rtol=2e-5, atol=2e-5 * float(np.abs(g_walls.numpy()).max()))
    # default cotangent: the gradient of the reduce_all sum
    values1, vjp1 = scene.accumulate_over_paths_value_and_vjp(received_power, **kw)
    g1, = torch.autograd.grad(Z.sum(), [tx])
    np.testing.assert_allclose(vjp1["transmitters"]["tx"], g1.numpy(), rtol=2e-5, atol=2e-5 * float(np.abs(g1.numpy()).max()))
    assert np.isclose(sum(float(v) for v in values1.values()), float(scene.accumulate_over_paths(received_power, reduce_all=True, **kw)), rtol=1e-6)


@pytest.mark.parametrize("native", ["tagged", "recognised", "host"])
def test_accumulate_on_receivers_grid_los(native):
    # tests/test_scene.py:558-627, with the test's own local `fun` ("recognised": the reference's call pattern as written,
    # grad=True included), the tagged library function, and a callable that only the host can evaluate (values only)
    from differt2d_amd.geometry import Point
    from differt2d_amd.scene import Scene, _native_fun
    from differt2d_amd.utils import path_length_squared

    fun = {"tagged": path_length_squared, "recognised": _length_sq, "host": _length_sq_host}[native]
    assert (_native_fun(fun, (), None) is None) == (native == "host")
    native = native != "host"
    scene = Scene(transmitters={"tx0": Point(xy=[0.0, 0.0]), "tx1": Point(xy=[1.0, 0.0])}, objects=[], receivers={})
    x = np.linspace(-3, 3, 10).astype(F)
    X, Y = np.meshgrid(x, x)
    got = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=fun, max_order=1, approx=False, key=1234)
    k0, Z0 = next(got)
    k1, Z1 = next(got)
    assert (k0, k1) == ("tx0", "tx1") and Z0.shape == X.shape and Z0.dtype == np.float32
    np.testing.assert_allclose(Z0, X**2 + Y**2, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(Z1, (X - 1.0) ** 2 + Y**2, rtol=1e-6, atol=1e-6)
    Z = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=fun, reduce_all=True, max_order=1, approx=False)
    expected_Z = X**2 + Y**2 + (X - 1.0) ** 2 + Y**2
    np.testing.assert_allclose(Z, expected_Z, rtol=1e-6, atol=1e-5)
    if not native:
        return
    # gradients, tests/test_scene.py:597-627
    expected_dZ = np.stack([2 * X, 2 * Y], axis=-1) + np.stack([2 * (X - 1.0), 2 * Y], axis=-1)
    dZ = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=fun, reduce_all=True, grad=True, max_order=1, approx=False)
    np.testing.assert_allclose(dZ, expected_dZ, rtol=1e-5, atol=1e-5)
    Z2, dZ2 = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=fun, reduce_all=True, value_and_grad=True,
                                                            max_order=1, approx=False)
    np.testing.assert_allclose(Z2, expected_Z, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(dZ2, expected_dZ, rtol=1e-5, atol=1e-5)
    per_tx = dict(scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=fun, grad=True, max_order=1, approx=False))
    np.testing.assert_allclose(per_tx["tx1"], np.stack([2 * (X - 1.0), 2 * Y], axis=-1), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("native", ["tagged", "recognised", "host"])
def test_accumulate_on_transmitters_grid_los(native):
    # tests/test_scene.py:487-556
    from differt2d_amd.geometry import Point
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import path_length_squared

    fun = {"tagged": path_length_squared, "recognised": _length_sq, "host": _length_sq_host}[native]
    native = native != "host"
    scene = Scene(transmitters={}, objects=[], receivers={"rx0": Point(xy=[0.0, 0.0]), "rx1": Point(xy=[0.0, 1.0])})
    x = np.linspace(-3, 3, 10).astype(F)
    X, Y = np.meshgrid(x, x)
    got = dict(scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=fun, max_order=1, approx=False, key=1234))
    np.testing.assert_allclose(got["rx0"], X**2 + Y**2, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got["rx1"], X**2 + (Y - 1.0) ** 2, rtol=1e-6, atol=1e-6)
    expected_Z = X**2 + Y**2 + X**2 + (Y - 1.0) ** 2
    Z = scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=fun, reduce_all=True, max_order=1, approx=False)
    np.testing.assert_allclose(Z, expected_Z, rtol=1e-6, atol=1e-5)
    if not native:
        return
    expected_dZ = np.stack([2 * X, 2 * Y], axis=-1) + np.stack([2 * X, 2 * (Y - 1.0)], axis=-1)
    dZ = scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=fun, reduce_all=True, grad=True, max_order=1, approx=False)
    np.testing.assert_allclose(dZ, expected_dZ, rtol=1e-5, atol=1e-5)
    Z2, dZ2 = scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=fun, reduce_all=True, value_and_grad=True,
                                                               max_order=1, approx=False)
    np.testing.assert_allclose(Z2, expected_Z, rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(dZ2, expected_dZ, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("approx,function", [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")])
def test_transmitters_grid_with_walls_matches_oracle(approx, function):
    # the benchmark harness of the reference sweeps a TX grid on basic_scene (tests/benchmarks/test_scene.py:9-29)
    from differt2d_amd import logic
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power
    from oracle import c_oracle as CO
    from oracle import ref as R

    scene = Scene.basic_scene()
    X, Y = scene.grid(m=27, n=19)
    X, Y = X * F(0.97) + F(0.013), Y * F(0.97) + F(0.017)
    kw = dict(min_order=0, max_order=2, approx=approx, function=function)
    got = scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=received_power, reduce_all=True, min_order=0, max_order=2,
                                                           approx=approx, function=getattr(logic, function))
    want = CO.power_map(_scene_walls(scene), scene.receivers["rx"].xy, X, Y, grid_role="tx", **kw)
    if function == "sigmoid":
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-5)
    else:
        assert np.array_equal(got, want)
    # per-cell gradient w.r.t. the transmitter vs autodiff of the oracle (order <= 1: basic_scene's collinear walls
    # make the reference's own order-2 gradient NaN everywhere)
    Z, dZ = scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=received_power, reduce_all=True, value_and_grad=True,
                                                             max_order=1, approx=approx, function=getattr(logic, function))
    g = R.power_map_value_and_grads(_scene_walls(scene), scene.receivers["rx"].xy, X, Y, dtype="float64", grid_role="tx",
                                    min_order=0, max_order=1, approx=approx, function=function)
    assert np.array_equal(np.isnan(dZ), np.isnan(g["grad_rx"]))
    scale = np.nanmax(np.abs(g["grad_rx"]))
    assert np.nanmax(np.abs(dZ - g["grad_rx"])) <= 3e-5 * scale


# ---- power maps with walls: Scene API == oracle (the reference has no numeric pin here) --------


@pytest.mark.parametrize("approx", [False, True])
@pytest.mark.parametrize("scene_name", ["square_scene_with_wall", "basic_scene", "square_scene_with_obstacle"])
def test_power_map_examples_match_oracle(scene_name, approx):
    # call pattern of examples/plot_power_map.py:60-67 at a smaller grid
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power
    from oracle import c_oracle as CO

    scene = Scene.from_scene_name(scene_name)
    X, Y = scene.grid(n=40)
    P = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=approx, key=1234)
    want = CO.power_map(_scene_walls(scene), scene.transmitters["tx"].xy, X, Y, min_order=0, max_order=1, approx=approx)
    assert np.array_equal(P, want)
    # same thing through the non-fused route (GPU trace + host fun) agrees to fp32 rounding of the host fun
    Pe = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=lambda *a: received_power(*a), reduce_all=True, approx=approx)
    np.testing.assert_allclose(Pe, want, rtol=1e-6, atol=1e-6)


def test_sweep_kwargs_and_filter_objects():
    from differt2d_amd import logic
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power
    from oracle import c_oracle as CO

    scene = Scene.basic_scene()
    X, Y = scene.grid(m=33, n=21)
    keep = lambda o: float(o.xys[0, 0]) != 0.4  # drop the two walls starting at x = 0.4 from the candidates
    allowed = np.array([1 if keep(o) else 0 for o in scene.objects], np.uint8)
    got = scene.accumulate_on_receivers_grid_over_paths(
        X, Y, fun=received_power, fun_kwargs={"r_coef": 0.7, "height": 0.2}, reduce_all=True, min_order=1, max_order=2,
        filter_objects=keep, approx=True, alpha=50.0, function=logic.hard_sigmoid, tol=0.02, patch=0.01)
    want = CO.power_map(_scene_walls(scene), scene.transmitters["tx"].xy, X, Y, allowed=allowed, min_order=1, max_order=2,
                        approx=True, alpha=50.0, tol=0.02, patch=0.01, r_coef=0.7, height=0.2)
    assert np.array_equal(got, want)
    # approx=None follows the module flag (reference logic.py:333-334)
    with logic.enable_approx(True):
        a = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True)
    b = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=True)
    assert np.array_equal(a, b)


def test_unsupported_is_loud():
    from differt2d_amd import _lib as L
    from differt2d_amd.geometry import MinPath, Path
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power

    class MyPath(Path):  # a user-defined solver cannot run on the GPU
        pass

    scene = Scene.square_scene()
    X, Y = scene.grid(n=4)
    with pytest.raises(L.D2DUnsupported):
        scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, path_cls=MyPath, key=1)
    # gradients through the MinPath Adam loop are supported since ABI v4 (tests/test_gpu_opt.py checks their values)
    gmap = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, path_cls=MinPath, key=1, grad=True,
                                                         reduce_all=True)
    assert gmap.shape == X.shape + (2,)
    with pytest.raises(L.D2DUnsupported):
        scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, path_cls=MinPath, key=1,
                                                      path_cls_kwargs={"optimizer": object()})
    with pytest.raises(L.D2DUnsupported):
        scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, function=lambda x, a: x, approx=True)


def test_scene_vjp_entry_point_matches_autodiff():
    # what examples/plot_power_optimize.py obtains with jax.value_and_grad over tx_coords
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power
    from oracle import ref as R

    # (basic_scene has two collinear walls: reflecting in both gives coincident interaction points and the
    #  reference's autodiff then returns NaN for every cell at order 2 -- not a useful comparison)
    scene = Scene.square_scene_with_wall()
    X, Y = scene.grid(m=18, n=14)
    X, Y = X * F(0.97) + F(0.013), Y * F(0.97) + F(0.017)  # keep cells off the walls (the reference's NaN traps)
    (name, out), = list(scene.receivers_grid_value_and_vjp(X, Y, fun=received_power, max_order=2, approx=True, alpha=50.0))
    want = R.power_map_value_and_grads(_scene_walls(scene), scene.transmitters["tx"].xy, X, Y, dtype="float64",
                                       min_order=0, max_order=2, approx=True, alpha=50.0)
    assert name == "tx"
    for k, w in (("grad_rx", "grad_rx"), ("tx_bar", "tx_bar"), ("objects_bar", "walls_bar")):
        assert not np.isnan(want[w]).any() and not np.isnan(out[k]).any(), k
        scale = np.abs(want[w]).max()
        assert np.abs(out[k] - want[w]).max() <= 3e-5 * scale, k


def test_schedule_diagnostics_roundtrip():
    """d2d_debug_{get,set}_schedule / d2d_debug_get_work: the built-in schedule is a permutation of the patches, an injected
    one (here: reversed) changes nothing but the order, and the work history is positive for every patch."""
    from conftest import random_scene, unit_grid
    from differt2d_amd.engine import Context, make_params

    tx, walls = random_scene(20, seed=2)
    X, Y = unit_grid(80, 64)
    n_patches = 10 * 8
    with Context(0) as ctx:
        ctx.set_scene(walls)
        ctx.set_grid(X, Y)
        ctx.set_option("sched_min_tiles", 1)
        p = make_params(max_order=2, approx=True)
        ctx.launch(p, tx)
        ref = ctx.get_map()
        order, key = ctx.debug_get_schedule(n_patches)
        assert sorted(order.tolist()) == list(range(n_patches)) and key.shape == (n_patches,)
        work = ctx.debug_get_work(n_patches)
        assert (work > 0).all()
        ctx.debug_set_schedule(order[::-1].copy())
        ctx.launch(p, tx)
        assert np.array_equal(ctx.get_map(), ref)
        with pytest.raises(Exception):
            ctx.debug_set_schedule(np.zeros(n_patches, np.int32))  # not a permutation
        ctx.debug_set_schedule(None)
        ctx.launch(p, tx)
        assert np.array_equal(ctx.get_map(), ref)


def test_resident_results_are_invalidated_with_what_they_were_computed_for():
    """include/d2d.h promises D2D_ERR_STATE, not stale or out-of-bounds reads: the per-cell gradient map belongs to the
    grid it was swept on, the scene VJP to the scene; a rejected d2d_set_scene leaves the previous scene in place."""
    from conftest import random_scene, unit_grid
    from differt2d_amd import _lib as L
    from differt2d_amd.engine import Context, make_params

    tx, walls = random_scene(6, seed=2)
    X, Y = unit_grid(16, 8)
    with Context(0) as ctx:
        ctx.set_scene(walls)
        ctx.set_grid(X, Y)
        p = make_params(max_order=2, approx=True)
        ctx.launch_vg(p, tx, scene_vjp=True)
        assert ctx.get_grad_rx().shape == (8, 16, 2)
        X2, Y2 = unit_grid(64, 64)
        ctx.set_grid(X2, Y2)  # 32x the cells: the old gradient buffer must not be read as if it covered them
        with pytest.raises(L.D2DError, match="D2D_ERR_STATE"):
            ctx.get_grad_rx()
        with pytest.raises(L.D2DError, match="D2D_ERR_STATE"):  # accumulate into a gradient map that does not exist
            ctx.launch_vg(make_params(max_order=2, approx=True, out_mode=L.OUT_ADD), tx)
        ctx.launch_vg(p, tx, scene_vjp=True)
        ref_map = ctx.get_map()
        _, wb = ctx.get_scene_vjp()
        assert wb.shape == (6, 2, 2)
        # a rejected scene (unknown kind) changes nothing: same map afterwards
        tx2, walls2 = random_scene(40, seed=3)
        with pytest.raises(L.D2DError, match="D2D_ERR_INVALID"):
            ctx.set_scene(walls2, kind=np.full(40, 7, np.uint8))
        ctx.launch(p, tx)
        assert np.array_equal(ctx.get_map(), ref_map)
        # a new, larger scene: the VJP of the old one (4 * 6 + 2 values) must not be handed out as 4 * 40 + 2
        ctx.set_scene(walls2)
        with pytest.raises(L.D2DError, match="D2D_ERR_STATE"):
            ctx.get_scene_vjp()
        tb, wb, pb = (ctx.launch_vg(p, tx, scene_vjp=True), ctx.get_scene_vjp(with_phi=True))[1]
        assert wb.shape == (40, 2, 2) and pb.shape == (40,) and not pb.any()  # ImagePath sweeps do not depend on phi


def test_trace_paths_checks_the_number_of_theta0_rows():
    from differt2d_amd.engine import Context, make_params
    from oracle import ref as R

    with Context(0) as ctx:
        ctx.set_scene(R.square_scene_walls())
        cands = [np.array([0], np.int32), np.array([1], np.int32), np.array([2], np.int32)]
        p = make_params(order=1, solver="min", steps=10)
        with pytest.raises(ValueError, match="theta0 must have 3 rows"):
            ctx.trace_paths(p, [[0.2, 0.2]], [[0.8, 0.6]], cands, theta0=[[0.5], [0.5]])
        out = ctx.trace_paths(p, [[0.2, 0.2]], [[0.8, 0.6]], cands, theta0=[[0.5], [0.5], [0.5]])
        assert out["xys"].shape[:2] == (1, 3)


# ---- the reference's calling pattern at the resident rate -------------------------------------------


def test_repeated_api_calls_reuse_the_resident_grid_and_the_work_history():
    """A caller of the reference's API hands X, Y and the objects over with every call (scene.py:1803-1826).  The grid and the
    scene that are resident already must be recognised -- immutable arrays by identity, writable ones byte for byte against
    the context's host copy -- so that nothing is uploaded again and the patch schedule keeps its work history (the keys of the schedule are
    then the ones derived from the counted work, not the cold launch's proxy).  A grid that differs in one bit is a new grid."""
    from differt2d_amd.engine import default_context
    from differt2d_amd.geometry import Point
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power
    from conftest import random_scene

    tx, walls = random_scene(30, seed=3)
    scene = Scene.from_walls_array(walls).with_transmitters(tx=Point(xy=tx))
    ctx = default_context()
    ctx.set_option("sched_min_tiles", 1)
    ctx.set_option("split_max_tiles", 0)
    try:
        x = np.linspace(0.0, 1.0, 160).astype(F)
        n_patches = 20 * 20
        for freeze in (True, False):
            X, Y = np.meshgrid(x * F(0.999) if freeze else x, x)
            if freeze:
                X.setflags(write=False)
                Y.setflags(write=False)
            kw = dict(fun=received_power, reduce_all=True, max_order=2)
            r0 = ctx.grid_reuses()
            maps = [scene.accumulate_on_receivers_grid_over_paths(X, Y, **kw) for _ in range(6)]
            assert ctx.grid_reuses() - r0 == 5  # every call after the first found its grid resident
            assert all(np.array_equal(m, maps[0]) for m in maps)
            # the schedule of the last launch was sorted by the work history (three launches old in the pipeline): its keys
            # are the history's keys.  The counted work is deterministic, so the current history gives the same keys.
            work = ctx.debug_get_work(n_patches)
            order, key = ctx.debug_get_schedule(n_patches)
            want = np.clip((8.0 * np.log2((work | 1).astype(np.float32))).astype(np.int64) - 16, 0, 255)
            assert sorted(order.tolist()) == list(range(n_patches))
            # (a patch that is cut in four records the sum of its parts' work, which counts the shared list culling four
            # times: the keys of the cut patches -- 3 in 32 of a launch -- move a little from launch to launch)
            assert (np.abs(key.astype(np.int64) - want) <= 1).mean() >= 0.85 and len(np.unique(key)) > 8
            assert (np.diff(key[order].astype(np.int64)) <= 0).all()  # dearest first
            if not freeze:
                Y2 = Y.copy()
                Y2.view(np.uint32)[77, 5] ^= np.uint32(1)  # one bit: another grid (upload, no reuse)
                r1 = ctx.grid_reuses()
                m2 = scene.accumulate_on_receivers_grid_over_paths(X, Y2, **kw)
                assert ctx.grid_reuses() == r1 and m2.shape == X.shape
                # ... and the same ARRAY written in place is a new grid too: the map follows the new contents
                from oracle import c_oracle as CO

                Y2[60:110, :] += F(0.0031)
                m3 = scene.accumulate_on_receivers_grid_over_paths(X, Y2, **kw)
                assert ctx.grid_reuses() == r1
                assert np.array_equal(m3, CO.power_map(walls, tx, X, Y2, min_order=0, max_order=2), equal_nan=True)
                assert not np.array_equal(m3[60:110], m2[60:110]) and np.array_equal(m3[:60], m2[:60], equal_nan=True)
                with pytest.raises(Exception):
                    ctx.debug_get_work(n_patches + 1)
        # a changed object is a new scene: the history goes (debug_get_work raises until a sweep has run), results follow
        from oracle import c_oracle as CO

        assert np.array_equal(maps[0], CO.power_map(walls, tx, X, Y, min_order=0, max_order=2), equal_nan=True)
        walls2 = walls.copy()
        walls2[:, :, 0] += F(0.01)
        scene2 = Scene.from_walls_array(walls2).with_transmitters(tx=Point(xy=tx))
        a = scene2.accumulate_on_receivers_grid_over_paths(X, Y, **kw)
        assert np.array_equal(a, CO.power_map(walls2, tx, X, Y, min_order=0, max_order=2), equal_nan=True)
        assert not np.array_equal(a, maps[0])
        # the candidate mask is reset by every call that has no filter (reference: filter_objects is per call)
        b = scene2.accumulate_on_receivers_grid_over_paths(X, Y, filter_objects=lambda o: o is not scene2.objects[1], **kw)
        c = scene2.accumulate_on_receivers_grid_over_paths(X, Y, **kw)
        assert np.array_equal(c, a) and not np.array_equal(b, a)
    finally:
        ctx.set_option("sched_min_tiles", 2048)
        ctx.set_option("split_max_tiles", -1)


def test_scene_vjp_accumulation_refuses_mixed_sweep_kinds():
    """D2D_OUT_ADD adds a sweep's scene VJP to the resident one (reduce_all over transmitters).  Adding an ImagePath sweep to a
    MinPath / FermatPath sweep's VJP (or the other way round) would silently drop or overwrite parts of it: D2D_ERR_STATE."""
    from differt2d_amd import _lib as L
    from differt2d_amd.engine import default_context, make_params

    ctx = default_context()
    walls = np.array([[[0, 0], [1, 0]], [[1, 0], [1, 1]], [[1, 1], [0, 1]], [[0, 1], [0, 0]]], F)
    ctx.set_scene(walls)
    x = np.linspace(0.1, 0.9, 16).astype(F)
    X, Y = np.meshgrid(x, x)
    ctx.set_grid(X, Y)
    tx = np.array([0.3, 0.4], F)
    ctx.set_theta0([np.array([0.5, 0, 0, 0], F)] * 4)
    img = dict(min_order=0, max_order=1, approx=True)
    opt = dict(min_order=1, max_order=1, approx=True, solver="min", steps=5)
    ctx.launch_vg(make_params(**img), tx, scene_vjp=True)
    ctx.launch_vg(make_params(out_mode=L.OUT_ADD, **img), tx, scene_vjp=True)  # same kind: fine
    one = ctx.get_scene_vjp()
    with pytest.raises(L.D2DError) as e:
        ctx.launch_vg(make_params(out_mode=L.OUT_ADD, **opt), tx, scene_vjp=True)
    assert e.value.status == -5
    assert np.array_equal(ctx.get_scene_vjp()[1], one[1])  # the refused launch left the resident VJP alone
    ctx.launch_vg(make_params(**opt), tx, scene_vjp=True)
    ctx.launch_vg(make_params(out_mode=L.OUT_ADD, **opt), tx, scene_vjp=True)
    with pytest.raises(L.D2DError) as e:
        ctx.launch_vg(make_params(out_mode=L.OUT_ADD, **img), tx, scene_vjp=True)
    assert e.value.status == -5


def test_user_functions_are_recognised_by_what_they_compute():
    """a local `fun` that is one of the fused closed forms runs fused -- with gradients -- without a rename;
    anything else goes to the host."""
    from differt2d_amd import _lib as L
    from differt2d_amd.scene import Scene, _native_fun
    from differt2d_amd.utils import received_power

    assert _native_fun(lambda t, r, p, o: p.length(), (), None) == ("length", {})
    assert _native_fun(lambda t, r, p, o: F(1.0), (), None) == ("one", {})
    assert _native_fun(lambda t, r, p, o: received_power(t, r, p, o, r_coef=0.3, height=0.2), (), None) == (
        "received_power", {"r_coef": 0.3, "height": 0.2})
    assert _native_fun(lambda t, r, p, o, h: received_power(t, r, p, o, height=h), (0.5,), None) == ("received_power", {"r_coef": 0.5, "height": 0.5})
    assert _native_fun(lambda t, r, p, o: np.sum((r.xy - t.xy) ** 2), (), None) is None  # = length ** 2 in line of sight only
    assert _native_fun(lambda t, r, p, o: p.length() ** 3, (), None) is None
    assert _native_fun(lambda t, r, p, o: p.nope(), (), None) is None  # raises on the probes: the host's business
    scene = Scene.square_scene_with_obstacle()
    X, Y = scene.grid(n=40)
    mine = lambda transmitter, receiver, path, interacting_objects: 0.25 ** (path.xys.shape[-2] - 2) / (0.04 + path.length() ** 2)  # noqa: E731
    kw = dict(reduce_all=True, max_order=2, approx=True)
    Z, dZ = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=mine, value_and_grad=True, **kw)
    Zr, dZr = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, fun_kwargs=dict(r_coef=0.25, height=0.2),
                                                            value_and_grad=True, **kw)
    assert np.array_equal(Z, Zr) and np.array_equal(dZ, dZr, equal_nan=True) and np.isfinite(dZ).mean() > 0.9
    # anything else: the host evaluates it, and its gradient goes through fun.value_and_grad or a tape of its operations
    # (fun_grad.py; the tests below) -- a function the tape cannot follow (numpy calls on recording tensors) is refused
    g = scene.accumulate_on_receivers_grid_over_paths(X[:8, :8], Y[:8, :8], fun=_length_sq_host, grad=True, **kw)
    g0 = scene.accumulate_on_receivers_grid_over_paths(X[:8, :8], Y[:8, :8], fun=_length_sq, grad=True, **kw)
    # (row 0 and column 0 of this grid lie ON the square's walls: NaN cells, by the reference's rules, in both)
    assert g.shape == (8, 8, 2) and np.array_equal(np.isnan(g), np.isnan(g0)) and np.isfinite(g).mean() > 0.7
    assert np.nanmax(np.abs(g - g0)) <= 3e-3 * np.nanmax(np.abs(g0))  # (the extra factor is 1 + 1e-3 k, k <= 2)
    with pytest.raises(L.D2DUnsupported):
        scene.accumulate_on_receivers_grid_over_paths(X[:8, :8], Y[:8, :8], fun=lambda t, r, p, o: np.sort(p.length()) * len(o),
                                                      grad=True, **kw)


# ---- gradients of sweeps whose path function is NOT fused natively (reference scene.py:1892-1923 with any callable) ----
def _power_like(transmitter, receiver, path, interacting_objects, r_coef=0.5, height=0.1):
    """received_power written with operators only: works on arrays and on recording tensors alike."""
    r = path.length()
    n = path.xys.shape[-2] - 2
    return (r_coef ** n) / (height * height + r * r)


_power_like._d2d_native = False  # (kept away from the recogniser: this test is about the host-evaluated route)


def _odd_fun(transmitter, receiver, path, interacting_objects, w=0.3):
    """Depends on the length, on both end points as arguments and on an interior point of the path."""
    r = path.length()
    dx = receiver.xy[..., 0] - transmitter.xy[..., 0]
    return w * r * r.sqrt() + dx * dx + path.xys[..., -2, 0] * receiver.xy[..., 1]


_odd_fun._d2d_native = False


def _odd_fun_oracle(pts, xp=None, w=0.3):
    from oracle import ref as R

    r = R.path_length(pts, xp)
    dx = pts[-1][..., 0] - pts[0][..., 0]
    return w * r * r.sqrt() + dx * dx + pts[-2][..., 0] * pts[-1][..., 1]


@pytest.mark.parametrize("role", ["rx", "tx"])
@pytest.mark.parametrize("approx", [False, True])
def test_gradient_of_a_host_evaluated_fun_equals_the_fused_one(approx, role):
    """A callable that computes what received_power computes, but on the host-evaluated route (paths traced on the GPU, fun
    and d fun / d xys from a tape of its operations, chained through the kernels' adjoint): the per-cell gradient, the NaN
    cells and the values of the natively fused sweep."""
    from conftest import random_scene, unit_grid
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power

    tx, walls = random_scene(7, seed=77)
    scene = Scene.from_walls_array(walls)
    X, Y = unit_grid(21, 13)
    kw = dict(min_order=0, max_order=2, approx=approx, reduce_all=True, value_and_grad=True)
    from differt2d_amd.geometry import Point

    if role == "rx":
        scene = scene.with_transmitters(tx=Point(xy=tx))
        sweep = scene.accumulate_on_receivers_grid_over_paths
    else:
        scene = scene.with_receivers(rx=Point(xy=tx))
        sweep = scene.accumulate_on_transmitters_grid_over_paths
    Z0, G0 = sweep(X, Y, fun=received_power, fun_kwargs=dict(r_coef=0.5, height=0.1), **kw)
    Z1, G1 = sweep(X, Y, fun=_power_like, **kw)
    np.testing.assert_allclose(Z1, Z0, rtol=2e-6, atol=1e-6)  # (r_coef ** n / (h h + r r): the tape rounds like the kernel up to the power)
    assert np.array_equal(np.isnan(G1), np.isnan(G0))
    scale = float(np.nanmax(np.abs(G0)))
    assert scale > 0
    assert np.nanmax(np.abs(G1 - G0)) <= 1e-5 * scale
    # grad alone, per fixed point
    (name, G2), = list(sweep(X, Y, fun=_power_like, min_order=0, max_order=2, approx=approx, grad=True))
    assert np.array_equal(G2, G1, equal_nan=True)


@pytest.mark.parametrize("approx,function", [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")])
def test_gradient_of_an_arbitrary_fun_against_autodiff_of_the_oracle(approx, function):
    """A function nothing in the library knows (length^1.5, both end points as arguments, an interior path point) through
    Scene.accumulate_on_receivers_grid_over_paths(value_and_grad=True), against reverse-mode autodiff of the oracle with the
    same function; then the same through a user-supplied derivative (fun.value_and_grad)."""
    from conftest import random_scene, unit_grid
    from differt2d_amd import logic
    from differt2d_amd.geometry import Point
    from differt2d_amd.scene import Scene
    from oracle import ref as R

    tx, walls = random_scene(6, seed=5)
    scene = Scene.from_walls_array(walls).with_transmitters(tx=Point(xy=tx))
    X, Y = unit_grid(17, 11)
    Z, G = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=_odd_fun, fun_kwargs=dict(w=0.25), reduce_all=True,
                                                         value_and_grad=True, min_order=0, max_order=2, approx=approx,
                                                         function=getattr(logic, function))
    want = R.power_map_value_and_grads(walls, tx, X, Y, dtype="float64", min_order=0, max_order=2, approx=approx,
                                       function=function, fun=_odd_fun_oracle, fun_kwargs=dict(w=0.25))
    want32 = R.power_map_value_and_grads(walls, tx, X, Y, dtype="float32", min_order=0, max_order=2, approx=approx,
                                         function=function, fun=_odd_fun_oracle, fun_kwargs=dict(w=0.25))
    np.testing.assert_allclose(Z, want["value"], rtol=3e-5, atol=1e-5)
    assert np.array_equal(np.isnan(G), np.isnan(want32["grad_rx"]))
    scale = float(np.nanmax(np.abs(want["grad_rx"])))
    err = np.abs(G - want["grad_rx"])
    bar = np.maximum(1e-5 * scale, 2.0 * np.abs(want32["grad_rx"] - want["grad_rx"]))  # (sigmoid at alpha = 100: the reference's own fp32 distance)
    assert np.nanmax(err - bar) <= 0.0, f"max err {np.nanmax(err):.3e} at scale {scale:.3e}"

    # the same function with its derivative supplied by the user (no tape)
    def supplied(transmitter, receiver, path, interacting_objects, w=0.3):
        xys = np.asarray(path.xys, np.float64)
        v = (xys[..., 1:, :] - xys[..., :-1, :]) + float(np.finfo(np.float32).eps)
        ln = np.sqrt((v * v).sum(-1))
        r = ln.sum(-1)
        txy, rxy = np.asarray(transmitter.xy, np.float64), np.asarray(receiver.xy, np.float64)
        dx = rxy[..., 0] - txy[..., 0]
        val = w * r * np.sqrt(r) + dx * dx + xys[..., -2, 0] * rxy[..., 1]
        dr = 1.5 * w * np.sqrt(r)
        bar = np.zeros_like(xys)
        u = dr[..., None, None] * v / ln[..., None]
        bar[..., 1:, :] += u
        bar[..., :-1, :] -= u
        bar[..., -2, 0] += rxy[..., 1]
        d_tx = np.stack([-2 * dx, np.zeros_like(dx)], -1)
        d_rx = np.stack([2 * dx, xys[..., -2, 0]], -1)
        return val, bar, d_tx, d_rx

    def fun2(*a, **k):  # never called for its value on this route
        raise AssertionError("fun.value_and_grad should have been used")

    fun2.value_and_grad = supplied
    fun2._d2d_native = False
    Z2, G2 = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=fun2, fun_kwargs=dict(w=0.25), reduce_all=True,
                                                           value_and_grad=True, min_order=0, max_order=2, approx=approx,
                                                           function=getattr(logic, function))
    np.testing.assert_allclose(Z2, Z, rtol=2e-6, atol=1e-6)
    assert np.array_equal(np.isnan(G2), np.isnan(G))
    assert np.nanmax(np.abs(G2 - G)) <= 1e-5 * scale


def test_a_custom_fun_that_is_not_finite_for_an_invalid_candidate_is_not_skipped():
    """the reference adds valid * fun for EVERY candidate, so a path function that is inf for one of them makes the
    cell NaN (0 * inf) whether that candidate is valid or not; the custom-function kernel used to return early when valid == 0
    in every lane of the wave and dropped the NaN that the plain sweep (GPU trace + host fun) keeps."""
    from differt2d_amd.geometry import Point
    from differt2d_amd.scene import Scene

    scene = Scene.square_scene_with_obstacle().with_transmitters(tx=Point(xy=np.array([0.2, 0.2], F)))
    X, Y = scene.grid(9, 7)

    def blows_up(transmitter, receiver, path, interacting_objects):  # (never called for its value: value_and_grad below)
        raise AssertionError("fun.value_and_grad should have been used")

    def vg(transmitter, receiver, path, interacting_objects):
        xys = np.asarray(path.xys, np.float64)
        v = (xys[..., 1:, :] - xys[..., :-1, :]) + float(np.finfo(np.float32).eps)
        ln = np.sqrt((v * v).sum(-1))
        bar = np.zeros_like(xys)
        u = v / ln[..., None]
        bar[..., 1:, :] += u
        bar[..., :-1, :] -= u
        val = ln.sum(-1)
        if any(o is scene.objects[0] for o in interacting_objects):
            val = np.full_like(val, np.inf)
        return val, bar

    blows_up.value_and_grad = vg
    blows_up._d2d_native = False
    Z, G = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=blows_up, reduce_all=True, value_and_grad=True, max_order=1, approx=False)
    # object 0 is the bottom wall: for most cells its reflection is a valid path (inf), for the others valid = 0 and 0 * inf = NaN
    assert not np.isfinite(Z).any(), "a candidate with fun = inf makes every cell inf or NaN, as in the reference's sum"
    assert np.isnan(Z).any() and np.isinf(Z).any()


def test_custom_fun_values_are_checked_by_the_library():
    """D2D_FUN_CUSTOM without rows, with the wrong number of candidates, on a forward launch: loud errors (include/d2d.h)."""
    from conftest import random_scene, unit_grid
    from differt2d_amd import _lib as L
    from differt2d_amd.engine import default_context, make_params

    ctx = default_context(0)
    tx, walls = random_scene(4, seed=3)
    X, Y = unit_grid(8, 8)
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    p = make_params(fun="custom", min_order=0, max_order=1)
    with pytest.raises(L.D2DError):
        ctx.launch_vg(p, tx)  # nothing set
    ctx.set_path_fun_values(np.zeros((3, 8, 8), F), np.zeros((3, 8, 8, L.D2D_MAX_ORDER + 2, 2), F))
    with pytest.raises(L.D2DError):
        ctx.launch_vg(p, tx)  # 1 + 4 candidates expected
    with pytest.raises(L.D2DError):
        ctx.launch(p, tx)  # forward launches have no use for it
    f = np.ones((5, 8, 8), F)
    ctx.set_path_fun_values(f, np.zeros((5, 8, 8, L.D2D_MAX_ORDER + 2, 2), F))
    ctx.launch_vg(p, tx)
    a = ctx.get_map()
    ctx.launch(make_params(fun="one", min_order=0, max_order=1), tx)
    assert np.array_equal(a, ctx.get_map())  # f = 1: the count of valid paths
    ctx.set_grid(X, Y + F(0.01))  # a new grid drops the rows
    with pytest.raises(L.D2DError):
        ctx.launch_vg(p, tx)
