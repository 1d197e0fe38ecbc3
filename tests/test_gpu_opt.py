"""
GPU tests of the optimiser-based solvers (MinPath, FermatPath: Adam on the parametric coordinates with a
hand-derived gradient) for Wall / RIS / Vertex scenes -- BASELINE.json configs[4], SURVEY.md section 8 row a-13.

There is no bit-exact bar here: 100-1000 sequential fp32 Adam steps amplify rounding differences between the
kernel's hand-derived gradient and the oracle's autodiff gradient, and the reference itself reports visible
non-convergence noise (papers/joss/paper.md:135).  Bars: the reference's own tolerances for its known answers
(rtol 1e-2 on the reflection point, loss <= 1e-4), and rtol 2e-3 / atol 2e-3 * max on maps against the oracle
run with the same explicit theta0.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32


def test_single_reflection_known_answers():
    # reference tests/test_geometry.py:503-525 (steps fixture = 1000)
    from differt2d_amd.geometry import FermatPath, MinPath, Point, Wall

    wall = Wall(xys=[[0.0, 0.0], [2.0, 0.0]])
    tx, rx = Point(xy=[0.0, 1.0]), Point(xy=[2.0, 1.0])
    expected = np.array([[0.0, 1.0], [1.0, 0.0], [2.0, 1.0]], F)
    for cls in (FermatPath, MinPath):
        got = cls.from_tx_objects_rx(tx, [wall], rx, steps=1000, key=1234)
        assert got.xys.shape == (3, 2)
        np.testing.assert_allclose(got.xys, expected, rtol=1e-2, atol=1e-2)
        if cls is MinPath:
            assert abs(float(got.loss)) <= 1e-4
    for cls in (FermatPath, MinPath):  # no object: straight line, tests/test_geometry.py:391-400
        path = cls.from_tx_objects_rx(tx, [], rx, key=1234)
        np.testing.assert_allclose(path.length(), 2.0, rtol=1e-6)


@pytest.mark.parametrize("approx", [True, False])
def test_is_valid_square_scene_all_path_classes(approx):
    # reference tests/test_geometry.py:451-467 (FermatPath / MinPath columns)
    from differt2d_amd import logic
    from differt2d_amd.geometry import FermatPath, MinPath
    from differt2d_amd.scene import Scene

    scene = Scene.square_scene()
    cand = np.arange(4, dtype=np.int32)
    with logic.enable_approx(approx):
        for cls in (FermatPath, MinPath):
            p = cls.from_tx_objects_rx(scene.transmitters["tx"], scene.objects, scene.receivers["rx"], key=1234)
            assert logic.is_true(p.is_valid(scene.objects, cand, scene.get_interacting_objects(cand)))


def _oracle_objs(scene):
    from differt2d_amd.geometry import RIS, Vertex
    from oracle import ref as R

    out = []
    for o in scene.objects:
        if isinstance(o, Vertex):
            out.append(R.Obj(R.VERTEX, o.xy))
        elif isinstance(o, RIS):
            out.append(R.Obj(R.RIS, o.xys, float(o.phi)))
        else:
            out.append(R.Obj(R.WALL, o.xys))
    return out


def _ris_scene():
    # examples/plot_ris_power_map.py:38-43 + the RIS end points as diffraction vertices (BASELINE.json configs[4])
    from differt2d_amd.geometry import RIS
    from differt2d_amd.scene import Scene

    scene = Scene.square_scene()
    ris = RIS(xys=[[0.5, 0.3], [0.5, 0.7]], phi=np.pi / 4)
    return scene.add_objects(ris, *ris.get_vertices())


@pytest.mark.parametrize("path_cls_name,steps", [("MinPath", 100), ("FermatPath", 100), ("MinPath", 400)])
@pytest.mark.parametrize("approx", [False, True])
def test_ris_vertex_sweep_matches_oracle(path_cls_name, steps, approx):
    import differt2d_amd.geometry as G
    from differt2d_amd.utils import received_power
    from oracle import ref as R

    scene = _ris_scene()
    path_cls = getattr(G, path_cls_name)
    X, Y = scene.grid(m=12, n=10)
    X, Y = X * F(0.96) + F(0.021), Y * F(0.96) + F(0.017)
    cands = scene.all_path_candidates(order=1)
    rng = np.random.default_rng(3)
    theta0 = [rng.random(sum(o.parameters_count() for o in scene.get_interacting_objects(c)), dtype=F) for c in cands]
    got = scene.accumulate_on_receivers_grid_over_paths(
        X, Y, fun=received_power, path_cls=path_cls, order=1, reduce_all=True, approx=approx,
        path_cls_kwargs={"steps": steps, "theta0": theta0}, key=1234)
    okw = dict(order=1, objs=_oracle_objs(scene), approx=approx, theta0s=theta0, steps=steps,
               solver={"MinPath": "min", "FermatPath": "fermat"}[path_cls_name])
    want = R.power_map(None, scene.transmitters["tx"].xy, X, Y, **okw)
    want64 = R.power_map(None, scene.transmitters["tx"].xy, X, Y, xp=R.NUMPY64, **okw)
    assert got.shape == X.shape
    scale = np.abs(want).max()
    # Hundreds of sequential fp32 Adam steps are not reproducible to the last bit across two gradient
    # implementations, and ill-conditioned cells (RX next to a wall) amplify that: the oracle run in fp64 differs
    # from the oracle run in fp32 in those very cells.  Bar: on the cells where the oracle is stable (fp32 ~ fp64)
    # the kernel must agree with it (>= 90 %; typically 95-100 %), and overall it must not be more unstable than the oracle itself.
    stable = np.isclose(want, want64, rtol=2e-3, atol=2e-3 * scale)
    close = np.isclose(got, want, rtol=2e-3, atol=2e-3 * scale)
    assert stable.mean() >= 0.7
    assert close[stable].mean() >= 0.90, f"{(~close[stable]).sum()} of {stable.sum()} stable cells differ"
    assert (~close).sum() <= 2 * (~stable).sum() + 0.05 * close.size
    assert np.median(np.abs(got - want)) <= 1e-4 * scale


def test_all_paths_with_minpath_and_key():
    from differt2d_amd.geometry import MinPath

    scene = _ris_scene()
    out = list(scene.all_paths(path_cls=MinPath, path_cls_kwargs={"steps": 200}, order=1, key=1234, approx=False))
    assert len(out) == 7 and all(p.xys.shape == (3, 2) for *_, p, _ in out)
    # vertex candidates: the path goes through the vertex exactly, loss 0 (reference geometry.py:381-385, 416-419)
    for _, _, _, path, cand in out[-2:]:
        assert np.array_equal(path.xys[1], scene.objects[int(cand[0])].xy) and float(path.loss) == 0.0
    again = list(scene.all_paths(path_cls=MinPath, path_cls_kwargs={"steps": 200}, order=1, key=1234, approx=False))
    assert all(np.array_equal(a[3].xys, b[3].xys) for a, b in zip(out, again))  # same key, same draw


def test_vertex_diffraction_call_pattern():
    # examples/plot_vertex_diffraction_power_map.py:81-90: FermatPath through Vertex objects only
    from differt2d_amd.geometry import FermatPath, Vertex
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power

    scene = Scene.basic_scene()
    wall = scene.objects[-2]
    _, vertex = wall.get_vertices()
    scene = scene.add_objects(vertex).filter_objects(lambda o: o is not wall)
    X, Y = scene.grid(n=24)
    P = scene.accumulate_on_receivers_grid_over_paths(
        X, Y, fun=received_power, order=1, filter_objects=lambda o: isinstance(o, Vertex), path_cls=FermatPath,
        reduce_all=True, key=1234)
    assert P.shape == X.shape and np.isfinite(P).all() and (P > 0).any() and (P == 0).any()


def test_missing_key_is_an_error():
    from differt2d_amd.geometry import MinPath
    from differt2d_amd.utils import received_power

    scene = _ris_scene()
    X, Y = scene.grid(n=4)
    with pytest.raises(TypeError):
        scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, path_cls=MinPath, order=1, reduce_all=True)


def test_many_random_starts_pick_the_best():
    """reference optimize.py:136-182: the start with the smallest recorded loss wins (first one on ties)."""
    from differt2d_amd.geometry import MinPath, Point, Wall
    from differt2d_amd.utils import received_power

    wall = Wall(xys=[[0.0, 0.0], [2.0, 0.0]])
    tx, rx = Point(xy=[0.0, 1.0]), Point(xy=[2.0, 1.0])
    starts = [[0.9], [0.05], [0.5]]
    singles = [MinPath.from_tx_objects_rx(tx, [wall], rx, steps=30, theta0=s) for s in starts]
    best = min(range(3), key=lambda i: float(singles[i].loss))
    multi = MinPath.from_tx_objects_rx(tx, [wall], rx, steps=30, many=3, theta0=starts)
    assert np.array_equal(multi.xys, singles[best].xys) and multi.loss == singles[best].loss
    # and through a sweep: many=3 with three identical starts equals many=1
    scene = _ris_scene()
    X, Y = scene.grid(m=8, n=6)
    cands = scene.all_path_candidates(order=1)
    rng = np.random.default_rng(0)
    th1 = [rng.random(sum(o.parameters_count() for o in scene.get_interacting_objects(c)), dtype=F) for c in cands]
    th3 = [t for t in th1 for _ in range(3)]
    a = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, path_cls=MinPath, order=1, reduce_all=True,
                                                      path_cls_kwargs={"steps": 50, "theta0": th1})
    b = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, path_cls=MinPath, order=1, reduce_all=True,
                                                      path_cls_kwargs={"steps": 50, "many": 3, "theta0": th3})
    assert np.array_equal(a, b)


@pytest.mark.parametrize("path_cls_name", ["MinPath", "FermatPath"])
def test_candidates_side_by_side_equal_one_after_the_other(path_cls_name):
    """Optimiser-based sweeps run their candidates side by side (one (cell, candidate) per lane, contributions added in
    candidate order afterwards): bit-identical to the one-lane-per-cell kernel that walks the candidates in order."""
    import differt2d_amd.geometry as G
    from differt2d_amd.engine import default_context
    from differt2d_amd.utils import received_power

    scene = _ris_scene()
    X, Y = scene.grid(m=40, n=33)
    cands = scene.all_path_candidates(min_order=0, max_order=1)
    rng = np.random.default_rng(5)
    theta0 = [rng.random(sum(o.parameters_count() for o in scene.get_interacting_objects(c)), dtype=F) for c in cands]
    kw = dict(fun=received_power, path_cls=getattr(G, path_cls_name), min_order=0, max_order=1, reduce_all=True, approx=True,
              path_cls_kwargs={"steps": 120, "theta0": theta0}, key=1)
    ctx = default_context()
    a = scene.accumulate_on_receivers_grid_over_paths(X, Y, **kw)
    ctx.set_option("opt_parallel", 0)
    try:
        b = scene.accumulate_on_receivers_grid_over_paths(X, Y, **kw)
    finally:
        ctx.set_option("opt_parallel", 1)
    assert np.array_equal(a, b, equal_nan=True) and np.isfinite(a).all() and (a > 0).any()
