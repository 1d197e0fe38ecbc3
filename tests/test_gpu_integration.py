"""The reference-side binding shipped in integration/ (the files INTEGRATION.md tells a DiffeRT2d maintainer to add) is
executed here: integration/_d2d.py is imported AS IS against the built libd2d.so, and integration/scene_hook.py's early
return is run on this repository's Scene mirror (same attributes as the reference's Scene: .objects[i].xys,
.transmitters / .receivers dicts of points with .xy) with xp = numpy.  Results must equal the engine's own."""

import importlib.util
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


@pytest.fixture(scope="module")
def d2d():
    from differt2d_amd import _lib as L

    os.environ["DIFFERT2D_LIBD2D"] = L.LIB_PATH
    spec = importlib.util.spec_from_file_location("integration_d2d", os.path.join(ROOT, "integration", "_d2d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def hook():
    sys.path.insert(0, ROOT)
    from integration import scene_hook

    return scene_hook


def test_binding_struct_matches_the_header(d2d):
    from differt2d_amd import _lib as L

    assert [f[0] for f in d2d.Params._fields_] == [f[0] for f in L.Params._fields_]
    import ctypes as C

    assert C.sizeof(d2d.Params) == C.sizeof(L.Params)
    assert d2d.D2D_ABI_VERSION == L.D2D_ABI_VERSION


def _names():
    from differt2d_amd import logic
    from differt2d_amd.geometry import ImagePath, Wall
    from differt2d_amd.utils import received_power

    return dict(hard_sigmoid=logic.hard_sigmoid, sigmoid=logic.sigmoid, received_power=received_power, ImagePath=ImagePath,
                Wall=Wall, enable_approx=False)


def _call(hook, d2d, scene, X, Y, **over):
    from differt2d_amd.geometry import ImagePath
    from differt2d_amd.utils import received_power

    args = dict(fun=received_power, fun_args=(), fun_kwargs=None, reduce_all=False, grad=False, value_and_grad=False,
                path_cls=ImagePath, min_order=0, max_order=1, order=None, filter_objects=None, kwargs={}, xp=np,
                names=_names(), binding=d2d)
    args.update(over)
    return hook.mi355x_sweep(scene, X, Y, args.pop("fun"), args.pop("fun_args"), args.pop("fun_kwargs"), **args)


def test_hook_equals_the_engine_per_transmitter_and_reduced(d2d, hook):
    from differt2d_amd.geometry import Point
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power

    scene = Scene.square_scene_with_obstacle().with_transmitters(tx0=Point(xy=np.array([0.2, 0.2], F)),
                                                                 tx1=Point(xy=np.array([0.7, 0.35], F)))
    X, Y = scene.grid(n=96)
    for kwargs in (dict(), dict(approx=True), dict(approx=True, alpha=50.0, tol=0.05, patch=0.01)):
        common = dict(min_order=0, max_order=2)
        got = dict(_call(hook, d2d, scene, X, Y, kwargs=kwargs, **common))
        want = dict(scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, **common, **kwargs))
        assert got.keys() == want.keys() and len(got) >= 1
        for k in want:
            assert np.array_equal(got[k], want[k], equal_nan=True)
        red = _call(hook, d2d, scene, X, Y, kwargs=kwargs, reduce_all=True, **common)
        assert np.array_equal(red, scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True,
                                                                                 **common, **kwargs), equal_nan=True)
        v, g = _call(hook, d2d, scene, X, Y, kwargs=kwargs, reduce_all=True, value_and_grad=True, **common)
        wv, wg = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True, value_and_grad=True,
                                                               **common, **kwargs)
        assert np.array_equal(v, wv, equal_nan=True) and np.array_equal(g, wg, equal_nan=True)
        gonly = _call(hook, d2d, scene, X, Y, kwargs=kwargs, reduce_all=True, grad=True, **common)
        assert np.array_equal(gonly, wg, equal_nan=True)


def test_hook_equals_the_engine_context_on_the_benchmark_scene(d2d, hook):
    """Same map as engine.Context on (a 256^2 version of) BASELINE.json configs[1], through the reference-side binding."""
    from conftest import random_scene, unit_grid
    from differt2d_amd.engine import Context
    from differt2d_amd.scene import Scene

    tx, walls = random_scene(50, seed=1234)
    X, Y = unit_grid(256)
    from differt2d_amd.geometry import Point

    scene = Scene.from_walls_array(walls).with_transmitters(tx=Point(xy=tx))
    got = _call(hook, d2d, scene, X, Y, reduce_all=True, max_order=2, filter_objects=lambda o: True)
    with Context(0) as c:
        c.set_scene(walls)
        want = c.power_map(tx, X, Y, min_order=0, max_order=2)
    assert np.array_equal(got, want)


def test_hook_declines_what_the_library_does_not_fuse(d2d, hook):
    from differt2d_amd.geometry import MinPath
    from differt2d_amd.scene import Scene

    scene = Scene.square_scene()
    X, Y = scene.grid(n=8)
    assert _call(hook, d2d, scene, X, Y, fun=lambda *a: 1.0) is hook.NOT_HANDLED          # arbitrary Python fun
    assert _call(hook, d2d, scene, X, Y, path_cls=MinPath) is hook.NOT_HANDLED            # optimiser-based path class
    assert _call(hook, d2d, scene, X, Y, kwargs=dict(function=lambda x: x)) is hook.NOT_HANDLED  # custom activation
