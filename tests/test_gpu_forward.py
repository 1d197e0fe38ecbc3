"""
GPU parity: the HIP forward sweep (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bar: bit-exact in hard and hard_sigmoid modes (identical fp32 op chain, IEEE
divide/sqrt, no contraction); sigmoid mode within rtol 1e-6 and bit-equal in >= 99.9 % of the cells (expf evaluated
differ by an ulp; BASELINE.json's tolerance is 1e-5).
"""

import os

import numpy as np
import pytest

from conftest import random_scene, unit_grid

pytestmark = pytest.mark.gpu

F = np.float32
MODES = [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")]


# Launch shapes: every patch shared between 4 waves (power_fwd_split_kernel) in identity order; the second variant forces what big grids get: one wave per patch, dearest patches first.
@pytest.fixture(scope="module", params=["shared_patches", "one_wave_per_patch_scheduled"])
def ctx(request):
    from differt2d_amd.engine import Context

    with Context(0) as c:
        if request.param == "one_wave_per_patch_scheduled":
            c.set_option("split_max_tiles", 0)
            c.set_option("sched_min_tiles", 1)
        else:  # (the default for grids this small is the candidate-sharing kernel: tests/test_gpu_coop.py)
            c.set_option("coop_waves", 0)
            c.set_option("split_max_tiles", 8192)
            c.set_option("split_sigmoid", 1)
        yield c


def _oracle(walls, tx, X, Y, **kw):
    from oracle import c_oracle as CO

    return CO.power_map(walls, tx, X, Y, prune=True, **kw)


def _compare(got, want, function):
    assert got.shape == want.shape and got.dtype == np.float32
    if function == "sigmoid":
        # The device evaluates expf exactly as the oracle's C library does (d2d_kernels.hpp: expf_libm), so sigmoid maps are
        # bit-comparable too -- up to two effects that touch a cell in a million: the oracle's libm may pick an FMA build of
        # the same polynomial (the double result then differs by an ulp before the rounding to float), and the kernel takes
        # sigmoid(min z) where the reference takes min(sigmoid z) (equal wherever the fp32 sigmoid is monotone).  Bar: 1e-6
        # (north_star: 1e-5) everywhere, and bit equality in all but 0.1 % of the cells.
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9)
        same = (got == want) | (np.isnan(got) & np.isnan(want))
        assert same.mean() >= 0.999, f"{int((~same).sum())} of {same.size} cells differ in some bit"
    else:
        bad = ~((got == want) | (np.isnan(got) & np.isnan(want)))
        assert not bad.any(), f"{bad.sum()} of {bad.size} cells differ; max abs {np.nanmax(np.abs(got - want))}"


@pytest.mark.parametrize("approx,function", MODES)
def test_cfg1_square_scene_64(ctx, approx, function):
    """BASELINE.json configs[0]: Scene.square_scene(), 1 TX, 64x64 RX grid, order <= 1."""
    from oracle import ref as R

    walls, tx = R.square_scene_walls(), np.array([0.2, 0.2], F)
    X, Y = unit_grid(64)
    ctx.set_scene(walls)
    got = ctx.power_map(tx, X, Y, min_order=0, max_order=1, approx=approx, function=function)
    _compare(got, _oracle(walls, tx, X, Y, min_order=0, max_order=1, approx=approx, function=function), function)


@pytest.mark.parametrize("approx,function", MODES)
@pytest.mark.parametrize("fun", ["received_power", "one", "length_squared", "length"])
def test_random_scene_order2(ctx, approx, function, fun):
    tx, walls = random_scene(12, seed=11)
    X, Y = unit_grid(37, 29)  # ragged: not a multiple of the 8x8 wave tile
    ctx.set_scene(walls)
    kw = dict(min_order=0, max_order=2, approx=approx, function=function, fun=fun)
    _compare(ctx.power_map(tx, X, Y, **kw), _oracle(walls, tx, X, Y, **kw), function)


@pytest.mark.parametrize("approx,function", MODES)
def test_order3_and_order4(ctx, approx, function):
    tx, walls = random_scene(6, seed=5)
    X, Y = unit_grid(16, 9)
    ctx.set_scene(walls)
    for lo, hi in [(3, 3), (0, 3), (4, 4)]:
        kw = dict(min_order=lo, max_order=hi, approx=approx, function=function)
        _compare(ctx.power_map(tx, X, Y, **kw), _oracle(walls, tx, X, Y, **kw), function)


@pytest.mark.parametrize("approx,function", MODES)
def test_kwargs_alpha_patch_tol_filter(ctx, approx, function):
    tx, walls = random_scene(8, seed=21)
    X, Y = unit_grid(24)
    allowed = np.array([1, 0, 1, 1, 0, 1, 1, 1], np.uint8)
    ctx.set_scene(walls)
    ctx.set_candidate_mask(allowed)
    kw = dict(min_order=1, max_order=2, approx=approx, function=function, alpha=50.0, patch=0.02, tol=0.05,
              r_coef=0.3, height=0.25)
    got = ctx.power_map(tx, X, Y, **kw)
    ctx.set_candidate_mask(None)
    _compare(got, _oracle(walls, tx, X, Y, allowed=allowed, **kw), function)


@pytest.mark.parametrize("approx,function", MODES[:2])
def test_degenerate_rx_on_walls_endpoints_and_tx(ctx, approx, function):
    from oracle import ref as R

    walls, tx = R.square_scene_with_wall_walls(), np.array([0.2, 0.5], F)
    X, Y = np.meshgrid(np.array([0.0, 0.2, 0.5, 1.0], F), np.array([0.0, 0.2, 0.5, 0.8, 1.0], F))
    ctx.set_scene(walls)
    kw = dict(min_order=0, max_order=2, approx=approx, function=function)
    _compare(ctx.power_map(tx, X, Y, **kw), _oracle(walls, tx, X, Y, **kw), function)


def test_empty_scene_los_analytic(ctx):
    """tests/test_scene.py:558-627 of the reference: no objects, fun = length**2 -> X^2 + Y^2."""
    x = np.linspace(-3, 3, 10).astype(F)
    X, Y = np.meshgrid(x, x)
    ctx.set_scene(np.zeros((0, 2, 2), F))
    Z0 = ctx.power_map([0.0, 0.0], X, Y, max_order=1, fun="length_squared")
    Z1 = ctx.power_map([1.0, 0.0], X, Y, max_order=1, fun="length_squared")
    np.testing.assert_allclose(Z0, X**2 + Y**2, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(Z1, (X - 1) ** 2 + Y**2, rtol=1e-6, atol=1e-6)


def test_notebook_valid_count_on_gpu(ctx):
    """6 valid order-2 candidates on square_scene_with_obstacle at rx (0.5, 0.6) (notebook cell 6)."""
    from oracle import ref as R

    ctx.set_scene(R.square_scene_with_obstacle_walls())
    got = ctx.power_map([0.2, 0.2], np.array([[0.5]], F), np.array([[0.6]], F), order=2, fun="one")
    assert got.shape == (1, 1) and got[0, 0] == 6.0


def test_out_mode_add_two_transmitters(ctx):
    """reduce_all over transmitters: Z = (0 + p0) + p1 (reference scene.py:1948-1952)."""
    from differt2d_amd import _lib as L
    from differt2d_amd.engine import make_params

    tx, walls = random_scene(7, seed=2)
    X, Y = unit_grid(20)
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    ctx.launch(make_params(max_order=2), tx)
    ctx.launch(make_params(max_order=2, out_mode=L.OUT_ADD), tx[::-1].copy())
    got = ctx.get_map()
    want = _oracle(walls, tx, X, Y, max_order=2) + _oracle(walls, tx[::-1].copy(), X, Y, max_order=2)
    _compare(got, want, "hard_sigmoid")


@pytest.mark.parametrize("approx,function", MODES)
def test_cfg2_subgrid_50_walls_order2(ctx, approx, function):
    """BASELINE.json configs[1] scene (50 random walls, order <= 2) on a 96x96 sub-grid of the
    1024x1024 grid (the oracle needs seconds there); intersection counts are bit-exact."""
    tx, walls = random_scene(50, seed=1234)
    x = np.linspace(0.0, 1.0, 1024).astype(F)[::11][:96]
    X, Y = np.meshgrid(x, x)
    ctx.set_scene(walls)
    kw = dict(min_order=0, max_order=2, approx=approx, function=function)
    _compare(ctx.power_map(tx, X, Y, **kw), _oracle(walls, tx, X, Y, **kw), function)
    if not approx:
        kw["fun"] = "one"
        got = ctx.power_map(tx, X, Y, **kw)
        want = _oracle(walls, tx, X, Y, **kw)
        assert np.array_equal(got, want), f"{(got != want).sum()} cells disagree on the valid-path count"


@pytest.mark.parametrize("approx,function", MODES[:2])
def test_cfg2_full_size_sampled_cells_and_block_invariance(ctx, approx, function):
    """BASELINE.json configs[1] exactly as bench.py runs it (50 walls, 1024 x 1024 cells, orders 0..2): 4096 random cells
    against the oracle, bit for bit; and a block cut out at an offset that is not a multiple of the 8 x 8 patch must
    reproduce the same cells (every cell is independent: what the culling decides per patch cannot matter)."""
    tx, walls = random_scene(50, seed=1234)
    x = np.linspace(0.0, 1.0, 1024).astype(F)
    X, Y = np.meshgrid(x, x)
    ctx.set_scene(walls)
    kw = dict(min_order=0, max_order=2, approx=approx, function=function)
    got = ctx.power_map(tx, X, Y, **kw)
    rng = np.random.default_rng(7)
    ii, jj = rng.integers(0, 1024, 4096), rng.integers(0, 1024, 4096)
    want = _oracle(walls, tx, X[ii, jj], Y[ii, jj], **kw)
    assert np.array_equal(got[ii, jj], want)
    assert (want > 0).sum() > 100  # the sample sees plenty of lit cells
    # repeated sweeps of the same grid (what bench.py times): work-history schedule, dearest patches cut in four
    from differt2d_amd.engine import make_params

    ctx.set_grid(X, Y)
    for _ in range(3):
        ctx.launch(make_params(**kw), tx)
        assert np.array_equal(ctx.get_map(), got)
    block = ctx.power_map(tx, X[403:446, 617:700], Y[403:446, 617:700], **kw)
    assert np.array_equal(block, got[403:446, 617:700])


def test_repeated_sweeps_reschedule_from_the_work_history(ctx):
    """A context that sweeps the same grid again orders its patches by the work each one took last time (whatever the
    transmitter was then): only the order changes, never a bit of the maps -- also for the value+grad sweep, whose
    scene VJP sums per-patch partials."""
    from differt2d_amd.engine import make_params

    tx, walls = random_scene(24, seed=11)
    X, Y = unit_grid(72, 56)
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    ctx.set_option("sched_min_tiles", 1)
    try:
        p = make_params(max_order=2, approx=True)
        txs = [tx, np.array([0.31, 0.62], F), tx, tx]
        maps = []
        for t in txs:
            ctx.launch(p, t)          # no set_grid in between: the 2nd .. 4th launch use the history
            maps.append(ctx.get_map())
        assert np.array_equal(maps[0], maps[2]) and np.array_equal(maps[0], maps[3])
        assert np.array_equal(maps[0], _oracle(walls, tx, X, Y, max_order=2, approx=True))
        assert np.array_equal(maps[1], _oracle(walls, txs[1], X, Y, max_order=2, approx=True))
        vj = []
        for _ in range(3):
            ctx.launch_vg(p, tx, scene_vjp=True)
            vj.append((ctx.get_map(), ctx.get_grad_rx(), *ctx.get_scene_vjp()))
        for a in vj[1:]:
            for u, v in zip(vj[0], a):
                assert np.array_equal(u, v, equal_nan=True)
        ctx.set_option("cost_history", 0)
        ctx.launch(p, tx)
        assert np.array_equal(ctx.get_map(), maps[0])
        # with a history, launches too big to share every patch cut their dearest patches in four parts that leave
        # ordered contribution lists in global memory (power_fwd_kernel): force that path on this small grid
        ctx.set_option("cost_history", 1)
        ctx.set_option("split_max_tiles", 0)
        for heavy in (3, 1000):
            ctx.set_option("heavy_split", heavy)
            for t, want in ((tx, maps[0]), (txs[1], maps[1]), (tx, maps[0])):
                ctx.launch(p, t)
                ctx.launch(p, t)
                assert np.array_equal(ctx.get_map(), want)
        for q in (make_params(max_order=2, approx=False), make_params(min_order=2, max_order=2, approx=True), make_params(min_order=1, max_order=2)):
            ctx.set_option("heavy_split", 0)
            ctx.launch(q, tx)
            ref = ctx.get_map()
            ctx.set_option("heavy_split", 1000)
            ctx.launch(q, tx)
            ctx.launch(q, tx)
            assert np.array_equal(ctx.get_map(), ref)
    finally:
        ctx.set_option("cost_history", 1)
        ctx.set_option("sched_min_tiles", 2048)
        ctx.set_option("heavy_split", -1)
        ctx.set_option("split_max_tiles", 8192)


def test_acceleration_masks_do_not_change_results(ctx):
    """The wall-to-wall masks (built once per scene and mode) and the work-history schedule are accelerators only: maps
    with and without them are bit-identical, across a mode change and a scene change that must rebuild the masks."""
    X, Y = unit_grid(64, 48)
    for seed, n in ((3, 20), (8, 33)):
        tx, walls = random_scene(n, seed=seed)
        ctx.set_scene(walls)
        for kw in (dict(approx=False), dict(approx=True), dict(approx=True, alpha=30.0), dict(approx=False, patch=0.03)):
            kw = dict(min_order=0, max_order=3 if n <= 20 else 2, **kw)
            on = ctx.power_map(tx, X, Y, **kw)
            ctx.set_option("pair_masks", 0)
            try:
                off = ctx.power_map(tx, X, Y, **kw)
            finally:
                ctx.set_option("pair_masks", 1)
            assert np.array_equal(on, off, equal_nan=True)
            assert np.array_equal(on, _oracle(walls, tx, X, Y, **kw))


def test_errors_are_loud(ctx):
    from differt2d_amd import _lib as L
    from differt2d_amd.engine import make_params

    ctx.set_scene(np.zeros((1, 2, 2), F), kind=[L.D2D_RIS])
    ctx.set_grid(*unit_grid(4))
    with pytest.raises(L.D2DUnsupported):
        ctx.launch(make_params(max_order=1), [0.1, 0.1])
    with pytest.raises(L.D2DError):
        ctx.launch(make_params(max_order=7), [0.1, 0.1])
    with pytest.raises(L.D2DError):
        ctx.launch(make_params(approx=True, alpha=0.0), [0.1, 0.1])
    # a scene whose tables do not fit one CU's LDS (include/d2d.h: d2d_set_scene) is refused, not truncated
    _, big = _short_walls(3000, seed=3)
    ctx.set_scene(big)
    with pytest.raises(L.D2DUnsupported):
        ctx.launch(make_params(max_order=1), [0.1, 0.1])


@pytest.mark.parametrize("approx,function", MODES)
def test_geojson_scene_degenerate_walls_and_large_offsets(ctx, approx, function):
    """The reference's example.geojson (28 walls, two of zero length, coordinates ~(4.6, 50.7) with wall lengths
    ~1e-4: fp32 is at its limits) -- every culling / filter margin must still be conservative: bit-exact maps."""
    import os

    from differt2d_amd.scene import Scene

    scene = Scene.from_geojson(open(os.path.join(os.path.dirname(__file__), "golden", "example.geojson")).read())
    walls = np.stack([o.xys for o in scene.objects])
    tx = scene.transmitters["tx"].xy
    X, Y = scene.grid(m=24, n=20)
    ctx.set_scene(walls)
    for lo, hi in [(0, 1), (2, 2)]:
        kw = dict(min_order=lo, max_order=hi, approx=approx, function=function)
        _compare(ctx.power_map(tx, X, Y, **kw), _oracle(walls, tx, X, Y, **kw), function)


@pytest.mark.parametrize("approx,function", MODES[:2])
@pytest.mark.parametrize("scale,offset", [(1e-3, 0.0), (1e3, 0.0), (1.0, 1e3), (1e-2, -50.0)])
def test_scaled_and_shifted_scenes(ctx, approx, function, scale, offset):
    """Margins of the culling and of the filters are relative to the magnitudes involved: tiny, huge and far-from-origin
    scenes must stay bit-exact."""
    tx, walls = random_scene(14, seed=23)
    X, Y = unit_grid(33, 25)
    s, o = F(scale), F(offset)
    walls, tx, X, Y = walls * s + o, tx * s + o, X * s + o, Y * s + o
    ctx.set_scene(walls)
    kw = dict(min_order=0, max_order=2, approx=approx, function=function, height=float(0.1 * scale))
    _compare(ctx.power_map(tx, X, Y, **kw), _oracle(walls, tx, X, Y, **kw), function)


@pytest.mark.parametrize("approx,function", MODES[:2])
def test_cfg4_scene_200_walls_small_grid(ctx, approx, function):
    """BASELINE.json configs[3] scene (200 random walls, NumPy seed 1234): orders 0..2 on a 16 x 12 patch of the 2048^2 grid
    (39 801 candidates per cell), and order 3 restricted to 12 candidate walls (filter_objects) so that the oracle finishes."""
    tx, walls = random_scene(200, seed=1234)
    x = np.linspace(0.0, 1.0, 2048).astype(F)
    X, Y = np.meshgrid(x[1000:1016], x[400:412])
    ctx.set_scene(walls)
    kw = dict(min_order=0, max_order=2, approx=approx, function=function)
    _compare(ctx.power_map(tx, X, Y, **kw), _oracle(walls, tx, X, Y, **kw), function)
    allowed = np.zeros(200, np.uint8)
    allowed[::17] = 1
    ctx.set_candidate_mask(allowed)
    kw = dict(min_order=3, max_order=3, approx=approx, function=function)
    got = ctx.power_map(tx, X, Y, **kw)
    ctx.set_candidate_mask(None)
    _compare(got, _oracle(walls, tx, X, Y, allowed=allowed, **kw), function)


def _short_walls(n, seed, half=0.006):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 2))
    ang = rng.random(n) * np.pi
    d = np.stack([np.cos(ang), np.sin(ang)], -1) * half
    return np.array([0.4503, 0.5211], F), np.stack([c - d, c + d], 1).astype(F)


@pytest.mark.parametrize("approx,function", MODES)
def test_scene_of_2000_walls(ctx, approx, function):
    """A scene beyond the 64 KB of LDS the tables of ~1 300 objects fill (VERDICT r4 missing #6: real maps): 2 000 short walls,
    128 KB of tables, one workgroup per CU (gfx950 gives a workgroup the CU's whole 160 KB).  Orders 0..1 over all walls
    (2 001 candidates per cell, 2 000 occluders each), order 2 over 30 candidate walls (filter_objects: 870 candidates, all
    2 000 walls occlude), both grid roles -- bit for bit against the C oracle like every other scene."""
    tx, walls = _short_walls(2000, seed=11)
    X, Y = unit_grid(24, 16)
    X, Y = (X * F(0.3) + F(0.31)).astype(F), (Y * F(0.2) + F(0.42)).astype(F)
    ctx.set_scene(walls)
    kw = dict(min_order=0, max_order=1, approx=approx, function=function)
    got = ctx.power_map(tx, X, Y, **kw)
    _compare(got, _oracle(walls, tx, X, Y, **kw), function)
    assert np.count_nonzero(got) > 100
    allowed = np.zeros(2000, np.uint8)
    allowed[np.argsort(((walls.mean(1) - np.array([0.46, 0.52], F)) ** 2).sum(-1))[:30]] = 1  # the 30 walls nearest to the grid
    ctx.set_candidate_mask(allowed)
    kw2 = dict(min_order=2, max_order=2, approx=approx, function=function)
    got2 = ctx.power_map(tx, X, Y, **kw2)
    from differt2d_amd import _lib as L

    got_tx = ctx.power_map(tx, X, Y, grid_role=L.GRID_TX, **kw2)
    ctx.set_candidate_mask(None)
    assert np.count_nonzero(got2) > 20
    _compare(got2, _oracle(walls, tx, X, Y, allowed=allowed, **kw2), function)
    _compare(got_tx, _oracle(walls, tx, X, Y, allowed=allowed, grid_role="tx", **kw2), function)


@pytest.mark.parametrize("mode", ["hard", "hsig"])
def test_cfg4_full_size_against_sampled_oracle_cells(ctx, mode):
    """BASELINE.json configs[3] at FULL size -- 200 walls, 2048 x 2048 cells, orders 0..3 = 7 960 201 candidates per cell
    (3.3e13 candidate evaluations) -- against 48 cells the C oracle computed in full (scripts/make_golden_cfg4.py, ~10 s
    of CPU per cell and core): bit for bit, plus order 3 on its own."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"cfg4_samples_{mode}.npz"))
    kw = dict(approx=False) if mode == "hard" else dict(approx=True, function="hard_sigmoid")
    tx, walls = random_scene(200, seed=1234)
    x = np.linspace(0.0, 1.0, 2048).astype(F)
    X, Y = np.meshgrid(x, x)
    ij = z["ij"]
    ctx.set_scene(walls)
    got = ctx.power_map(tx, X, Y, min_order=0, max_order=3, **kw)
    assert np.array_equal(got[ij[:, 0], ij[:, 1]], z["total"])
    assert np.isfinite(got).all() and (got >= 0).all()
    got3 = ctx.power_map(tx, X, Y, order=3, **kw)
    assert np.array_equal(got3[ij[:, 0], ij[:, 1]], z["per_order"][3])
    assert (z["per_order"][3] != 0).sum() >= 5  # the sample does see third-order paths


@pytest.mark.parametrize("mode", ["hard", "hsig"])
def test_cfg4_full_size_against_contiguous_oracle_blocks(ctx, mode):
    """BASELINE.json configs[3] at FULL size, inside the full 2048^2 launch, against six CONTIGUOUS blocks
    of 64 x 64 cells that the C oracle computed cell by cell (scripts/make_golden_cfg4_blocks.py: per-cell pruning that
    knows nothing of the GPU's patch culling) -- the transmitter's block, where the region candidate lists are longest, two
    of its neighbours (one straddling four top regions), a block in full shadow, the corner and a random one: 24 576 cells
    per mode, every bit.  Both launch shapes of the sweep kernel (one / four patches per workgroup)."""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg4_blocks.npz"))
    B = int(z["block_size"])
    assert int(z["grid"]) == 2048 and z[mode].shape == (len(z["blocks"]), B, B)
    kw = dict(approx=False) if mode == "hard" else dict(approx=True, function="hard_sigmoid")
    tx, walls = random_scene(200, seed=1234)
    x = np.linspace(0.0, 1.0, 2048).astype(F)
    X, Y = np.meshgrid(x, x)
    ctx.set_scene(walls)
    for waves in (0, 1):
        ctx.set_option("fwd_waves", waves)
        try:
            got = ctx.power_map(tx, X, Y, min_order=0, max_order=3, **kw)
        finally:
            ctx.set_option("fwd_waves", 0)
        bad = 0
        for b, (i0, j0) in enumerate(z["blocks"]):
            blk = got[i0 : i0 + B, j0 : j0 + B]
            n = int((~((blk == z[mode][b]) | (np.isnan(blk) & np.isnan(z[mode][b])))).sum())
            print(f"   cfg4 {mode} block {b} at ({int(i0)}, {int(j0)}): {n} of {B * B} cells differ; {int((z[mode][b] != 0).sum())} non-zero")
            bad += n
        assert bad == 0
    assert sum(int((z[mode][b] != 0).sum()) for b in range(len(z["blocks"]))) >= 4096  # the blocks do see paths


def test_bad_inputs_are_rejected_or_propagated(ctx):
    from differt2d_amd import _lib as L

    with pytest.raises(L.D2DError):
        ctx.set_scene(np.array([[[0.0, np.nan], [1.0, 0.0]]], F))
    with pytest.raises(ValueError):
        ctx.set_grid(np.zeros((3, 4), F), np.zeros((4, 3), F))
    with pytest.raises((L.D2DError, ValueError)):
        ctx.set_grid(np.zeros((0, 4), F), np.zeros((0, 4), F))
    # NaN / inf receiver coordinates: the cell's value is NaN like the reference's, its neighbours are untouched
    tx, walls = random_scene(9, seed=5)
    X, Y = unit_grid(16, 8)
    X = X.copy()
    X[3, 5] = np.nan
    X[6, 1] = np.inf
    ctx.set_scene(walls)
    got = ctx.power_map(tx, X, Y, max_order=2)
    want = _oracle(walls, tx, X, Y, max_order=2)
    assert np.isnan(got[3, 5]) and np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("approx,function", MODES)
def test_tx_grid_culled_kernel_matches_oracle_and_exhaustive_kernel(ctx, approx, function):
    """accumulate_on_transmitters_grid_over_paths (scene.py:1489-1648): the cells are transmitters.  The culled kernel
    (a path is its own reverse: culling runs from the fixed receiver towards the patch) against the oracle and against
    the exhaustive kernel, orders 0..3."""
    from differt2d_amd import _lib as L

    rx, walls = random_scene(14, seed=21)
    X, Y = unit_grid(45, 38)
    ctx.set_scene(walls)
    kw = dict(min_order=0, max_order=3, approx=approx, function=function)
    got = ctx.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
    _compare(got, CO_tx(walls, rx, X, Y, **kw), function)
    ctx.set_option("txg_exhaustive", 1)
    try:
        ref = ctx.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
    finally:
        ctx.set_option("txg_exhaustive", 0)
    assert np.array_equal(got, ref, equal_nan=True)
    assert (got > 0).mean() > 0.05


def test_tx_grid_cfg2_scene_block(ctx):
    """50 walls, orders 0..2, a 64 x 48 block of the 1024^2 grid as TRANSMITTER positions, receiver fixed."""
    from differt2d_amd import _lib as L

    rx, walls = random_scene(50, seed=1234)
    x = np.linspace(0.0, 1.0, 1024).astype(F)
    X, Y = np.meshgrid(x[930:994], x[960:1008])
    ctx.set_scene(walls)
    for approx in (False, True):
        kw = dict(min_order=0, max_order=2, approx=approx)
        got = ctx.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
        assert np.array_equal(got, CO_tx(walls, rx, X, Y, **kw))
        assert (got > 0).any()


def CO_tx(walls, rx, X, Y, **kw):
    from oracle import c_oracle as CO

    return CO.power_map(walls, rx, X, Y, prune=True, grid_role="tx", **kw)


@pytest.mark.parametrize("grid", [96, 1024])
def test_instrumented_build_writes_the_same_map(grid):
    """d2d_power_map_stats ("same results, not for timing", include/d2d.h): the instrumented kernels' maps against the product
    kernels', bit for bit, in every validity mode -- bench.py's roofline counters are only worth something if the kernel that
    counts does the work the timed kernel does (scripts/stats_cmp.py; they differed at shadow boundaries until round 4)."""
    from bench import workload
    from differt2d_amd.engine import Context, make_params

    tx, walls, X, Y = workload(50, grid)
    with Context(0) as c:
        c.set_scene(walls)
        c.set_grid(X, Y)
        for kw in ({}, dict(approx=True), dict(approx=True, function="sigmoid")):
            for orders in ((0, 2), (0, 0)) if grid == 96 else ((0, 2),):
                p = make_params(min_order=orders[0], max_order=orders[1], **kw)
                for _ in range(3):  # (the third launch runs with the work history and the last-segment masks)
                    c.launch(p, tx)
                Z = c.get_map().copy()
                st = c.launch_stats(p, tx)
                assert np.array_equal(Z, c.get_map(), equal_nan=True), (kw, orders)
                assert st[0] > 0 and st[4] > 0


def test_tx_grid_with_a_degenerate_step_that_still_counts():
    """scripts/fuzz_parity.py, seed 4003, case 1295 (the one mismatch in 26 000 fuzz cases over four rounds): a TX grid, sigmoid
    validity with alpha = 10 and tol = 0.5, walls and cells on a lattice at offset -300.  For one candidate the exact backward
    scan hits un == 0 (geometry.py:1105) in its first step -- the point stays on the receiver, loss = 1 -- and the path still
    counts sigmoid(-8) = 3.4e-4, while the TX-grid culling, which walks the chain from the other end, reasoned about the
    geometric path and dropped it.  TX grids now take the exhaustive kernel whenever a degenerate path is not exactly invalid."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from fuzz_parity import random_case

    from differt2d_amd import _lib as L
    from differt2d_amd.engine import Context
    from oracle import c_oracle as CO

    rng = np.random.default_rng(4003)
    for case in range(1296):
        walls, tx, X, Y, kw, allowed = random_case(rng, big=case % 2 == 1)
    assert kw["function"] == "sigmoid" and kw["alpha"] == 10.0 and len(walls) == 126
    with Context(0) as c:
        c.set_option("hidden_min_tiles", 0)
        c.set_scene(walls)
        for mask in (allowed, np.isin(np.arange(len(walls)), [1, 7]).astype(np.uint8)):
            c.set_candidate_mask(mask)
            got = c.power_map(tx, X, Y, grid_role=L.GRID_TX, **kw)
            want = CO.power_map(walls, tx, X, Y, allowed=mask, prune=True, grid_role="tx", **kw)
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-5 * max(1.0, float(np.abs(want).max())))
            assert (got != want).mean() < 0.01  # (sigmoid: bit-equal but for a host libm with an FMA build)
        assert c.txg_fallbacks() == 2  # (the diagnostic says so: d2d_debug_txg_fallbacks)
        c.power_map(tx, X, Y, grid_role=L.GRID_TX, **dict(kw, alpha=100.0, tol=1e-2))
        c.power_map(tx, X, Y, grid_role=L.GRID_RX, **kw)
        assert c.txg_fallbacks() == 2  # alpha (tol - 0.999) <= -89.5: culled; RX grids never fall back
