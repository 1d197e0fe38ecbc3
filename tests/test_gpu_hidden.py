"""
GPU: the last-segment masks of the leaf regions (hidden_region_kernel, "hidden_masks" option: per region and wall, the bins
of the wall that are certainly hidden from the WHOLE region; candidates whose last interaction point can only lie in
hidden bins leave the region's list / the patch's culling) change speed, never a bit.  They depend on the scene, the grid
and the validity mode only, are built by the second launch in a row that would use them, and must be rebuilt when any of
those changes.  Compared with a context that never builds them and, at sizes the oracle finishes in seconds, with the
oracle: validity modes, orders, launch shapes, scenes with shared corners / collinear / zero-length walls, cells on walls,
moving transmitters, scene / grid / mode changes, candidate masks, accumulation; full-size cfg2 against the committed map.
"""

import os
import zlib

import numpy as np
import pytest

from conftest import random_scene, unit_grid

pytestmark = pytest.mark.gpu

F = np.float32
MODES = [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")]
SHAPES = {"auto": {}, "shared_patches": {"coop_waves": 0, "split_sigmoid": 1, "split_max_tiles": 8192}, "one_wave_per_patch": {"split_max_tiles": 0, "sched_min_tiles": 1}}


def _ctx(**opts):
    from differt2d_amd.engine import Context

    c = Context(0)
    c.set_option("hidden_min_tiles", 0)  # (the default leaves launches below 400 patches without masks: these grids are small)
    for k, v in opts.items():
        c.set_option(k, v)
    return c


def _same(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def _twice(c, tx, X, Y, **kw):
    """Two launches in a row: the second one builds (or already has) the masks."""
    c.power_map(tx, X, Y, **kw)
    return c.power_map(tx, X, Y, **kw)


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("approx,function", MODES)
def test_masks_change_no_bit(shape, approx, function):
    tx, walls = random_scene(22, seed=13)
    X, Y = unit_grid(150, 110)  # ragged: 19 x 14 patches, 5 x 4 leaf regions
    with _ctx(**SHAPES[shape]) as on, _ctx(hidden_masks=0, **SHAPES[shape]) as off:
        on.set_scene(walls)
        off.set_scene(walls)
        for lo, hi in [(0, 2), (1, 1), (2, 3)] if function != "sigmoid" else [(0, 2)]:
            kw = dict(min_order=lo, max_order=hi, approx=approx, function=function)
            b = off.power_map(tx, X, Y, **kw)
            n_off = sum(off.debug_region_stats()["leaf_entries"].values())
            builds0 = on.hidden_masks()[0]
            a1 = on.power_map(tx, X, Y, **kw)
            a2 = on.power_map(tx, X, Y, **kw)
            if hi >= 2:
                assert on.hidden_masks()[1] and on.hidden_masks()[0] <= builds0 + 1
                assert sum(on.debug_region_stats()["leaf_entries"].values()) < n_off  # the masks do drop candidates
            assert off.hidden_masks() == (0, False)
            assert _same(a1, b) and _same(a2, b), (lo, hi, int((a2 != b).sum()))


@pytest.mark.parametrize("approx", [False, True])
def test_against_the_oracle_on_awkward_scenes(approx):
    from oracle import c_oracle as CO

    for n, snap in ((16, False), (14, True), (30, True)):
        tx, walls = random_scene(n, seed=200 + n)
        if snap:
            walls = (np.round(walls * 8) / 8).astype(F)  # shared corners, collinear and zero-length walls
            tx = (np.round(tx * 8) / 8 + F(0.03)).astype(F)
        X, Y = unit_grid(72, 64)
        X, Y = X.copy(), Y.copy()
        X[0, 0], Y[0, 0] = tx                                   # a cell on the transmitter
        X[9, 9], Y[9, 9] = walls[0, 0]                          # on a wall's end point
        X[20, 31], Y[20, 31] = 0.5 * (walls[1, 0] + walls[1, 1])  # on a wall
        kw = dict(min_order=0, max_order=2, approx=approx, function="hard_sigmoid")
        want = CO.power_map(walls, tx, X, Y, prune=2, **kw)
        with _ctx() as c:
            c.set_scene(walls)
            got = _twice(c, tx, X, Y, **kw)
            assert c.hidden_masks() == (1, True)
        assert _same(got, want), (n, snap, int((got != want).sum()))


def test_rebuilt_when_scene_grid_or_mode_change():
    tx, walls = random_scene(18, seed=3)
    tx2, walls2 = random_scene(18, seed=4)
    X, Y = unit_grid(96)
    X2, Y2 = unit_grid(96, 96)
    X2 = (X2 * F(0.5) + F(0.25)).astype(F)
    kw = dict(min_order=0, max_order=2, approx=False)
    kw2 = dict(min_order=0, max_order=2, approx=True, function="hard_sigmoid")
    with _ctx() as c, _ctx(hidden_masks=0) as off:
        steps = [(walls, X, Y, kw), (walls2, X, Y, kw), (walls2, X2, Y2, kw), (walls2, X2, Y2, kw2), (walls2, X2, Y2, dict(kw2, alpha=30.0)), (walls, X, Y, kw)]
        for i, (w, gx, gy, k) in enumerate(steps):
            c.set_scene(w)
            off.set_scene(w)
            want = off.power_map(tx, gx, gy, **k)
            first = c.power_map(tx, gx, gy, **k)
            assert not c.hidden_masks()[1], i        # the scene, the grid or the mode changed: the old masks are not used
            second = c.power_map(tx, gx, gy, **k)
            assert c.hidden_masks() == (i + 1, True), i
            third = c.power_map(tx2, gx, gy, **k)    # another transmitter: the same masks
            assert c.hidden_masks() == (i + 1, True), i
            assert _same(first, want) and _same(second, want) and _same(third, off.power_map(tx2, gx, gy, **k)), i
        # modes that alternate every launch never see the same key twice in a row: no masks, same results
        n = c.hidden_masks()[0]
        for i in range(4):
            k = kw if i % 2 else kw2
            assert _same(c.power_map(tx, X, Y, **k), off.power_map(tx, X, Y, **k))
        assert c.hidden_masks()[0] == n


def test_moving_transmitter_masks_accumulation_and_candidate_masks():
    from differt2d_amd import _lib as L
    from differt2d_amd.engine import make_params

    tx, walls = random_scene(24, seed=21)
    X, Y = unit_grid(128)
    rng = np.random.default_rng(5)
    txs = np.clip(tx + np.cumsum(rng.normal(0, 0.03, (6, 2)), axis=0), 0.02, 0.98).astype(F)
    allowed = np.ones(24, np.uint8)
    allowed[[1, 5, 17]] = 0
    kw = dict(min_order=0, max_order=2, approx=True, function="hard_sigmoid")
    with _ctx() as c, _ctx(hidden_masks=0) as off:
        maps = []
        for k in (c, off):
            k.set_scene(walls)
            k.set_grid(X, Y)
            k.launch(make_params(**kw), txs[0])
            for t in txs[1:]:
                k.launch(make_params(out_mode=L.OUT_ADD, **kw), t)  # reduce_all over transmitters
            total = k.get_map()
            k.set_candidate_mask(allowed)
            masked = [k.power_map(t, X, Y, **kw) for t in txs[:3]]
            k.set_candidate_mask(None)
            maps.append((total, masked, k.value_and_grads(txs[0], X, Y, **kw)))
        assert c.hidden_masks() == (1, True)
    assert _same(maps[0][0], maps[1][0])
    for a, b in zip(maps[0][1], maps[1][1]):
        assert _same(a, b)
    for key in ("value", "grad_rx", "tx_bar", "walls_bar"):  # the value+grad sweep does not consult the masks
        assert np.array_equal(maps[0][2][key], maps[1][2][key], equal_nan=True), key


@pytest.mark.parametrize("mode", ["hard", "hsig"])
def test_cfg2_full_map_with_masks(mode):
    """BASELINE.json configs[1] at full size, every cell, against the committed map of the C oracle -- the launch that builds
    the masks and two launches that use them (work-history schedule, dearest patches cut in four)."""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg2_fullmap_crc.npz"))
    tx, walls = random_scene(50, seed=1234)
    x = np.linspace(0.0, 1.0, 1024).astype(F)
    X, Y = np.meshgrid(x, x)
    kw = dict(approx=False) if mode == "hard" else dict(approx=True, function="hard_sigmoid")
    with _ctx() as c:
        c.set_scene(walls)
        for i in range(4):
            got = c.power_map(tx, X, Y, min_order=0, max_order=2, **kw)
            assert c.hidden_masks() == (min(i, 1), i >= 1)
            crc = np.array([zlib.crc32(np.ascontiguousarray(got[r]).tobytes()) for r in range(1024)], dtype=np.uint32)
            assert np.array_equal(crc, gold[f"rx_{mode}_power_crc"]), (i, int((crc != gold[f"rx_{mode}_power_crc"]).sum()))


@pytest.mark.parametrize("approx", [False, True])
def test_tx_grid_masks(approx):
    """accumulate_on_transmitters_grid_over_paths: the cells are transmitters, the masks are built for the segment from the cell
    to the path's FIRST wall (roles of P3 / P4 swapped) and consulted where the leaf lists are refined.  A context that
    alternates between the two grid roles rebuilds them (the role is part of their key)."""
    from differt2d_amd import _lib as L
    from oracle import c_oracle as CO

    rx, walls = random_scene(20, seed=9)
    X, Y = unit_grid(120, 90)
    for lo, hi in [(0, 2), (2, 3)]:
        kw = dict(min_order=lo, max_order=hi, approx=approx, function="hard_sigmoid")
        with _ctx() as on, _ctx(hidden_masks=0) as off:
            on.set_scene(walls)
            off.set_scene(walls)
            b = off.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
            n_off = sum(off.debug_region_stats()["leaf_entries"].values())
            a1 = on.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
            a2 = on.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
            assert on.hidden_masks() == (1, True) and sum(on.debug_region_stats()["leaf_entries"].values()) < n_off
            r1 = on.power_map(rx, X, Y, **kw)   # the other role: other masks
            assert not on.hidden_masks()[1]
            r2 = on.power_map(rx, X, Y, **kw)
            assert on.hidden_masks() == (2, True)
            a3 = on.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
            r0 = off.power_map(rx, X, Y, **kw)
        assert _same(a1, b) and _same(a2, b) and _same(a3, b) and _same(r1, r0) and _same(r2, r0), (lo, hi)
        if hi <= 2:
            assert _same(a2, CO.power_map(walls, rx, X, Y, prune=2, grid_role="tx", **kw))


def test_cfg2_full_tx_grid_map_with_masks():
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg2_fullmap_crc.npz"))
    from differt2d_amd import _lib as L

    tx, walls = random_scene(50, seed=1234)
    x = np.linspace(0.0, 1.0, 1024).astype(F)
    X, Y = np.meshgrid(x, x)
    with _ctx() as c:
        c.set_scene(walls)
        for mode, kw in (("hard", dict(approx=False)), ("hsig", dict(approx=True, function="hard_sigmoid"))):
            for i in range(3):
                got = c.power_map(tx, X, Y, min_order=0, max_order=2, grid_role=L.GRID_TX, **kw)
                crc = np.array([zlib.crc32(np.ascontiguousarray(got[r]).tobytes()) for r in range(1024)], dtype=np.uint32)
                assert np.array_equal(crc, gold[f"tx_{mode}_power_crc"]), (mode, i, int((crc != gold[f"tx_{mode}_power_crc"]).sum()))
            assert c.hidden_masks()[1]
