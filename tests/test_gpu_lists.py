"""
GPU: the region candidate lists (region_list_kernel / region_refine_kernel and the LISTED sweep kernels) change speed,
never a bit.  Every case compares a launch that reads the lists with the same launch enumerating every prefix in every
patch (option region_lists = 0) and, at sizes the oracle finishes in seconds, with the oracle itself.  Covered: all
validity modes, orders 2..4, the three launch shapes (patches shared by 4 waves prefix by prefix / by 4, 8 or 16 waves candidate by candidate / one wave
per patch with the dearest ones cut in parts), region sizes, a list pool that is too small (the patches of the lists that did not fit are handed to the
enumerating kernel), non-finite cells (their patches never use lists), ragged grids, the value+grad sweep, TX grids.
"""

import numpy as np
import pytest

from conftest import random_scene, unit_grid

pytestmark = pytest.mark.gpu

F = np.float32
MODES = [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")]
SHAPES = {"shared_patches": {"coop_waves": 0, "split_sigmoid": 1, "split_max_tiles": 8192}, "shared_candidates": {"coop_waves": 8, "coop_max_tiles": 1 << 20},
          "one_wave_per_patch": {"split_max_tiles": 0, "sched_min_tiles": 1}}
WAVES = {"shared_patches": (4, False), "shared_candidates": (8, True), "one_wave_per_patch": (1, False)}


def _ctx(**opts):
    from differt2d_amd.engine import Context

    c = Context(0)
    for k, v in opts.items():
        c.set_option(k, v)
    return c


def _same(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("shape", sorted(SHAPES))
@pytest.mark.parametrize("approx,function", MODES)
def test_lists_change_no_bit(shape, approx, function):
    tx, walls = random_scene(14, seed=3)
    X, Y = unit_grid(83, 61)  # ragged: 11 x 8 patches, 3 x 2 leaf regions, 1 top region
    with _ctx(**SHAPES[shape]) as on, _ctx(region_lists=0, **SHAPES[shape]) as off:
        for c in (on, off):
            c.set_scene(walls)
        for lo, hi in [(0, 2), (2, 3), (3, 3), (1, 4)] if not approx or function == "hard_sigmoid" else [(0, 2), (3, 3)]:
            kw = dict(min_order=lo, max_order=hi, approx=approx, function=function)
            a, b = on.power_map(tx, X, Y, **kw), off.power_map(tx, X, Y, **kw)
            st = on.debug_region_stats()
            assert on.sweep_shape() == WAVES[shape]
            assert st["leaf_regions"] == 6 and st["patches_enumerated"] == 0 and st["regions_not_listed"] == 0, st
            assert sum(st["leaf_entries"].values()) > 0
            assert off.debug_region_stats()["leaf_regions"] == 0
            assert _same(a, b), f"orders {lo}..{hi}: {(a != b).sum()} cells differ"


@pytest.mark.parametrize("approx", [False, True])
def test_lists_against_the_oracle_and_region_sizes(approx):
    from differt2d_amd.engine import make_params
    from oracle import c_oracle as CO

    tx, walls = random_scene(20, seed=8)
    X, Y = unit_grid(96, 72)
    kw = dict(min_order=0, max_order=2, approx=approx, function="hard_sigmoid")
    want = CO.power_map(walls, tx, X, Y, prune=True, **kw)
    for opts in ({}, {"region_size": 1, "region_size_top": 1}, {"region_size": 2, "region_size_top": 6}, {"region_size": 8, "region_size_top": 64},
                 {"region_slices": 1}, {"region_slices": 64}, {"split_max_tiles": 0, "sched_min_tiles": 1, "heavy_split": 8},
                 {"coop_waves": 0}, {"coop_waves": 4}, {"coop_waves": 8}, {"coop_waves": 16}, {"coop_waves": 16, "region_size": 8, "region_size_top": 64}):
        with _ctx(**opts) as c:
            c.set_scene(walls)
            got = c.power_map(tx, X, Y, **kw)
            c.launch(make_params(**kw), tx)  # second launch on the same grid: work history, dearest patches cut in parts
            got2 = c.get_map()
            assert c.debug_region_stats()["leaf_regions"] > 0
        assert _same(got, want) and _same(got2, want), opts


def test_list_pool_too_small_falls_back_to_enumeration():
    tx, walls = random_scene(32, seed=12)
    X, Y = unit_grid(128)
    kw = dict(min_order=2, max_order=4, approx=True, function="hard_sigmoid")
    with _ctx(region_lists=0) as off:
        off.set_scene(walls)
        want = off.power_map(tx, X, Y, **kw)
    seen = set()
    for mb in (512, 1):
        with _ctx(region_budget_mb=mb, split_max_tiles=0) as c:
            c.set_scene(walls)
            got = c.power_map(tx, X, Y, **kw)
            st = c.debug_region_stats()
        assert _same(got, want), (mb, st)
        seen.add(st["patches_enumerated"] > 0)
        if mb == 512:
            assert st["patches_enumerated"] == 0 and st["regions_not_listed"] == 0
        elif st["leaf_regions"]:
            # 1 MB holds 1024 chunks: either the lists were refused outright or some of them did not fit
            assert st["regions_not_listed"] > 0 and st["patches_enumerated"] > 0 and st["pool_chunks"] == 1024, st


@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_non_finite_cells_never_use_lists(shape):
    tx, walls = random_scene(10, seed=4)
    X, Y = unit_grid(64)
    X, Y = X.copy(), Y.copy()
    X[5, 7] = np.nan
    Y[40, 41] = np.inf
    X[63, 63] = -np.inf
    kw = dict(min_order=0, max_order=2, approx=False)
    with _ctx(region_size=2, region_size_top=4, **SHAPES[shape]) as on, _ctx(region_lists=0, **SHAPES[shape]) as off:
        on.set_scene(walls)
        off.set_scene(walls)
        a, b = on.power_map(tx, X, Y, **kw), off.power_map(tx, X, Y, **kw)
        st = on.debug_region_stats()
    assert _same(a, b)
    # the three cells sit in two of the four top regions (4 x 4 patches): none of their 2 x 4 leaf regions is listed
    assert np.isnan(a[5, 7]) and st["regions_not_listed"] == 8 and st["patches_enumerated"] == 8 * 4, st


def test_value_and_grad_with_lists():
    tx, walls = random_scene(12, seed=6)
    X, Y = unit_grid(72, 56)
    for approx in (False, True):
        kw = dict(min_order=0, max_order=2, approx=approx, function="hard_sigmoid")
        with _ctx() as on, _ctx(region_lists=0) as off:
            on.set_scene(walls)
            off.set_scene(walls)
            a, b = on.value_and_grads(tx, X, Y, **kw), off.value_and_grads(tx, X, Y, **kw)
            assert on.debug_region_stats()["leaf_regions"] > 0
        assert _same(a["value"], b["value"])
        # gradients: the same sums of the same terms, and the same NaN positions -- the reference's autodiff artefacts come from
        # the NaN scan, whatever the lists did (DESIGN.md "NaN parity"; one NaN cell makes the summed scene VJP NaN, as in the
        # reference)
        for k in ("grad_rx", "tx_bar", "walls_bar"):
            assert np.array_equal(np.isnan(a[k]), np.isnan(b[k])), k
            both = np.isfinite(a[k]) & np.isfinite(b[k])
            assert np.array_equal(a[k][both], b[k][both]), k
        assert np.isfinite(a["grad_rx"]).mean() > 0.9


@pytest.mark.parametrize("approx", [False, True])
def test_tx_grid_lists_change_no_bit(approx):
    """accumulate_on_transmitters_grid_over_paths: the lists are built for the reversed chain (region_list_kernel<.., TXG>)."""
    from differt2d_amd import _lib as L
    from oracle import c_oracle as CO

    rx, walls = random_scene(16, seed=9)
    X, Y = unit_grid(90, 70)
    X, Y = X.copy(), Y.copy()
    for lo, hi in [(0, 2), (2, 3)]:
        kw = dict(min_order=lo, max_order=hi, approx=approx, function="hard_sigmoid")
        with _ctx(sched_min_tiles=1) as on, _ctx(region_lists=0) as off:
            on.set_scene(walls)
            off.set_scene(walls)
            a = on.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
            b = off.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
            st = on.debug_region_stats()
            assert st["leaf_regions"] > 0 and sum(st["leaf_entries"].values()) > 0 and st["patches_enumerated"] == 0
            # value + per-cell d/d tx + scene VJP through the lists
            ga = on.value_and_grads(rx, X, Y, grid_role=L.GRID_TX, **kw)
            gb = off.value_and_grads(rx, X, Y, grid_role=L.GRID_TX, **kw)
        want = CO.power_map(walls, rx, X, Y, prune=True, grid_role="tx", **kw)
        assert _same(a, b) and _same(a, want), (lo, hi)
        assert _same(ga["value"], a)
        for k in ("grad_rx", "tx_bar", "walls_bar"):
            assert np.array_equal(np.isnan(ga[k]), np.isnan(gb[k])), k
            both = np.isfinite(ga[k]) & np.isfinite(gb[k])
            assert np.array_equal(ga[k][both], gb[k][both]), k
        assert np.isfinite(ga["grad_rx"]).mean() > 0.9
    # a cell that is not finite: its patch (and its region's) go to the enumerating TX-grid kernel
    X[3, 4] = np.nan
    with _ctx() as on, _ctx(region_lists=0) as off:
        on.set_scene(walls)
        off.set_scene(walls)
        kw = dict(min_order=0, max_order=2, approx=approx, function="hard_sigmoid")
        a = on.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
        b = off.power_map(rx, X, Y, grid_role=L.GRID_TX, **kw)
        assert on.debug_region_stats()["patches_enumerated"] > 0
    assert _same(a, b)


@pytest.mark.parametrize("approx", [False, True])
def test_cut_patches_hand_over_with_a_moving_transmitter(approx):
    """The parts of a cut patch run in different workgroups (different XCDs) and hand their lists over through global
    memory without cache maintenance (write-through stores, agent-scope loads: DESIGN.md "Hand-over").  A different
    transmitter every launch makes a stale read visible: it would return the previous launch's bytes."""
    from differt2d_amd.engine import make_params

    tx, walls = random_scene(24, seed=21)
    X, Y = unit_grid(192, 160)  # 24 x 20 patches
    txs = [tx, (tx + F([0.013, -0.021])).astype(F), (F([0.9, 0.1]) - tx * F(0.5)).astype(F)]
    p = make_params(min_order=0, max_order=2, approx=approx, function="hard_sigmoid")
    with _ctx(region_lists=0, split_max_tiles=0, sched_min_tiles=1, heavy_split=0) as ref:
        ref.set_scene(walls)
        want = [ref.power_map(t, X, Y, min_order=0, max_order=2, approx=approx, function="hard_sigmoid") for t in txs]
    for cut in (16, 200, 480):
        with _ctx(split_max_tiles=0, sched_min_tiles=1, heavy_split=cut) as c:
            c.set_scene(walls)
            c.set_grid(X, Y)
            for i in range(15):
                k = i % 3
                c.launch(p, txs[k])
                assert _same(c.get_map(), want[k]), (cut, i)
