"""
GPU: the kernel of the smallest launches (power_fwd_coop_kernel: a patch shared by 4 / 8 / 16 waves candidate by candidate,
wave 0 adding the contributions in candidate order) changes speed, never a bit.  Compared with the enumerating kernel
(region_lists = 0), with the 4-waves-per-patch kernel and, at sizes the oracle finishes in seconds, with the oracle:
all validity modes, orders 0..4 and sub-ranges, every path function, lists longer than one window of 256 batches, rounds
that end inside a batch, accumulation (D2D_OUT_ADD), repeated launches (work history), ragged grids, non-finite cells
(their patches go to the enumerating kernel's queue), candidate masks, and the automatic choice by launch size.
"""

import numpy as np
import pytest

from conftest import random_scene, unit_grid

pytestmark = pytest.mark.gpu

F = np.float32
MODES = [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")]
BIG = 1 << 20


def _ctx(**opts):
    from differt2d_amd.engine import Context

    c = Context(0)
    for k, v in opts.items():
        c.set_option(k, v)
    return c


def _same(a, b):
    return bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("waves", [4, 8, 16])
@pytest.mark.parametrize("approx,function", MODES)
def test_every_order_range_against_the_oracle(waves, approx, function):
    from oracle import c_oracle as CO

    tx, walls = random_scene(11, seed=31)
    X, Y = unit_grid(45, 37)  # ragged: 6 x 5 patches
    with _ctx(coop_waves=waves, coop_max_tiles=BIG) as c, _ctx(coop_waves=0) as four:
        c.set_scene(walls)
        four.set_scene(walls)
        for lo, hi in [(0, 0), (0, 1), (1, 1), (0, 2), (2, 2), (1, 3), (0, 4), (4, 4)]:
            if function == "sigmoid" and hi > 3:
                continue
            kw = dict(min_order=lo, max_order=hi, approx=approx, function=function)
            got = c.power_map(tx, X, Y, **kw)
            if hi >= 2:  # (orders < 2 build no lists: the 4-waves kernel)
                assert c.sweep_shape() == (waves, True)
            ref = four.power_map(tx, X, Y, **kw)
            assert four.sweep_shape()[1] is False
            assert _same(got, ref), (lo, hi, int((got != ref).sum()))
            if hi <= 3 and function != "sigmoid":
                want = CO.power_map(walls, tx, X, Y, prune=2, **kw)
                assert _same(got, want), (lo, hi)


@pytest.mark.parametrize("fun", ["received_power", "one", "length", "length_squared"])
def test_path_functions_and_masks(fun):
    from oracle import c_oracle as CO

    tx, walls = random_scene(13, seed=5)
    X, Y = unit_grid(40, 40)
    allowed = np.ones(13, np.uint8)
    allowed[[2, 7]] = 0
    kw = dict(min_order=0, max_order=2, approx=True, function="hard_sigmoid", fun=fun)
    with _ctx(coop_waves=16) as c:
        c.set_scene(walls)
        got = c.power_map(tx, X, Y, **kw)
        assert c.sweep_shape() == (16, True)
        c.set_candidate_mask(allowed)
        got_m = c.power_map(tx, X, Y, **kw)
    assert _same(got, CO.power_map(walls, tx, X, Y, prune=2, **kw))
    assert _same(got_m, CO.power_map(walls, tx, X, Y, prune=2, allowed=allowed, **kw))


@pytest.mark.parametrize("waves", [4, 16])
def test_lists_longer_than_a_window(waves):
    """One leaf region whose order-3 list is longer than COOP_MAXB x 64 = 16 384 candidates: processed window by window.
    (alpha = 1: the soft conditions are non-zero almost everywhere and almost nothing can be culled.)"""
    tx, walls = random_scene(44, seed=17)
    X, Y = np.meshgrid(np.linspace(0.0, 0.3, 24, dtype=F), np.linspace(0.0, 0.3, 24, dtype=F))
    kw = dict(min_order=3, max_order=3, approx=True, function="hard_sigmoid", alpha=1.0)
    with _ctx(region_lists=0) as off:
        off.set_scene(walls)
        want = off.power_map(tx, X, Y, **kw)
    with _ctx(coop_waves=waves, coop_max_tiles=BIG, region_size=8, region_size_top=8) as c:
        c.set_scene(walls)
        got = c.power_map(tx, X, Y, **kw)
        st = c.debug_region_stats()
        assert c.sweep_shape() == (waves, True)
    assert st["leaf_regions"] == 1 and st["leaf_entries"][3] > 2 * 16384, st
    assert _same(got, want)


def test_accumulation_and_repeated_launches():
    from differt2d_amd import _lib as L
    from differt2d_amd.engine import make_params

    tx, walls = random_scene(16, seed=23)
    X, Y = unit_grid(64, 48)
    kw = dict(min_order=0, max_order=2, approx=False)
    txs = [tx, (tx + F(0.07)).astype(F), (tx - F(0.05)).astype(F)]
    with _ctx(coop_waves=0) as ref, _ctx() as c:
        maps = []
        for k in (ref, c):
            k.set_scene(walls)
            k.set_grid(X, Y)
            k.launch(make_params(**kw), txs[0])
            for t in txs[1:]:
                k.launch(make_params(out_mode=L.OUT_ADD, **kw), t)
            first = k.get_map()
            for t in txs:  # the same launches again: schedules from the work history of the first round
                k.launch(make_params(**kw), t)
            maps.append((first, k.get_map()))
        assert c.sweep_shape() == (16, True) and ref.sweep_shape() == (4, False)  # 8 x 6 patches: 16 waves by default
    assert _same(maps[0][0], maps[1][0]) and _same(maps[0][1], maps[1][1])


def test_non_finite_cells_go_to_the_queue():
    tx, walls = random_scene(10, seed=4)
    X, Y = unit_grid(64)
    X, Y = X.copy(), Y.copy()
    X[5, 7] = np.nan
    Y[40, 41] = np.inf
    kw = dict(min_order=0, max_order=2, approx=False)
    with _ctx(region_size=2, region_size_top=4, coop_waves=8) as on, _ctx(region_lists=0) as off:
        on.set_scene(walls)
        off.set_scene(walls)
        a, b = on.power_map(tx, X, Y, **kw), off.power_map(tx, X, Y, **kw)
        st = on.debug_region_stats()
        assert on.sweep_shape() == (8, True)
    assert _same(a, b) and np.isnan(a[5, 7]) and st["patches_enumerated"] > 0, st


def test_automatic_choice_by_launch_size():
    """Measured thresholds (d2d.hip): patches -> kernel, per validity mode; only hard validity uses
    the 4-waves-per-patch kernel (the others: one wave per patch beyond the candidate-sharing kernel's range)."""
    tx, walls = random_scene(12, seed=2)
    want = {("hard", 64): (16, True), ("hard", 128): (16, True), ("hard", 136): (8, True), ("hard", 200): (8, True), ("hard", 208): (4, False),
            ("hsig", 128): (16, True), ("hsig", 300): (8, True), ("hsig", 384): (8, True), ("hsig", 392): (1, False),
            ("sig", 200): (16, True), ("sig", 208): (8, True), ("sig", 320): (8, True), ("sig", 328): (4, True), ("sig", 512): (4, True),
            ("sig", 520): (1, False)}
    kws = {"hard": dict(approx=False), "hsig": dict(approx=True, function="hard_sigmoid"), "sig": dict(approx=True, function="sigmoid")}
    seen = {}
    with _ctx() as c:
        c.set_scene(walls)
        for mode, n in want:
            X, Y = unit_grid(n)
            c.power_map(tx, X, Y, min_order=0, max_order=2, **kws[mode])
            seen[(mode, n)] = c.sweep_shape()
    assert seen == want, seen


def test_sigmoid_floor_is_exact():
    """Sigmoid validity: the waves drop what the sums wave 0 held after the last round certainly absorb (acc_floor).  Same
    bits as one wave per patch (which tests against the sum itself) and as the kernel that drops nothing but exact zeros."""
    tx, walls = random_scene(18, seed=41)
    X, Y = unit_grid(56, 40)
    for fun in ("received_power", "one"):
        for alpha in (100.0, 1000.0, 10.0):
            kw = dict(min_order=0, max_order=3, approx=True, function="sigmoid", alpha=alpha, fun=fun)
            maps = []
            for opts in ({"coop_waves": 16}, {"coop_waves": 4}, {"coop_waves": 0}, {"coop_waves": 0, "split_sigmoid": 1, "split_max_tiles": 8192}, {"region_lists": 0}):
                with _ctx(**opts) as c:
                    c.set_scene(walls)
                    maps.append(c.power_map(tx, X, Y, **kw))
            for m in maps[1:]:
                assert _same(maps[0], m), (fun, alpha, int((maps[0] != m).sum()))
