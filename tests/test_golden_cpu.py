"""The committed golden fixtures (scripts/make_golden.py) still equal what the oracle computes today."""

import glob
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import ref as R

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if not os.path.basename(p).startswith("cfg"))  # cfg2_fullmap / cfg3_grad / cfg4_samples: full-size fixtures, own tests


def test_fixtures_exist():
    assert len(GOLDEN) >= 12


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden_values(path):
    d = np.load(path)
    kw = eval(str(d["kwargs"]))
    fun_kw = {q: kw.pop(q) for q in ("r_coef", "height") if q in kw}
    got = R.power_map(d["walls"], d["tx"], d["X"], d["Y"], fun_kwargs=fun_kw, **kw)
    assert np.array_equal(got, d["value"])
    c = CO.power_map(d["walls"], d["tx"], d["X"], d["Y"], **kw, **fun_kw)
    if kw.get("function") == "sigmoid":
        np.testing.assert_allclose(c, d["value"], rtol=2e-5, atol=1e-5)
    else:
        assert np.array_equal(c, d["value"])


def test_oracle_fp32_autodiff_matches_golden_fp64():
    d = np.load([p for p in GOLDEN if "random7_o2_hsig" in p][0])
    kw = eval(str(d["kwargs"]))
    g = R.power_map_value_and_grads(d["walls"], d["tx"], d["X"], d["Y"], **kw)
    for k in ("grad_rx", "tx_bar", "walls_bar"):
        scale = np.abs(d[k]).max()
        assert np.abs(g[k] - d[k]).max() <= 2e-5 * scale


def test_cfg4_sample_fixtures_are_consistent():
    """Full-size configs[3] samples (scripts/make_golden_cfg4.py): shapes, and two cells recomputed for orders 0..2."""
    import os

    from conftest import random_scene
    from oracle import c_oracle

    tx, walls = random_scene(200, seed=1234)
    x = np.linspace(0.0, 1.0, 2048).astype(np.float32)
    for mode, kw in (("hard", dict(approx=False)), ("hsig", dict(approx=True, function="hard_sigmoid"))):
        z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"cfg4_samples_{mode}.npz"))
        ij = z["ij"]
        assert z["per_order"].shape == (4, len(ij)) and z["total"].shape == (len(ij),)
        sel = [0, len(ij) - 1]
        X, Y = x[ij[sel, 1]], x[ij[sel, 0]]
        for k in range(3):
            got = c_oracle.power_map(walls, tx, X, Y, min_order=k, max_order=k, prune=True, **kw)
            assert np.array_equal(got, z["per_order"][k][sel])


def test_cfg2_fullmap_fixture_rows_recompute():
    """Full-size configs[1] fixture (scripts/make_golden_fullmap.py: one CRC-32 per row of the oracle's 1024^2 maps):
    a few rows recomputed here give the same CRCs, for both grid roles and both bit-comparable validity modes."""
    import zlib

    from conftest import random_scene

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg2_fullmap_crc.npz"))
    assert int(z["grid"]) == 1024
    tx, walls = random_scene(50, seed=1234)
    x = np.linspace(0.0, 1.0, 1024).astype(np.float32)
    X, Y = np.meshgrid(x, x)
    rows = [0, 517, 1023]
    for role in ("rx", "tx"):
        for mode, kw in (("hard", dict(approx=False)), ("hsig", dict(approx=True, function="hard_sigmoid"))):
            p, c = CO.power_and_count_maps(walls, tx, X[rows], Y[rows], min_order=0, max_order=2, prune=True, grid_role=role, **kw)
            for what, a in (("power", p), ("count", c)):
                crc = z[f"{role}_{mode}_{what}_crc"]
                assert crc.shape == (1024,) and crc.dtype == np.uint32
                assert [zlib.crc32(a[i].tobytes()) for i in range(len(rows))] == [int(crc[r]) for r in rows], (role, mode, what)
