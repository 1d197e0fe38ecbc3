"""The committed golden fixtures (scripts/make_golden.py) still equal what the oracle computes today."""

import glob
import os

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import ref as R

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


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
