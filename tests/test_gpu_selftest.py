"""The kernels' bare fma division chain must be bit-identical to the compiler's correctly rounded x / y wherever
it is used (operands in [2^-62, 2^62] or x == 0); results are also cross-checked against NumPy's IEEE division."""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_bare_division_chain_is_correctly_rounded():
    from differt2d_amd.engine import Context

    rng = np.random.default_rng(7)
    n = 1 << 22
    parts_x, parts_y = [], []
    # uniform mantissas, exponents spread over the safe range; plus ratios near 1 and near powers of two
    for lo, hi in [(-60, 60), (-20, 20), (-2, 2)]:
        e1 = rng.integers(lo, hi, n // 4)
        e2 = rng.integers(lo, hi, n // 4)
        parts_x.append(np.ldexp(rng.random(n // 4, dtype=np.float32) + 1, e1).astype(np.float32) * rng.choice([-1, 1], n // 4).astype(np.float32))
        parts_y.append(np.ldexp(rng.random(n // 4, dtype=np.float32) + 1, e2).astype(np.float32) * rng.choice([-1, 1], n // 4).astype(np.float32))
    y = np.ldexp(rng.random(n // 4, dtype=np.float32) + 1, rng.integers(-10, 10, n // 4)).astype(np.float32)
    k = rng.integers(1, 1 << 24, n // 4).astype(np.float32)
    parts_x.append((y * k).astype(np.float32))  # near-exact quotients
    parts_y.append(y)
    x = np.concatenate(parts_x)
    y = np.concatenate(parts_y)
    x[:1000] = 0.0
    with Context(0) as ctx:
        q_fast, q_ref, q_hostr = ctx.selftest_div(x, y)
    want = (x / y).astype(np.float32)
    assert np.array_equal(q_ref, want), "generic expansion is not IEEE"
    bad = np.flatnonzero(q_fast.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, f"{bad.size} of {x.size} quotients differ, e.g. {x[bad[:3]]} / {y[bad[:3]]}"


def test_device_expf_is_the_host_libms():
    """The sigmoid activation's expf (d2d_kernels.hpp: expf_libm -- glibc's algorithm in double, rounded once) against the C
    library of this host (what oracle/d2d_oracle.c calls): bit for bit over the range the sweeps use, the overflow / underflow
    edges, subnormal results, and the special values."""
    import ctypes

    from differt2d_amd.engine import Context

    libm = ctypes.CDLL("libm.so.6")
    libm.expf.restype = ctypes.c_float
    libm.expf.argtypes = [ctypes.c_float]
    rng = np.random.default_rng(3)
    parts = [(rng.random(60000) * (hi - lo) + lo).astype(np.float32) for lo, hi in ((-104.5, 89.5), (-20, 20), (-1, 1), (-1e-3, 1e-3), (-104, -86))]
    parts.append(np.array([0.0, -0.0, 88.72283, 88.72284, 88.7229, -103.27893, -103.2789, -103.97207, -103.97208, -87.33655, np.inf, -np.inf,
                           1e-40, -1e-40, 3.4e38, -3.4e38], np.float32))
    x = np.concatenate(parts)
    with np.errstate(over="ignore"):
        want = np.array([libm.expf(float(v)) for v in x], np.float32)
    with Context(0) as ctx:
        got = ctx.selftest_expf(x)
        nan = ctx.selftest_expf(np.array([np.nan], np.float32))
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, f"{bad.size} of {x.size} differ, e.g. x = {x[bad[:4]]}: device {got[bad[:4]]}, libm {want[bad[:4]]}"
    assert np.isnan(nan[0])
