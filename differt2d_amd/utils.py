"""Utilities (reference ``differt2d/utils.py``): the path functions the kernels fuse natively.

Each function here works on host ``Path`` objects (so user code can call it directly) and carries a
``_d2d_native`` tag: when it is passed as ``fun`` to a ``Scene`` sweep it is recognised and evaluated
inside the HIP kernel instead of being called from Python."""

from __future__ import annotations

import numpy as np

from .defaults import DEFAULT_HEIGHT, DEFAULT_R_COEF
from .geometry import Path, Point

P0: float = 100.0
"""Received power at zero distance with the default parameters (reference utils.py:12)."""

F = np.float32


def _integer_pow(x, n: int):
    """x ** n by square-and-multiply in fp32 (the lowering of ``lax.integer_pow``)."""
    x = F(x)
    if n == 0:
        return F(1.0)
    acc = None
    while n > 0:
        if n & 1:
            acc = x if acc is None else F(acc * x)
        n >>= 1
        if n > 0:
            x = F(x * x)
    return acc


def received_power(transmitter, receiver, path: Path, interacting_objects, r_coef: float = DEFAULT_R_COEF,
                   height: float = DEFAULT_HEIGHT):
    """``r_coef ** n / (height**2 + length**2)`` with ``n`` the number of interactions (reference utils.py:17-54)."""
    r = path.length()
    n = path.xys.shape[-2] - 2
    h = F(height)
    return (_integer_pow(r_coef, n) / (h * h + r * r)).astype(F)


def path_length_squared(transmitter, receiver, path: Path, interacting_objects):
    """``path.length() ** 2`` -- the function the reference's accumulate tests use (tests/test_scene.py:444)."""
    r = path.length()
    return (r * r).astype(F)


def path_length_fun(transmitter, receiver, path: Path, interacting_objects):
    """``path.length()``."""
    return path.length()


def one(transmitter, receiver, path: Path, interacting_objects):
    """Constant 1: sweeps then count valid paths per cell."""
    return np.ones(path.xys.shape[:-2], F)


received_power._d2d_native = "received_power"
path_length_squared._d2d_native = "length_squared"
path_length_fun._d2d_native = "length"
one._d2d_native = "one"
