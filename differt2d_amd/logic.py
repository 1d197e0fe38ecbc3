"""
Soft / hard boolean algebra with the reference's names and semantics (``differt2d/logic.py``).

These are host-side NumPy element-wise helpers for user code and for the API mirror; the fused
kernels implement the same algebra on the GPU (``csrc/d2d_kernels.hpp``) and never call this module.

``approx`` follows the reference's three-state convention: ``None`` -> module flag
:data:`ENABLE_APPROX` (initialised from the ``ENABLE_APPROX`` environment variable, logic.py:58),
``True`` -> min/max/activation, ``False`` -> plain boolean logic.
"""

from __future__ import annotations

import os
from contextlib import contextmanager
from threading import RLock
from typing import Callable, Optional, Union

import numpy as np

from .defaults import DEFAULT_ALPHA

__all__ = (
    "ENABLE_APPROX", "activation", "disable_approx", "enable_approx", "greater", "greater_equal",
    "hard_sigmoid", "is_false", "is_true", "less", "less_equal", "logical_all", "logical_and",
    "logical_any", "logical_not", "logical_or", "set_approx", "sigmoid", "true_value", "false_value",
)

ENABLE_APPROX: bool = "ENABLE_APPROX" in os.environ
_LOCK = RLock()
_F = np.float32


def _resolve(approx: Optional[bool]) -> bool:
    return ENABLE_APPROX if approx is None else bool(approx)


def set_approx(enable: bool) -> None:
    """Sets the module-wide default used when ``approx=None`` (reference logic.py:68-92)."""
    global ENABLE_APPROX
    with _LOCK:
        ENABLE_APPROX = bool(enable)


@contextmanager
def enable_approx(enable: bool = True):
    """Context manager that temporarily sets the default (reference logic.py:95-196)."""
    global ENABLE_APPROX
    with _LOCK:
        previous = ENABLE_APPROX
        ENABLE_APPROX = bool(enable)
        try:
            yield
        finally:
            ENABLE_APPROX = previous


@contextmanager
def disable_approx(disable: bool = True):
    """Counterpart of :func:`enable_approx` (reference logic.py:199-215)."""
    with enable_approx(not disable):
        yield


def _f(x):
    return np.asarray(x, dtype=_F)


def sigmoid(x, alpha):
    """``1 / (1 + exp(-alpha x))`` in fp32 (reference logic.py:218-235)."""
    z = _F(alpha) * _f(x)
    with np.errstate(over="ignore"):
        return _F(1.0) / (_F(1.0) + np.exp(-z))


def hard_sigmoid(x, alpha):
    """``relu6(alpha x + 3) / 6`` in fp32 (reference logic.py:238-255)."""
    z = _F(alpha) * _f(x)
    return np.minimum(np.maximum(z + _F(3.0), _F(0.0)), _F(6.0)) / _F(6.0)


sigmoid._d2d_native = "sigmoid"
hard_sigmoid._d2d_native = "hard_sigmoid"


def native_activation_name(function: Union[str, Callable, None]) -> str:
    """Maps the ``function`` kwarg to one of the activations the kernels implement."""
    if function is None:
        return "hard_sigmoid"
    if isinstance(function, str):
        name = function
    else:
        name = getattr(function, "_d2d_native", None)
    if name not in ("hard_sigmoid", "sigmoid"):
        from ._lib import D2DUnsupported

        raise D2DUnsupported(-4, f"activation {function!r} has no native kernel (use hard_sigmoid or sigmoid)")
    return name


def activation(x, alpha=DEFAULT_ALPHA, function: Callable = hard_sigmoid):
    """Smooth 0 -> 1 transition centred at 0 (reference logic.py:258-312)."""
    if isinstance(function, str):
        function = {"hard_sigmoid": hard_sigmoid, "sigmoid": sigmoid}[function]
    return function(x, alpha)


def logical_or(x, y, approx: Optional[bool] = None):
    return np.maximum(x, y) if _resolve(approx) else np.logical_or(x, y)


def logical_and(x, y, approx: Optional[bool] = None):
    return np.minimum(x, y) if _resolve(approx) else np.logical_and(x, y)


def logical_not(x, approx: Optional[bool] = None):
    return np.subtract(_F(1.0), x) if _resolve(approx) else np.logical_not(x)


def greater(x, y, approx: Optional[bool] = None, **kwargs):
    return activation(np.subtract(_f(x), _f(y)), **kwargs) if _resolve(approx) else np.greater(x, y)


def greater_equal(x, y, approx: Optional[bool] = None, **kwargs):
    return activation(np.subtract(_f(x), _f(y)), **kwargs) if _resolve(approx) else np.greater_equal(x, y)


def less(x, y, approx: Optional[bool] = None, **kwargs):
    return activation(np.subtract(_f(y), _f(x)), **kwargs) if _resolve(approx) else np.less(x, y)


def less_equal(x, y, approx: Optional[bool] = None, **kwargs):
    return activation(np.subtract(_f(y), _f(x)), **kwargs) if _resolve(approx) else np.less_equal(x, y)


def logical_all(*x, axis=None, approx: Optional[bool] = None):
    arr = np.asarray(x)
    return np.min(arr, axis=axis) if _resolve(approx) else np.all(arr, axis=axis)


def logical_any(*x, axis=None, approx: Optional[bool] = None):
    arr = np.asarray(x)
    return np.max(arr, axis=axis) if _resolve(approx) else np.any(arr, axis=axis)


def is_true(x, tol=0.5, approx: Optional[bool] = None):
    return np.greater(x, 1.0 - tol) if _resolve(approx) else np.asarray(x)


def is_false(x, tol=0.5, approx: Optional[bool] = None):
    return np.less(x, tol) if _resolve(approx) else np.logical_not(x)


def true_value(approx: Optional[bool] = None):
    return _F(1.0) if _resolve(approx) else np.bool_(True)


def false_value(approx: Optional[bool] = None):
    return _F(0.0) if _resolve(approx) else np.bool_(False)
