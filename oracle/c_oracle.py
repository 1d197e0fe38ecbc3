"""
ctypes loader for the C oracle (oracle/d2d_oracle.c) -- test infrastructure only.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; nothing under ``differt2d_amd/`` does.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libd2d_oracle.so")
ORC_MAX_ORDER = 4


class OrcParams(C.Structure):
    _fields_ = [
        ("min_order", C.c_int32),
        ("max_order", C.c_int32),
        ("approx", C.c_int32),
        ("act", C.c_int32),
        ("alpha", C.c_float),
        ("tol", C.c_float),
        ("patch", C.c_float),
        ("seg_tol", C.c_float),
        ("fun_id", C.c_int32),
        ("r_coef", C.c_float),
        ("height", C.c_float),
        ("prune", C.c_int32),
        ("grid_is_tx", C.c_int32),
    ]


FUN_IDS = {"received_power": 0, "length_squared": 1, "length": 2, "one": 3}
ACT_IDS = {"hard_sigmoid": 0, "sigmoid": 1}


def build(force: bool = False) -> str:
    """Compile the C oracle in place (gcc, a few hundred ms)."""
    srcs = [os.path.join(_HERE, f) for f in ("d2d_oracle.c", "d2d_oracle_grad.c", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or any(os.path.getmtime(_LIB_PATH) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libd2d_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        L.orc_num_candidates.restype = C.c_long
        L.orc_num_candidates.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_power_map.restype = C.c_int
        L.orc_power_map.argtypes = [fp, C.c_int, C.c_void_p, C.POINTER(OrcParams), fp, fp, fp, C.c_long, fp, C.c_int]
        L.orc_power_map_ex.restype = C.c_int
        L.orc_power_map_ex.argtypes = [fp, C.c_int, C.c_void_p, C.POINTER(OrcParams), fp, fp, fp, C.c_long, fp, fp, C.c_int]
        L.orc_eval_candidates.restype = C.c_int
        L.orc_eval_candidates.argtypes = [fp, C.c_int, C.c_void_p, C.POINTER(OrcParams), fp, fp, fp, fp,
                                          np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")]
        L.orc_max_threads.restype = C.c_int
        L.orc_power_map_grad.restype = C.c_int
        L.orc_power_map_grad.argtypes = [fp, C.c_int, C.c_void_p, C.POINTER(OrcParams), fp, fp, fp, C.c_long, fp,
                                         np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def make_params(min_order=0, max_order=1, order=None, approx=False, function="hard_sigmoid", alpha=100.0,
                tol=1e-2, patch=0.0, seg_tol=0.005, fun="received_power", r_coef=0.5, height=0.1, prune=False,
                grid_role="rx"):
    if order is not None:
        min_order = max_order = order
    return OrcParams(min_order, max_order, int(bool(approx)), ACT_IDS[function], alpha, tol, patch, seg_tol,
                     FUN_IDS[fun], r_coef, height, int(prune), 1 if grid_role == "tx" else 0)


def _allowed_ptr(allowed):
    if allowed is None:
        return None, None
    a = np.ascontiguousarray(allowed, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def num_candidates(N, min_order, max_order, allowed=None):
    keep, ptr = _allowed_ptr(allowed)
    return lib().orc_num_candidates(N, ptr, min_order, max_order)


def power_map(walls, tx, X, Y, allowed=None, nthreads=0, **kw):
    """C-oracle power map for one transmitter (grid_role="rx") or one receiver (grid_role="tx", `tx` is then the
    fixed receiver and the grid cells are transmitters); X, Y any shape, returns the same shape."""
    walls = np.ascontiguousarray(walls, dtype=np.float32).reshape(-1, 2, 2)
    Xc = np.ascontiguousarray(X, dtype=np.float32)
    Yc = np.ascontiguousarray(Y, dtype=np.float32)
    out = np.empty(Xc.shape, dtype=np.float32)
    p = make_params(**kw)
    keep, ptr = _allowed_ptr(allowed)
    rc = lib().orc_power_map(walls.reshape(-1) if walls.size else np.zeros(1, np.float32), walls.shape[0], ptr,
                             C.byref(p), np.ascontiguousarray(tx, dtype=np.float32), Xc.reshape(-1), Yc.reshape(-1),
                             Xc.size, out.reshape(-1), nthreads)
    if rc != 0:
        raise RuntimeError(f"orc_power_map failed: {rc}")
    return out


def power_and_count_maps(walls, tx, X, Y, allowed=None, nthreads=0, **kw):
    """One pass of the C oracle that returns (map of `fun`, map of valid-path counts = the same sweep with fun = 1)."""
    walls = np.ascontiguousarray(walls, dtype=np.float32).reshape(-1, 2, 2)
    Xc = np.ascontiguousarray(X, dtype=np.float32)
    Yc = np.ascontiguousarray(Y, dtype=np.float32)
    out, cnt = np.empty(Xc.shape, dtype=np.float32), np.empty(Xc.shape, dtype=np.float32)
    p = make_params(**kw)
    keep, ptr = _allowed_ptr(allowed)
    rc = lib().orc_power_map_ex(walls.reshape(-1) if walls.size else np.zeros(1, np.float32), walls.shape[0], ptr,
                                C.byref(p), np.ascontiguousarray(tx, dtype=np.float32), Xc.reshape(-1), Yc.reshape(-1),
                                Xc.size, out.reshape(-1), cnt.reshape(-1), nthreads)
    if rc != 0:
        raise RuntimeError(f"orc_power_map_ex failed: {rc}")
    return out, cnt


def power_map_grad(walls, tx, X, Y, allowed=None, nthreads=0, with_gabs=False, with_kink=False, with_amp=False, **kw):
    """Value map and per-cell gradient d facc / d cell (oracle/d2d_oracle_grad.c: forward-mode dual numbers through the C
    oracle's op chain, the reference's two reverse-mode NaN traps stated as rules).  Returns (value[shape] float32,
    grad[shape + (2,)] float64, NaN where the reference's autodiff yields NaN) and, with_gabs, the per-cell sum of the
    candidates' |gradient contributions| (the magnitude an fp32 evaluation's rounding scales with); with_kink: a bool map of
    the cells where a minimum / maximum tied between arguments with different tangents (the gradient there is JAX's
    convention -- the mean -- and an evaluation with another rounding may not see the tie at all); with_amp: the largest
    |u| / |u.n| of the backward scans behind the cell's gradient (a pole of the image method nearby: ill-conditioned)."""
    walls = np.ascontiguousarray(walls, dtype=np.float32).reshape(-1, 2, 2)
    Xc = np.ascontiguousarray(X, dtype=np.float32)
    Yc = np.ascontiguousarray(Y, dtype=np.float32)
    value = np.empty(Xc.shape, dtype=np.float32)
    grad = np.empty(Xc.shape + (2,), dtype=np.float64)
    gabs = np.empty(Xc.shape, dtype=np.float64) if with_gabs else None
    kink = np.zeros(Xc.shape, dtype=np.uint8) if with_kink else None
    amp = np.zeros(Xc.shape, dtype=np.float64) if with_amp else None
    p = make_params(**kw)
    keep, ptr = _allowed_ptr(allowed)
    rc = lib().orc_power_map_grad(walls.reshape(-1) if walls.size else np.zeros(1, np.float32), walls.shape[0], ptr,
                                  C.byref(p), np.ascontiguousarray(tx, dtype=np.float32), Xc.reshape(-1), Yc.reshape(-1),
                                  Xc.size, value.reshape(-1), grad.reshape(-1),
                                  None if gabs is None else gabs.ctypes.data_as(C.c_void_p),
                                  None if kink is None else kink.ctypes.data_as(C.c_void_p),
                                  None if amp is None else amp.ctypes.data_as(C.c_void_p), nthreads)
    if rc != 0:
        raise RuntimeError(f"orc_power_map_grad failed: {rc}")
    out = (value, grad)
    if with_gabs:
        out += (gabs,)
    if with_kink:
        out += (kink.astype(bool),)
    if with_amp:
        out += (amp,)
    return out


def eval_candidates(walls, tx, rx, allowed=None, **kw):
    """Per-candidate (valid, fun, indices) for one RX."""
    walls = np.ascontiguousarray(walls, dtype=np.float32).reshape(-1, 2, 2)
    p = make_params(**kw)
    n = num_candidates(walls.shape[0], p.min_order, p.max_order, allowed)
    valid = np.empty(n, np.float32)
    fun = np.empty(n, np.float32)
    idx = np.empty((n, ORC_MAX_ORDER), np.int32)
    keep, ptr = _allowed_ptr(allowed)
    lib().orc_eval_candidates(walls.reshape(-1) if walls.size else np.zeros(1, np.float32), walls.shape[0], ptr,
                              C.byref(p), np.ascontiguousarray(tx, np.float32), np.ascontiguousarray(rx, np.float32),
                              valid, fun, idx)
    return valid, fun, idx


def max_threads():
    return lib().orc_max_threads()
