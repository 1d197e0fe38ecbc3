"""ctypes binding of libd2d.so (MI355X fused power-map sweep) -- the file a DiffeRT2d maintainer drops into the
reference as ``differt2d/_d2d.py``.  It depends on ``ctypes`` and NumPy only and mirrors
``include/d2d.h`` (ABI version 10); ``tests/test_gpu_integration.py`` imports THIS file and checks it against the
repository's own engine.

The library is found through ``$DIFFERT2D_LIBD2D`` (default: ``libd2d.so`` on the loader path).
"""

import ctypes as C
import os

import numpy as np

D2D_ABI_VERSION = 10
D2D_GRID_RX, D2D_GRID_TX = 0, 1
D2D_OUT_OVERWRITE, D2D_OUT_ADD = 0, 1
D2D_FUN_RECEIVED_POWER = 0
D2D_SOLVER_IMAGE = 0
D2D_ACT_HARD_SIGMOID, D2D_ACT_SIGMOID = 0, 1

_lib = C.CDLL(os.environ.get("DIFFERT2D_LIBD2D", "libd2d.so"))


class Params(C.Structure):  # d2d_params, include/d2d.h
    _fields_ = [
        ("min_order", C.c_int32), ("max_order", C.c_int32), ("approx", C.c_int32), ("act", C.c_int32),
        ("alpha", C.c_float), ("tol", C.c_float), ("patch", C.c_float), ("seg_tol", C.c_float),
        ("fun_id", C.c_int32), ("r_coef", C.c_float), ("height", C.c_float), ("solver", C.c_int32),
        ("steps", C.c_int32), ("out_mode", C.c_int32), ("grid_role", C.c_int32), ("strict_nan", C.c_int32),
        ("many", C.c_int32), ("reserved", C.c_int32 * 1),
    ]


_f32 = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_lib.d2d_abi_version.restype = C.c_int
_lib.d2d_last_error.restype = C.c_char_p
_lib.d2d_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
_lib.d2d_destroy.argtypes = [C.c_void_p]
_lib.d2d_destroy.restype = None
_lib.d2d_set_scene.argtypes = [C.c_void_p, _f32, C.c_void_p, C.c_void_p, C.c_int32]
_lib.d2d_set_candidate_mask.argtypes = [C.c_void_p, C.c_void_p]
_lib.d2d_set_grid.argtypes = [C.c_void_p, _f32, _f32, C.c_int32, C.c_int32]
_lib.d2d_power_map_launch.argtypes = [C.c_void_p, C.POINTER(Params), _f32]
_lib.d2d_power_map_vg_launch.argtypes = [C.c_void_p, C.POINTER(Params), _f32, C.c_int32]
_lib.d2d_get_map.argtypes = [C.c_void_p, _f32]
_lib.d2d_get_grad_rx.argtypes = [C.c_void_p, _f32]

if _lib.d2d_abi_version() != D2D_ABI_VERSION:
    raise ImportError(f"libd2d ABI version {_lib.d2d_abi_version()} != {D2D_ABI_VERSION} (this binding)")


def _check(rc):
    if rc < 0:
        raise RuntimeError(_lib.d2d_last_error().decode())


_ctx = C.c_void_p()


def ctx():
    """One context (GPU 0, one HIP stream) per process, created on first use; raises without a gfx950 device."""
    if not _ctx.value:
        _check(_lib.d2d_create(0, C.byref(_ctx)))
    return _ctx


def make_params(min_order, max_order, approx, act, alpha, tol, patch, r_coef, height, grid_role=D2D_GRID_RX):
    p = Params()
    p.min_order, p.max_order, p.approx, p.act = int(min_order), int(max_order), int(bool(approx)), int(act)
    p.alpha, p.tol, p.patch, p.seg_tol = float(alpha), float(tol), float(patch), 0.005  # geometry.py:89
    p.fun_id, p.r_coef, p.height = D2D_FUN_RECEIVED_POWER, float(r_coef), float(height)
    p.solver, p.steps, p.out_mode, p.grid_role, p.strict_nan, p.many = D2D_SOLVER_IMAGE, 100, D2D_OUT_OVERWRITE, int(grid_role), 0, 1
    return p


def set_problem(walls, allowed, X, Y):
    """Uploads the scene (walls [N,2,2] f32), the candidate filter ([N] truthy or None) and the grid (X, Y [m,n])."""
    c = ctx()
    walls = np.ascontiguousarray(walls, np.float32).reshape(-1, 2, 2)
    _check(_lib.d2d_set_scene(c, walls.reshape(-1) if walls.size else np.zeros(1, np.float32), None, None, len(walls)))
    a = None if allowed is None else np.ascontiguousarray(allowed, np.uint8)
    _check(_lib.d2d_set_candidate_mask(c, None if a is None else a.ctypes.data_as(C.c_void_p)))
    X, Y = np.ascontiguousarray(X, np.float32), np.ascontiguousarray(Y, np.float32)
    _check(_lib.d2d_set_grid(c, X, Y, *X.shape))
    return X.shape


def sweep(params, fixed_point, add, grad):
    """One launch for one transmitter (RX grid) / receiver (TX grid); ``add`` accumulates into the resident maps
    (reduce_all, scene.py:1939-1952)."""
    params.out_mode = D2D_OUT_ADD if add else D2D_OUT_OVERWRITE
    t = np.ascontiguousarray(fixed_point, np.float32).reshape(2)
    _check(_lib.d2d_power_map_vg_launch(ctx(), C.byref(params), t, 0) if grad else _lib.d2d_power_map_launch(ctx(), C.byref(params), t))


def fetch(shape, grad):
    """Z [m,n] -- and dZ [m,n,2] with ``grad``."""
    Z = np.empty(shape, np.float32)
    _check(_lib.d2d_get_map(ctx(), Z))
    if not grad:
        return Z
    dZ = np.empty((*shape, 2), np.float32)
    _check(_lib.d2d_get_grad_rx(ctx(), dZ.reshape(-1)))
    return Z, dZ
