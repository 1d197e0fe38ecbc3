"""
ctypes binding of libd2d.so (C ABI: include/d2d.h).  No PyTorch, no JAX, no CPU fallback:
if the HIP library is missing or no MI355X is visible, every compute call raises.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# D2D_LIB selects another build of the same ABI (A/B variants built aside by scripts/ab_build.sh); the product library is
# never overwritten by an experiment.
LIB_PATH = os.environ.get("D2D_LIB") or os.path.join(CSRC, "libd2d.so")

D2D_MAX_ORDER = 4
D2D_NUM_STATS = 16
D2D_COMM_ID_BYTES = 128
D2D_ABI_VERSION = 10

D2D_WALL, D2D_RIS, D2D_VERTEX = 0, 1, 2
SOLVER_IMAGE, SOLVER_MINPATH, SOLVER_FERMAT = 0, 1, 2
ACT_HARD_SIGMOID, ACT_SIGMOID = 0, 1
FUN_RECEIVED_POWER, FUN_LENGTH_SQUARED, FUN_LENGTH, FUN_ONE, FUN_CUSTOM = 0, 1, 2, 3, 4
OUT_OVERWRITE, OUT_ADD = 0, 1
GRID_RX, GRID_TX = 0, 1

STATUS_NAMES = {
    0: "D2D_OK",
    -1: "D2D_ERR_INVALID",
    -2: "D2D_ERR_HIP",
    -3: "D2D_ERR_NO_DEVICE",
    -4: "D2D_ERR_UNSUPPORTED",
    -5: "D2D_ERR_STATE",
    -6: "D2D_ERR_COMM",
}


class D2DError(RuntimeError):
    """A libd2d call returned a negative status."""

    def __init__(self, status: int, message: str):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


class D2DUnsupported(D2DError, NotImplementedError):
    """Feature outside the native closed set (mirrors TypeError/ValueError raised by the reference)."""


class Params(C.Structure):
    """Mirror of ``d2d_params`` (include/d2d.h)."""

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
        ("solver", C.c_int32),
        ("steps", C.c_int32),
        ("out_mode", C.c_int32),
        ("grid_role", C.c_int32),
        ("strict_nan", C.c_int32),
        ("many", C.c_int32),
        ("reserved", C.c_int32 * 1),
    ]


# every symbol include/d2d.h declares: (name, restype, argtypes)
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_ctx = C.c_void_p
SYMBOLS = [
    ("d2d_abi_version", C.c_int, []),
    ("d2d_last_error", C.c_char_p, []),
    ("d2d_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("d2d_device_info", C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    ("d2d_create", C.c_int, [C.c_int, C.POINTER(_ctx)]),
    ("d2d_destroy", None, [_ctx]),
    ("d2d_synchronize", C.c_int, [_ctx]),
    ("d2d_set_scene", C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    ("d2d_set_candidate_mask", C.c_int, [_ctx, C.c_void_p]),
    ("d2d_count_candidates", C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    ("d2d_enumerate_candidates", C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    ("d2d_num_candidates", C.c_int, [_ctx, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    ("d2d_list_candidates", C.c_int, [_ctx, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]),
    ("d2d_set_grid", C.c_int, [_ctx, _f32p, _f32p, C.c_int32, C.c_int32]),
    ("d2d_set_grid_versioned", C.c_int, [_ctx, _f32p, _f32p, C.c_int32, C.c_int32, C.c_uint64]),
    ("d2d_debug_grid_reuses", C.c_int, [_ctx, C.POINTER(C.c_int64)]),
    ("d2d_debug_sweep_shape", C.c_int, [_ctx, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("d2d_debug_txg_fallbacks", C.c_int, [_ctx, C.POINTER(C.c_int64)]),
    ("d2d_debug_hidden_masks", C.c_int, [_ctx, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    ("d2d_set_optimizer", C.c_int, [_ctx, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double]),
    ("d2d_power_map_launch", C.c_int, [_ctx, C.POINTER(Params), _f32p]),
    ("d2d_set_cotangent", C.c_int, [_ctx, C.c_void_p]),
    ("d2d_set_path_fun_values", C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int64]),
    ("d2d_power_map_vg_launch", C.c_int, [_ctx, C.POINTER(Params), _f32p, C.c_int32]),
    ("d2d_get_grad_rx", C.c_int, [_ctx, _f32p]),
    ("d2d_get_scene_vjp", C.c_int, [_ctx, _f32p, C.c_void_p, C.c_void_p]),
    ("d2d_power_map_stats", C.c_int, [_ctx, C.POINTER(Params), _f32p, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]),
    ("d2d_set_option", C.c_int, [_ctx, C.c_char_p, C.c_int64]),
    ("d2d_debug_set_schedule", C.c_int, [_ctx, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"), C.c_int64]),
    ("d2d_debug_get_schedule", C.c_int, [_ctx, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"),
                                         np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), C.c_int64]),
    ("d2d_debug_get_work", C.c_int, [_ctx, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS"), C.c_int64]),
    ("d2d_debug_region_stats", C.c_int, [_ctx, np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]),
    ("d2d_debug_nan_scan", C.c_int, [_ctx, np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]),
    ("d2d_last_kernel_ms", C.c_int, [_ctx, C.POINTER(C.c_float)]),
    ("d2d_power_map_wave_cycles", C.c_int, [_ctx, C.POINTER(Params), _f32p, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS"),
                                            C.c_int64, C.POINTER(C.c_int64)]),
    ("d2d_selftest_div", C.c_int, [_ctx, _f32p, _f32p, C.c_int64, _f32p, _f32p, _f32p]),
    ("d2d_selftest_expf", C.c_int, [_ctx, _f32p, C.c_int64, _f32p]),
    ("d2d_get_map", C.c_int, [_ctx, _f32p]),
    ("d2d_power_map", C.c_int, [_ctx, C.POINTER(Params), _f32p, _f32p, _f32p, C.c_int32, C.c_int32, _f32p]),
    ("d2d_trace_paths", C.c_int, [_ctx, C.POINTER(Params), _f32p, _f32p, C.c_int32, _i32p, _i32p, C.c_int32,
                                  C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, _f32p, _f32p, _f32p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    ("d2d_set_theta0", C.c_int, [_ctx, C.c_void_p, C.c_int64]),
    ("d2d_comm_unique_id", C.c_int, [C.c_void_p]),
    ("d2d_comm_init", C.c_int, [_ctx, C.c_void_p, C.c_int32, C.c_int32]),
    ("d2d_comm_destroy", C.c_int, [_ctx]),
    ("d2d_comm_count", C.c_int, [_ctx, C.POINTER(C.c_int32)]),
    ("d2d_comm_allgather_map", C.c_int, [_ctx, C.c_int32]),
    ("d2d_comm_gather_map", C.c_int, [_ctx, C.c_int32, C.c_int32]),
    ("d2d_comm_get_gathered", C.c_int, [_ctx, C.c_int32, _f32p, C.c_int64]),
    ("d2d_comm_allreduce_vjp", C.c_int, [_ctx]),
    ("d2d_comm_allreduce_host", C.c_int, [_ctx, np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.c_int32, C.c_int32]),
    ("d2d_timer_begin", C.c_int, [_ctx]),
    ("d2d_timer_end", C.c_int, [_ctx, C.POINTER(C.c_float)]),
]

_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile libd2d.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "d2d.h"))
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        jobs = str(max(1, min(8, len(os.sched_getaffinity(0)))))  # one object per (kernel family, mode): builds in parallel
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j", jobs, "libd2d.so"] + (["-B"] if force else []))
    return LIB_PATH


def load():
    """Loads libd2d.so (never builds implicitly, never falls back)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(or `make -C differt2d_amd/csrc`). differt2d_amd has no CPU fallback."
                )
            L = C.CDLL(LIB_PATH)
            for name, restype, argtypes in SYMBOLS:
                fn = getattr(L, name)
                fn.restype = restype
                fn.argtypes = argtypes
            got = L.d2d_abi_version()
            if got != D2D_ABI_VERSION:
                raise ImportError(f"libd2d ABI version {got} != expected {D2D_ABI_VERSION}")
            _lib = L
    return _lib


def check(status: int):
    if status < 0:
        msg = load().d2d_last_error().decode("utf-8", "replace")
        raise (D2DUnsupported if status == -4 else D2DError)(status, msg)


def device_count() -> int:
    n = C.c_int(0)
    check(load().d2d_device_count(C.byref(n)))
    return n.value


def count_candidates(num_nodes: int, min_order: int = 0, max_order: int = 1, allowed=None) -> int:
    """How many candidates :func:`enumerate_candidates` would return (no list is built)."""
    lib = load()
    a = None if allowed is None else np.ascontiguousarray(allowed, dtype=np.uint8)
    ap = None if a is None else a.ctypes.data_as(C.c_void_p)
    n = C.c_int64(0)
    check(lib.d2d_count_candidates(int(num_nodes), ap, int(min_order), int(max_order), C.byref(n)))
    return int(n.value)


def enumerate_candidates(num_nodes: int, min_order: int = 0, max_order: int = 1, allowed=None):
    """Host-native candidate enumeration (no GPU needed): list of int32 arrays of shape (k,)."""
    lib = load()
    a = None if allowed is None else np.ascontiguousarray(allowed, dtype=np.uint8)
    ap = None if a is None else a.ctypes.data_as(C.c_void_p)
    n = C.c_int64(0)
    check(lib.d2d_count_candidates(int(num_nodes), ap, int(min_order), int(max_order), C.byref(n)))
    cand = np.empty((max(n.value, 1), D2D_MAX_ORDER), np.int32)
    order = np.empty(max(n.value, 1), np.int32)
    check(lib.d2d_enumerate_candidates(int(num_nodes), ap, int(min_order), int(max_order),
                                       cand.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p), n.value))
    return [cand[i, : order[i]].copy() for i in range(n.value)]
