"""
Thin object wrapper over the C ABI: one :class:`Context` = one MI355X + one HIP stream.

The higher-level mirror of the reference API (:mod:`differt2d_amd.scene`) drives this class;
benchmarks and tests may use it directly.
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from .defaults import DEFAULT_ALPHA, DEFAULT_HEIGHT, DEFAULT_PATCH, DEFAULT_R_COEF

FUN_IDS = {
    "received_power": L.FUN_RECEIVED_POWER,
    "length_squared": L.FUN_LENGTH_SQUARED,
    "length": L.FUN_LENGTH,
    "one": L.FUN_ONE,
    "custom": L.FUN_CUSTOM,  # values and derivatives from the host: Context.set_path_fun_values (value+grad launches only)
}
ACT_IDS = {"hard_sigmoid": L.ACT_HARD_SIGMOID, "sigmoid": L.ACT_SIGMOID}
SOLVER_IDS = {"image": L.SOLVER_IMAGE, "min": L.SOLVER_MINPATH, "fermat": L.SOLVER_FERMAT}


def make_params(
    min_order: int = 0,
    max_order: int = 1,
    order: Optional[int] = None,
    approx: bool = False,
    function: str = "hard_sigmoid",
    alpha: float = DEFAULT_ALPHA,
    tol: float = 1e-2,
    patch: float = DEFAULT_PATCH,
    seg_tol: float = 0.005,
    fun: str = "received_power",
    r_coef: float = DEFAULT_R_COEF,
    height: float = DEFAULT_HEIGHT,
    solver: str = "image",
    steps: int = 100,
    out_mode: int = L.OUT_OVERWRITE,
    grid_role: int = L.GRID_RX,
    strict_nan: bool = False,
    many: int = 1,
) -> L.Params:
    """Builds a ``d2d_params``; keyword names follow the reference's kwargs
    (scene.py:1803-1826, geometry.py:910-919, logic.py:258-267, utils.py:17-24)."""
    if order is not None:
        min_order = max_order = order
    if function not in ACT_IDS:
        raise L.D2DUnsupported(-4, f"activation {function!r} is not native (hard_sigmoid, sigmoid)")
    if fun not in FUN_IDS:
        raise L.D2DUnsupported(-4, f"path function {fun!r} is not native ({', '.join(FUN_IDS)})")
    p = L.Params()
    p.min_order, p.max_order = int(min_order), int(max_order)
    p.approx, p.act = int(bool(approx)), ACT_IDS[function]
    p.alpha, p.tol, p.patch, p.seg_tol = float(alpha), float(tol), float(patch), float(seg_tol)
    p.fun_id, p.r_coef, p.height = FUN_IDS[fun], float(r_coef), float(height)
    p.solver, p.steps, p.out_mode = SOLVER_IDS[solver], int(steps), int(out_mode)
    p.grid_role = int(grid_role)
    p.strict_nan = int(bool(strict_nan))
    p.many = int(many)
    return p


def _immutable(a: np.ndarray) -> bool:
    """Nobody can write to this array's memory through NumPy: it and every array it is a view of are read-only."""
    while isinstance(a, np.ndarray):
        if a.flags.writeable:
            return False
        a = a.base
    return a is None


class Context:
    """Owns a ``d2d_ctx``: device buffers for one scene and one RX grid stay resident in HBM."""

    def __init__(self, device: int = 0):
        self._lib = L.load()
        self._ctx = C.c_void_p()
        L.check(self._lib.d2d_create(int(device), C.byref(self._ctx)))
        self.device = int(device)
        self.shape = None
        self.n_objects = 0
        self._grid_held = None
        self._grid_serial = 0

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.d2d_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(0), C.c_int64(0)
        L.check(self._lib.d2d_device_info(self.device, name, 256, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "cus": cus.value, "mem_bytes": mem.value}

    # -- scene ------------------------------------------------------------------------
    def set_scene(self, xys, kind=None, phi=None):
        xys = np.ascontiguousarray(xys, dtype=np.float32).reshape(-1, 2, 2)
        n = xys.shape[0]
        kind_a = None if kind is None else np.ascontiguousarray(kind, dtype=np.uint8)
        phi_a = None if phi is None else np.ascontiguousarray(phi, dtype=np.float32)
        if kind_a is not None and kind_a.shape != (n,):
            raise ValueError("kind must have one entry per object")
        if phi_a is not None and phi_a.shape != (n,):
            raise ValueError("phi must have one entry per object")
        L.check(
            self._lib.d2d_set_scene(
                self._ctx,
                xys.ctypes.data_as(C.c_void_p) if n else None,
                None if kind_a is None else kind_a.ctypes.data_as(C.c_void_p),
                None if phi_a is None else phi_a.ctypes.data_as(C.c_void_p),
                n,
            )
        )
        self.n_objects = n

    def set_candidate_mask(self, allowed=None):
        if allowed is None:
            L.check(self._lib.d2d_set_candidate_mask(self._ctx, None))
            return
        a = np.ascontiguousarray(allowed, dtype=np.uint8)
        if a.shape != (self.n_objects,):
            raise ValueError("mask must have one entry per object")
        L.check(self._lib.d2d_set_candidate_mask(self._ctx, a.ctypes.data_as(C.c_void_p)))

    def num_candidates(self, min_order=0, max_order=1) -> int:
        n = C.c_int64(0)
        L.check(self._lib.d2d_num_candidates(self._ctx, min_order, max_order, C.byref(n)))
        return n.value

    def list_candidates(self, min_order=0, max_order=1):
        """Returns the reference's candidate list: a python list of int32 arrays of shape (k,)."""
        n = self.num_candidates(min_order, max_order)
        cand = np.empty((max(n, 1), L.D2D_MAX_ORDER), np.int32)
        order = np.empty(max(n, 1), np.int32)
        L.check(
            self._lib.d2d_list_candidates(
                self._ctx, min_order, max_order, cand.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p), n
            )
        )
        return [cand[i, : order[i]].copy() for i in range(n)]

    # -- grid sweep -------------------------------------------------------------------
    def set_grid(self, X, Y):
        """Makes (X, Y) the resident grid.  A grid that is resident already is recognised -- by identity when both arrays
        are immutable (read-only, like the reference's JAX arrays: no byte of them is read then), else byte for byte against
        the library's host copy of the resident arrays --
        and is not uploaded again; the schedule's work history and the regions' boxes stay valid (include/d2d.h)."""
        Xc = np.ascontiguousarray(X, dtype=np.float32)
        Yc = np.ascontiguousarray(Y, dtype=np.float32)
        if Xc.shape != Yc.shape or Xc.ndim != 2:
            raise ValueError(f"X and Y must be 2-D arrays of one shape, got {Xc.shape} and {Yc.shape}")
        token = 0
        if _immutable(Xc) and _immutable(Yc):
            held = self._grid_held
            if held is not None and held[0] is Xc and held[1] is Yc:
                token = held[2]
            else:
                self._grid_serial += 1
                token = self._grid_serial
                self._grid_held = (Xc, Yc, token)  # (keeps the arrays alive: their identity stays theirs)
            L.check(self._lib.d2d_set_grid_versioned(self._ctx, Xc, Yc, Xc.shape[0], Xc.shape[1], token))
        else:
            self._grid_held = None
            L.check(self._lib.d2d_set_grid(self._ctx, Xc, Yc, Xc.shape[0], Xc.shape[1]))
        self.shape = Xc.shape

    def grid_reuses(self) -> int:
        """Diagnostic: set_grid calls that found their grid resident already."""
        n = C.c_int64(0)
        L.check(self._lib.d2d_debug_grid_reuses(self._ctx, C.byref(n)))
        return int(n.value)

    def sweep_shape(self) -> tuple:
        """Diagnostic: (waves per patch, shared candidate by candidate?) of the last RX-grid value sweep (include/d2d.h)."""
        w, k = C.c_int32(0), C.c_int32(0)
        L.check(self._lib.d2d_debug_sweep_shape(self._ctx, C.byref(w), C.byref(k)))
        return int(w.value), bool(k.value)

    def txg_fallbacks(self) -> int:
        """Diagnostic: TX-grid sweeps that ran exhaustively because a degenerate path is not exactly invalid under their
        tol / alpha / activation (include/d2d.h, d2d_params.grid_role)."""
        n = C.c_int64(0)
        L.check(self._lib.d2d_debug_txg_fallbacks(self._ctx, C.byref(n)))
        return int(n.value)

    def hidden_masks(self) -> tuple:
        """Diagnostic: (times the last-segment masks were built, valid now?) (include/d2d.h)."""
        n, v = C.c_int64(0), C.c_int32(0)
        L.check(self._lib.d2d_debug_hidden_masks(self._ctx, C.byref(n), C.byref(v)))
        return int(n.value), bool(v.value)

    def launch(self, params: L.Params, tx):
        tx = np.ascontiguousarray(tx, dtype=np.float32).reshape(2)
        L.check(self._lib.d2d_power_map_launch(self._ctx, C.byref(params), tx))

    def set_cotangent(self, cot=None):
        """Cotangent of the value map for the scene VJP (None = ones)."""
        if cot is None:
            L.check(self._lib.d2d_set_cotangent(self._ctx, None))
            return
        cot = np.ascontiguousarray(cot, dtype=np.float32)
        if cot.shape != self.shape:
            raise ValueError(f"cotangent must have the grid's shape {self.shape}, got {cot.shape}")
        L.check(self._lib.d2d_set_cotangent(self._ctx, cot.ctypes.data_as(C.c_void_p)))

    def set_path_fun_values(self, f=None, xys_bar=None):
        """A host-evaluated path function for ``launch_vg(make_params(fun="custom", ...))``: ``f`` [C, m, n] and its derivative
        w.r.t. the path points ``xys_bar`` [C, m, n, D2D_MAX_ORDER + 2, 2], candidates in the sweep's order (include/d2d.h)."""
        if f is None:
            L.check(self._lib.d2d_set_path_fun_values(self._ctx, None, None, 0))
            return
        f = np.ascontiguousarray(f, dtype=np.float32)
        xys_bar = np.ascontiguousarray(xys_bar, dtype=np.float32)
        if f.ndim != 3 or f.shape[1:] != self.shape or xys_bar.shape != f.shape + (L.D2D_MAX_ORDER + 2, 2):
            raise ValueError(f"f must be [C, {self.shape[0]}, {self.shape[1]}] and xys_bar f.shape + ({L.D2D_MAX_ORDER + 2}, 2); "
                             f"got {f.shape} and {xys_bar.shape}")
        L.check(self._lib.d2d_set_path_fun_values(self._ctx, f.ctypes.data_as(C.c_void_p), xys_bar.ctypes.data_as(C.c_void_p), f.shape[0]))

    def launch_vg(self, params: L.Params, tx, scene_vjp: bool = False):
        """Fused value+grad sweep (per-cell d/d rx; optionally the VJP w.r.t. tx and object end points)."""
        tx = np.ascontiguousarray(tx, dtype=np.float32).reshape(2)
        L.check(self._lib.d2d_power_map_vg_launch(self._ctx, C.byref(params), tx, int(bool(scene_vjp))))

    def get_grad_rx(self) -> np.ndarray:
        out = np.empty((*self.shape, 2), np.float32)
        L.check(self._lib.d2d_get_grad_rx(self._ctx, out.reshape(-1)))
        return out

    def get_scene_vjp(self, with_phi: bool = False):
        """Returns (tx_bar[2], xys_bar[N,2,2]) -- and phi_bar[N] (RIS angles) with ``with_phi``."""
        tx_bar = np.zeros(2, np.float32)
        xys_bar = np.zeros((max(self.n_objects, 1), 2, 2), np.float32)
        phi_bar = np.zeros(max(self.n_objects, 1), np.float32)
        L.check(self._lib.d2d_get_scene_vjp(self._ctx, tx_bar, xys_bar.ctypes.data_as(C.c_void_p),
                                            phi_bar.ctypes.data_as(C.c_void_p) if with_phi else None))
        if with_phi:
            return tx_bar, xys_bar[: self.n_objects], phi_bar[: self.n_objects]
        return tx_bar, xys_bar[: self.n_objects]

    def value_and_grads(self, tx, X, Y, cotangent=None, **kw):
        """One call: value map, per-cell grad_rx, tx_bar, walls_bar (mirrors oracle.ref.power_map_value_and_grads)."""
        self.set_grid(X, Y)
        self.set_cotangent(cotangent)
        self.launch_vg(make_params(**kw), tx, scene_vjp=True)
        value = self.get_map()
        tx_bar, walls_bar, phi_bar = self.get_scene_vjp(with_phi=True)
        return {"value": value, "grad_rx": self.get_grad_rx(), "tx_bar": tx_bar, "walls_bar": walls_bar, "phi_bar": phi_bar}

    def launch_stats(self, params: L.Params, tx) -> np.ndarray:
        """Runs the instrumented kernel build; returns the executed-work counters (include/d2d.h)."""
        tx = np.ascontiguousarray(tx, dtype=np.float32).reshape(2)
        stats = np.zeros(L.D2D_NUM_STATS, np.uint64)
        L.check(self._lib.d2d_power_map_stats(self._ctx, C.byref(params), tx, stats))
        return stats

    def set_option(self, name: str, value: int) -> None:
        """Launch-shape tuning (``split_max_tiles``, ``sched_min_tiles``): changes speed, never a result bit."""
        L.check(self._lib.d2d_set_option(self._ctx, name.encode(), int(value)))

    def debug_set_schedule(self, order=None) -> None:
        """Diagnostic: later launches of ``len(order)`` patches start their patches in this order (None: built-in)."""
        if order is None:
            L.check(self._lib.d2d_debug_set_schedule(self._ctx, np.zeros(1, np.int32), 0))
        else:
            o = np.ascontiguousarray(order, dtype=np.int32)
            L.check(self._lib.d2d_debug_set_schedule(self._ctx, o, o.size))

    def debug_get_schedule(self, n_patches: int):
        """Diagnostic: (order, cost keys) of the schedule built by the last scheduled launch."""
        order, key = np.empty(n_patches, np.int32), np.empty(n_patches, np.uint8)
        L.check(self._lib.d2d_debug_get_schedule(self._ctx, order, key, n_patches))
        return order, key

    def debug_get_work(self, n_patches: int) -> np.ndarray:
        """Diagnostic: the work (units of ~25 wave-instructions) each patch took in the last culled sweep."""
        out = np.empty(n_patches, np.uint32)
        L.check(self._lib.d2d_debug_get_work(self._ctx, out, n_patches))
        return out

    def debug_region_stats(self) -> dict:
        """Diagnostic: the region candidate lists of the last launch that built any (include/d2d.h)."""
        o = np.zeros(8, np.int64)
        L.check(self._lib.d2d_debug_region_stats(self._ctx, o))
        return {"pool_chunks_used": int(o[0]), "pool_chunks": int(o[1]), "patches_enumerated": int(o[2]), "regions_not_listed": int(o[3]),
                "leaf_entries": {2: int(o[4]), 3: int(o[5]), 4: int(o[6])}, "leaf_regions": int(o[7])}

    def debug_nan_scan(self) -> dict:
        """Diagnostic: counters of the last value+grad launch's NaN scan (needs ``set_option("nan_scan_stats", 1)``)."""
        o = np.zeros(6, np.int64)
        L.check(self._lib.d2d_debug_nan_scan(self._ctx, o))
        return {"probes": int(o[0]), "nan_cells": int(o[1]), "nan_patches": int(o[2]), "self_probes": int(o[3]), "bad_items": int(o[4]),
                "rounds": int(o[5])}

    def last_kernel_ms(self) -> float:
        """Duration of the sweep kernel of the last launch (needs ``set_option("time_kernel", 1)``)."""
        ms = C.c_float(0.0)
        L.check(self._lib.d2d_last_kernel_ms(self._ctx, C.byref(ms)))
        return float(ms.value)

    def wave_cycles(self, params: L.Params, tx) -> np.ndarray:
        """Shader-clock ticks per wave (8 x 8 patch) of the instrumented forward sweep, shape (patch rows, patch cols)."""
        tx = np.ascontiguousarray(tx, dtype=np.float32).reshape(2)
        ty, tx_ = -(-self.shape[0] // 8), -(-self.shape[1] // 8)
        out = np.zeros(ty * tx_, np.uint64)
        n = C.c_int64(0)
        L.check(self._lib.d2d_power_map_wave_cycles(self._ctx, C.byref(params), tx, out, out.size, C.byref(n)))
        return out.reshape(ty, tx_)

    def selftest_div(self, x, y):
        """(q_fast, q_ref, q_hostr) of the division self-test (include/d2d.h)."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        y = np.ascontiguousarray(y, dtype=np.float32).reshape(-1)
        outs = [np.empty_like(x) for _ in range(3)]
        L.check(self._lib.d2d_selftest_div(self._ctx, x, y, x.size, *outs))
        return outs

    def selftest_expf(self, x) -> np.ndarray:
        """The device's expf of the sigmoid activation (include/d2d.h: must equal the host libm's bit for bit)."""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
        y = np.empty_like(x)
        L.check(self._lib.d2d_selftest_expf(self._ctx, x, x.size, y))
        return y

    def get_map(self) -> np.ndarray:
        out = np.empty(self.shape, np.float32)
        L.check(self._lib.d2d_get_map(self._ctx, out))
        return out

    def synchronize(self):
        L.check(self._lib.d2d_synchronize(self._ctx))

    def power_map(self, tx, X, Y, **kw) -> np.ndarray:
        """One transmitter, one grid: upload, sweep, download."""
        self.set_grid(X, Y)
        self.launch(make_params(**kw), tx)
        return self.get_map()

    # -- individual paths -------------------------------------------------------------
    def set_theta0(self, theta0):
        """Initial guesses ``[C, <=D2D_MAX_ORDER]`` of the MinPath / FermatPath solvers for the next sweeps."""
        t = np.zeros((len(theta0), L.D2D_MAX_ORDER), np.float32)
        for i, row in enumerate(theta0):
            row = np.asarray(row, np.float32).reshape(-1)
            t[i, : row.size] = row
        L.check(self._lib.d2d_set_theta0(self._ctx, t.ctypes.data_as(C.c_void_p) if t.size else None, t.shape[0]))

    def set_optimizer(self, optimizer=None):
        """The optimiser of the MinPath / FermatPath solvers for the next sweeps (include/d2d.h: d2d_set_optimizer).  ``None`` =
        the reference's default, ``optax.adam(0.1)``; else an :class:`differt2d_amd.optimize.Adam`."""
        from .optimize import Adam, default_optimizer

        o = default_optimizer() if optimizer is None else optimizer
        if not isinstance(o, Adam):
            raise L.D2DUnsupported(-4, f"optimizer {optimizer!r} is not native: differt2d_amd.optimize.adam(learning_rate, b1, b2, eps) is")
        L.check(self._lib.d2d_set_optimizer(self._ctx, 0, float(o.learning_rate), float(o.b1), float(o.b2), float(o.eps)))

    def trace_paths(self, params: L.Params, tx, rx, candidates, xys_in=None, loss_in=None, theta0=None):
        """Solves (or validates ``xys_in``) every candidate for every (tx, rx) pair on the GPU.

        ``tx``/``rx``: (P, 2); ``candidates``: list of int arrays. Returns a dict of arrays with leading
        shape (P, C): ``xys`` (P, C, D2D_MAX_ORDER+2, 2), ``loss``, ``valid``, ``on``, ``hit``, ``length``."""
        tx = np.ascontiguousarray(tx, dtype=np.float32).reshape(-1, 2)
        rx = np.ascontiguousarray(rx, dtype=np.float32).reshape(-1, 2)
        if tx.shape != rx.shape:
            raise ValueError("tx and rx must pair up")
        P, Cn = tx.shape[0], len(candidates)
        cand = np.full((max(Cn, 1), L.D2D_MAX_ORDER), -1, np.int32)
        order = np.zeros(max(Cn, 1), np.int32)
        for i, c in enumerate(candidates):
            c = np.asarray(c, dtype=np.int32).reshape(-1)
            if c.size > L.D2D_MAX_ORDER:
                raise L.D2DError(-1, f"candidate order {c.size} exceeds D2D_MAX_ORDER={L.D2D_MAX_ORDER}")
            cand[i, : c.size] = c
            order[i] = c.size
        NP = L.D2D_MAX_ORDER + 2
        out = {
            "xys": np.full((P, Cn, NP, 2), np.nan, np.float32),
            "loss": np.zeros((P, Cn), np.float32),
            "valid": np.zeros((P, Cn), np.float32),
            "on": np.zeros((P, Cn), np.float32),
            "hit": np.zeros((P, Cn), np.float32),
            "length": np.zeros((P, Cn), np.float32),
        }
        vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        if xys_in is not None:
            xys_in = np.ascontiguousarray(xys_in, dtype=np.float32).reshape(P, Cn, NP, 2)
        if loss_in is not None:
            loss_in = np.ascontiguousarray(loss_in, dtype=np.float32).reshape(P, Cn)
        th = None
        if theta0 is not None:
            # the library reads C * max(1, many) rows from this pointer: a shorter array would be a host out-of-bounds read
            need = Cn * max(1, int(params.many))
            if params.solver != L.SOLVER_IMAGE and xys_in is None and len(theta0) != need:
                raise ValueError(f"theta0 must have {need} rows (candidates x max(1, many)), got {len(theta0)}")
            th = np.zeros((max(len(theta0), 1), L.D2D_MAX_ORDER), np.float32)
            for i, row in enumerate(theta0):
                row = np.asarray(row, np.float32).reshape(-1)
                th[i, : row.size] = row
        L.check(
            self._lib.d2d_trace_paths(
                self._ctx, C.byref(params), tx, rx, P, cand, order, Cn, vp(th), 0 if theta0 is None else len(theta0),
                vp(xys_in), vp(loss_in),
                out["xys"].reshape(-1), out["loss"].reshape(-1), out["valid"].reshape(-1),
                vp(out["on"]), vp(out["hit"]), vp(out["length"]),
            )
        )
        out["order"] = order[:Cn]
        return out

    # -- RCCL ---------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(L.D2D_COMM_ID_BYTES)
        L.check(L.load().d2d_comm_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != L.D2D_COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        buf = C.create_string_buffer(unique_id, L.D2D_COMM_ID_BYTES)
        L.check(self._lib.d2d_comm_init(self._ctx, C.cast(buf, C.c_void_p), int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)

    def comm_destroy(self):
        L.check(self._lib.d2d_comm_destroy(self._ctx))

    def comm_count(self) -> int:
        """Ranks of the communicator as RCCL reports them (ncclCommCount)."""
        n = C.c_int32(0)
        L.check(self._lib.d2d_comm_count(self._ctx, C.byref(n)))
        return int(n.value)

    def comm_allgather_map(self, grad: bool = False):
        L.check(self._lib.d2d_comm_allgather_map(self._ctx, 1 if grad else 0))

    def comm_gather_map(self, root: int = 0, grad: bool = False):
        """Gathers the resident map to ``root`` only (ncclSend / ncclRecv): only ``root`` may fetch it afterwards."""
        L.check(self._lib.d2d_comm_gather_map(self._ctx, 1 if grad else 0, int(root)))

    def comm_get_gathered(self, world: int, grad: bool = False) -> np.ndarray:
        shape = (world, *self.shape, 2) if grad else (world, *self.shape)
        out = np.empty(shape, np.float32)
        L.check(self._lib.d2d_comm_get_gathered(self._ctx, 1 if grad else 0, out.reshape(-1), out.size))
        return out

    def comm_allreduce_vjp(self):
        L.check(self._lib.d2d_comm_allreduce_vjp(self._ctx))

    def comm_allreduce_host(self, values, op: str = "sum") -> np.ndarray:
        """All-reduces a few host doubles over ranks through RCCL (synchronous)."""
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64).copy()
        L.check(self._lib.d2d_comm_allreduce_host(self._ctx, v, v.size, {"sum": 0, "max": 1}[op]))
        return v

    def comm_barrier(self):
        """Stream-synchronising barrier over all ranks of the communicator."""
        self.comm_allreduce_host([1.0], "sum")

    # -- timing -----------------------------------------------------------------------
    def timer_begin(self):
        L.check(self._lib.d2d_timer_begin(self._ctx))

    def timer_end(self) -> float:
        ms = C.c_float(0.0)
        L.check(self._lib.d2d_timer_end(self._ctx, C.byref(ms)))
        return ms.value


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    """Process-wide context per device (created on first use; raises without a GPU)."""
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
