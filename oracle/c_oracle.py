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


class OrcOptParams(C.Structure):
    _fields_ = [
        ("approx", C.c_int32),
        ("act", C.c_int32),
        ("fun_id", C.c_int32),
        ("alpha", C.c_double),
        ("tol", C.c_double),
        ("patch", C.c_double),
        ("seg_tol", C.c_double),
        ("r_coef", C.c_double),
        ("height", C.c_double),
        ("solver", C.c_int32),
        ("steps", C.c_int32),
        ("lr", C.c_double),
        ("b1", C.c_double),
        ("b2", C.c_double),
        ("eps", C.c_double),
        ("grid_is_tx", C.c_int32),
        ("g_ulps", C.c_int32),
    ]


FUN_IDS = {"received_power": 0, "length_squared": 1, "length": 2, "one": 3}
ACT_IDS = {"hard_sigmoid": 0, "sigmoid": 1}


def build(force: bool = False) -> str:
    """Compile the C oracle in place (gcc, a few hundred ms)."""
    srcs = [os.path.join(_HERE, f) for f in ("d2d_oracle.c", "d2d_oracle_grad.c", "d2d_oracle_opt.c", "Makefile")]
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
        dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        ip = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        up = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        L.orc_opt_power_map.restype = C.c_int
        L.orc_opt_power_map.argtypes = [C.c_int, dp, up, dp, C.c_int, C.POINTER(OrcOptParams), dp, dp, dp, C.c_long, ip, ip, C.c_long,
                                        dp, dp, C.c_void_p, C.c_void_p, C.c_void_p, ip, C.c_int, C.c_int]
        L.orc_opt_objective.restype = C.c_int
        L.orc_opt_objective.argtypes = [C.c_int, dp, up, dp, C.c_int, C.c_int, dp, dp, ip, C.c_int, dp, dp, dp]
        L.orc_opt_adam_step.restype = C.c_int
        L.orc_opt_adam_step.argtypes = [C.c_int, C.POINTER(OrcOptParams), C.c_int, C.c_double, dp, dp, dp]
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
        out += (kink if with_kink == "sites" else kink.astype(bool),)
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


# ---------------------------------------------------------------------- MinPath / FermatPath sweeps (oracle/d2d_oracle_opt.c)
SOLVER_IDS = {"min": 1, "fermat": 2}


def _opt_scene(kinds, xys, phis, dtype):
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    dt = np.float32 if np.dtype(dtype) == np.float32 else np.float64
    xys = np.ascontiguousarray(np.asarray(xys, dtype=np.float32).reshape(-1, 2, 2), dtype=np.float64)
    # sin / cos of the RIS angles as oracle/ref.py's backend evaluates them (np.sin on the working dtype)
    ph = np.asarray(phis, dtype=np.float64).astype(dt)
    sincos = np.ascontiguousarray(np.stack([np.sin(ph), np.cos(ph)], -1), dtype=np.float64)
    return kinds, xys, sincos


def _opt_cands(cands, theta0s):
    C_ = len(cands)
    ci = np.zeros((max(C_, 1), ORC_MAX_ORDER), np.int32)
    ck = np.zeros(max(C_, 1), np.int32)
    th = np.zeros((max(C_, 1), ORC_MAX_ORDER), np.float64)
    for i, c in enumerate(cands):
        c = np.asarray(c, np.int32).reshape(-1)
        ck[i] = c.size
        ci[i, : c.size] = c
        t = np.asarray(theta0s[i], np.float32).reshape(-1) if theta0s is not None else np.zeros(0, np.float32)
        th[i, : t.size] = t[:ORC_MAX_ORDER]
    return ci, ck, th


def make_opt_params(solver="min", steps=100, approx=False, function="hard_sigmoid", alpha=100.0, tol=1e-2, patch=0.0, seg_tol=0.005,
                    fun="received_power", r_coef=0.5, height=0.1, lr=0.1, b1=0.9, b2=0.999, eps=1e-8, grid_role="rx", g_ulps=0):
    return OrcOptParams(int(bool(approx)), ACT_IDS[function], FUN_IDS[fun], alpha, tol, patch, seg_tol, r_coef, height,
                        SOLVER_IDS[solver], int(steps), lr, b1, b2, eps, 1 if grid_role == "tx" else 0, int(g_ulps))


def opt_power_map(kinds, xys, phis, fixed, X, Y, cands, theta0s, dtype="float32", grad=False, with_paths=False, nthreads=0,
                  snaps=None, fp32_tangents=False, **kw):
    """MinPath / FermatPath power map over a scene of Wall / RIS / Vertex objects (kinds [N]: 0 / 1 / 2, xys [N, 2, 2] -- a
    Vertex keeps its point in row 0 --, phis [N]) for one fixed end point; ``cands``: list of index arrays, ``theta0s``: one
    array of initial guesses per candidate (shared by all cells, scene.py:1887-1890).  dtype float32: the reference's chain;
    float64: the same chain in double (for the conditioning mask).  Returns value [shape] (dtype), and with ``grad`` the
    per-cell gradient [shape + (2,)] float64 (NaN where the reference's reverse mode yields NaN; ``fp32_tangents``: the
    derivative arithmetic itself in fp32 too -- the yardstick for what any fp32 autodiff loses), and with ``with_paths`` the
    solver's interaction points after ``snaps`` (default: all ``steps``) updates [shape + (C, len(snaps), ORC_MAX_ORDER, 2)] and
    the recorded losses [shape + (C,)]."""
    f64 = np.dtype(dtype) == np.float64
    kinds, xys, sincos = _opt_scene(kinds, xys, phis, dtype)
    ci, ck, th = _opt_cands(cands, theta0s)
    Xc = np.ascontiguousarray(np.asarray(X, np.float32), dtype=np.float64)
    Yc = np.ascontiguousarray(np.asarray(Y, np.float32), dtype=np.float64)
    p = make_opt_params(**kw)
    value = np.empty(Xc.shape, np.float64)
    g = np.empty(Xc.shape + (2,), np.float64) if grad else None
    snaps = np.ascontiguousarray([p.steps] if snaps is None else snaps, dtype=np.int32)
    pts = np.empty(Xc.shape + (len(cands), snaps.size, ORC_MAX_ORDER, 2), np.float64) if with_paths else None
    loss = np.empty(Xc.shape + (len(cands),), np.float64) if with_paths else None
    ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = lib().orc_opt_power_map(2 if (fp32_tangents and grad and not f64) else int(f64), xys.reshape(-1) if xys.size else np.zeros(1), kinds if kinds.size else np.zeros(1, np.uint8),
                                 sincos.reshape(-1) if sincos.size else np.zeros(1), int(kinds.size), C.byref(p),
                                 np.ascontiguousarray(np.asarray(fixed, np.float32), dtype=np.float64), Xc.reshape(-1), Yc.reshape(-1), Xc.size,
                                 ci.reshape(-1), ck, len(cands), th.reshape(-1), value.reshape(-1), ptr(g), ptr(pts), ptr(loss), snaps, int(snaps.size), nthreads)
    if rc != 0:
        raise RuntimeError(f"orc_opt_power_map failed: {rc}")
    out = (value.astype(np.float64 if f64 else np.float32),)
    if grad:
        out += (g,)
    if with_paths:
        out += (pts, loss)
    return out[0] if len(out) == 1 else out


def opt_objective(kinds, xys, phis, tx, rx, cand, theta, solver="min", dtype="float32"):
    """(objective value, d objective / d theta) of one candidate at given parameters."""
    f64 = np.dtype(dtype) == np.float64
    kinds, xys, sincos = _opt_scene(kinds, xys, phis, dtype)
    cand = np.ascontiguousarray(np.asarray(cand, np.int32).reshape(-1))
    theta = np.ascontiguousarray(np.asarray(theta, np.float32 if not f64 else np.float64), dtype=np.float64).reshape(-1)
    val, g = np.zeros(1), np.zeros(ORC_MAX_ORDER)
    d = lambda a: np.ascontiguousarray(np.asarray(a, np.float32), dtype=np.float64)  # noqa: E731
    lib().orc_opt_objective(int(f64), xys.reshape(-1), kinds, sincos.reshape(-1), int(kinds.size), SOLVER_IDS[solver], d(tx), d(rx),
                            cand if cand.size else np.zeros(1, np.int32), int(cand.size), np.concatenate([theta, np.zeros(ORC_MAX_ORDER)]), val, g)
    return float(val[0]), g[: theta.size].copy()


def opt_adam_step(t, g, x, mu, nu, dtype="float32", **kw):
    """One optax.adam update (oracle/ref.py:616-638's order) -> (x, mu, nu)."""
    p = make_opt_params(**kw)
    x, mu, nu = (np.array([float(v)]) for v in (x, mu, nu))
    lib().orc_opt_adam_step(int(np.dtype(dtype) == np.float64), C.byref(p), int(t), float(g), x, mu, nu)
    return float(x[0]), float(mu[0]), float(nu[0])


def opt_conditioning(kinds, xys, phis, fixed, X, Y, cands, theta0s, steps, tol_pts=2e-5, tol_val=2e-3, with_grad=False, **kw):
    """Which cells of a MinPath / FermatPath sweep are well conditioned -- decided by the oracle alone (the rule of
    scripts/make_golden_cfg5.py::solver_agreement, here at C speed for whole maps): the solver of EVERY candidate follows the
    same trajectory in the fp64 run, in the fp32 run, in the fp32 runs from a cell one ulp away and from a fixed end point and
    initial guesses one ulp away, and in fp32 runs whose every objective gradient is moved by one ulp either way (what another
    implementation of the same derivative differs by) -- interaction points within ``tol_pts`` after 30, 100, 300 and all
    ``steps`` iterations -- and all the values agree to ``tol_val`` of the map's scale.  Returns dict(value32, value64, stable, dist) with ``dist``
    the largest distance of an fp32 run's value from the fp64 one (the bar no fp32 evaluation can be held below); with_grad
    also grad32 / grad64 (per-cell gradients of the plain fp32 and fp64 runs), grad32t (the fp32 run with its derivative
    arithmetic in fp32 too) and grad32n (the fp32 run from a cell one ulp away)."""
    F = np.float32
    up = lambda a: np.nextafter(np.asarray(a, F), F(np.inf)).astype(F)  # noqa: E731
    X, Y, fixed = np.asarray(X, F), np.asarray(Y, F), np.asarray(fixed, F)
    th_up = [up(t) for t in theta0s]
    variants = ((fixed, X, Y, theta0s, 0), (fixed, up(X), up(Y), theta0s, 0), (up(fixed), X, Y, th_up, 0), (fixed, X, Y, theta0s, 1),
                (fixed, X, Y, theta0s, -1))
    snaps = sorted({s for s in (30, 100, 300, steps) if s <= steps})
    out = {}
    r64 = opt_power_map(kinds, xys, phis, fixed, X, Y, cands, theta0s, dtype="float64", with_paths=True, steps=steps, snaps=snaps,
                        grad=with_grad, **kw)
    v64, p64 = r64[0], r64[-2]
    stable = np.ones(X.shape, bool)
    dist = np.zeros(X.shape)
    for i, (f_, X_, Y_, th_, gu) in enumerate(variants):
        r32 = opt_power_map(kinds, xys, phis, f_, X_, Y_, cands, th_, dtype="float32", with_paths=True, steps=steps, snaps=snaps,
                            grad=with_grad and i <= 1, g_ulps=gu, **kw)
        if with_grad and i == 1:
            out["grad32n"] = r32[1]  # (the cell one ulp away: how far one input ulp moves the derivative through the loop)
        if len(cands):
            with np.errstate(invalid="ignore"):
                stable &= np.abs(r32[-2] - p64).max(axis=(-1, -2, -3, -4)) <= tol_pts
        dist = np.maximum(dist, np.abs(r32[0].astype(np.float64) - v64))
        if i == 0:
            out["value32"] = r32[0]
            if with_grad:
                out["grad32"] = r32[1]
                # the same run with its derivative arithmetic in fp32 as well: what fp32 autodiff (forward or reverse) loses
                out["grad32t"] = opt_power_map(kinds, xys, phis, f_, X_, Y_, cands, th_, dtype="float32", steps=steps, grad=True,
                                               fp32_tangents=True, **kw)[1]
    scale = float(np.nanmax(np.abs(v64))) if v64.size else 0.0
    with np.errstate(invalid="ignore"):
        stable &= dist <= tol_val * scale + tol_val * np.abs(v64)
    # Adam with a fixed step ends many near-grazing solves in a period-2 limit cycle (lr = 0.1: two points some 5e-3 apart), and
    # on which of the two a run sits after `steps` updates is decided when its trajectory enters the cycle -- by rounding: the
    # probes above agree there, a gradient 2 .. 16 ulps away flips it (scripts/diag_cfg5_full.py).  `parity`: cells whose value
    # changes with one more update; `value32_next` is what the other parity gives.
    v_next = opt_power_map(kinds, xys, phis, fixed, X, Y, cands, theta0s, dtype="float32", steps=steps + 1, **kw).astype(np.float64)
    with np.errstate(invalid="ignore"):
        parity = ~(np.abs(v_next - out["value32"].astype(np.float64)) <= 1e-5 * scale + 1e-5 * np.abs(v64))
    out.update(value64=v64, dist=dist, scale=scale, stable=stable, parity=parity, value32_next=v_next)
    if with_grad:
        out["grad64"] = r64[1]
    return out
