"""
CPU ORACLE -- test infrastructure only, never shipped, never on the product path.

A restatement of DiffeRT2d v0.4.0's hot path (image-method / min-path solvers,
segment intersection, soft visibility, power-map accumulation) in plain array code.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.

Every function cites the reference ``file:line`` (relative to the DiffeRT2d checkout)
it follows.  The arithmetic is written exactly as the reference writes it (same op
order, one rounding per op, fp32 by default); leading batch dimensions broadcast, which
is what ``jax.vmap`` over the RX grid does in ``scene.py:1927-1932``.

Two array backends run the same code:

* ``NUMPY`` (default)  -- values; bit-for-bit deterministic fp32.
* ``TorchBackend()``   -- the same op chain under ``torch.autograd`` (CPU), used to
  obtain reverse-mode gradients with JAX-compatible semantics (``where`` masks the
  untaken branch, ``minimum``/``maximum`` split ties evenly) for the gradient
  fixtures in ``tests/golden/``.

Parity status: JAX cannot be imported in the build container or on the GPU box, so
this oracle is pinned by the reference's own known answers only (tests, doctests and
recorded notebook outputs, see ``tests/test_oracle_known_answers.py``); everything
those do not cover is "restatement only".
"""

from __future__ import annotations

import math
from typing import Any, Callable, Optional, Sequence

import numpy as np

# --------------------------------------------------------------------------------------
# Array backends
# --------------------------------------------------------------------------------------


class NumpyBackend:
    name = "numpy"

    def __init__(self, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.eps = self.dtype.type(np.finfo(self.dtype).eps)

    def c(self, x):
        """Scalar constant in the working dtype."""
        return self.dtype.type(x)

    def asarray(self, x):
        return np.asarray(x, dtype=self.dtype)

    def where(self, c, a, b):
        return np.where(c, a, b).astype(self.dtype, copy=False)

    def minimum(self, a, b):
        return np.minimum(a, b)

    def maximum(self, a, b):
        return np.maximum(a, b)

    def sqrt(self, x):
        return np.sqrt(x)

    def div(self, a, b):
        with np.errstate(divide="ignore", invalid="ignore"):
            return a / b

    def exp(self, x):
        with np.errstate(over="ignore"):
            return np.exp(x)

    def logistic(self, z):
        """lax.logistic: 1 / (1 + exp(-z))."""
        return self.c(1.0) / (self.c(1.0) + self.exp(-z))

    def sin(self, x):
        return np.sin(x)

    def cos(self, x):
        return np.cos(x)

    def stack(self, xs, axis=-1):
        return np.stack(np.broadcast_arrays(*xs), axis=axis)

    def nan_to_num(self, x):
        if np.asarray(x).dtype == np.bool_:
            return x
        return np.nan_to_num(x)

    def logical_and(self, a, b):
        return np.logical_and(a, b)

    def logical_or(self, a, b):
        return np.logical_or(a, b)

    def logical_not(self, a):
        return np.logical_not(a)

    def zeros_like(self, x):
        return np.zeros_like(x, dtype=self.dtype)

    def full_like(self, x, v):
        return np.full_like(x, v, dtype=self.dtype)

    def to_float(self, x):
        return np.asarray(x).astype(self.dtype)

    def inf(self):
        return self.dtype.type(np.inf)


class TorchBackend:
    """Same ops under torch autograd (CPU). Import of torch is deferred to construction."""

    name = "torch"

    def __init__(self, dtype="float32", diff_solver=False):
        import torch

        self.torch = torch
        self.tdtype = getattr(torch, dtype)
        self.eps = torch.finfo(self.tdtype).eps
        # True: MinPath / FermatPath keep the autograd graph through every Adam step (the reference differentiates
        # through its lax.scan, optimize.py:83-97); False: the solver's result is a constant of the outer graph
        self.diff_solver = diff_solver

    def c(self, x):
        return self.torch.tensor(x, dtype=self.tdtype)

    def asarray(self, x):
        if isinstance(x, self.torch.Tensor):
            return x.to(self.tdtype)
        return self.torch.as_tensor(np.asarray(x), dtype=self.tdtype)

    def where(self, c, a, b):
        a = self.asarray(a) if not isinstance(a, self.torch.Tensor) else a
        b = self.asarray(b) if not isinstance(b, self.torch.Tensor) else b
        return self.torch.where(c, a, b)

    def _t(self, a):
        return a if isinstance(a, self.torch.Tensor) else self.asarray(a)

    def minimum(self, a, b):
        return self.torch.minimum(self._t(a), self._t(b))

    def maximum(self, a, b):
        return self.torch.maximum(self._t(a), self._t(b))

    def sqrt(self, x):
        return self.torch.sqrt(x)

    def div(self, a, b):
        return a / b

    def exp(self, x):
        return self.torch.exp(x)

    def logistic(self, z):
        """lax.logistic: value 1 / (1 + exp(-z)); JVP rule logistic(z) * (1 - logistic(z)) (jax/_src/lax/lax.py),
        which stays finite where differentiating the quotient would give 0 * inf."""
        torch = self.torch

        class _Logistic(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                y = 1.0 / (1.0 + torch.exp(-x))
                ctx.save_for_backward(y)
                return y

            @staticmethod
            def backward(ctx, g):
                (y,) = ctx.saved_tensors
                return g * (y * (1.0 - y))

        return _Logistic.apply(z)

    def sin(self, x):
        return self.torch.sin(self._t(x))

    def cos(self, x):
        return self.torch.cos(self._t(x))

    def stack(self, xs, axis=-1):
        xs = self.torch.broadcast_tensors(*[self._t(x) for x in xs])
        return self.torch.stack(xs, dim=axis)

    def nan_to_num(self, x):
        if x.dtype == self.torch.bool:
            return x
        return self.torch.nan_to_num(x)

    def logical_and(self, a, b):
        return self.torch.logical_and(a, b)

    def logical_or(self, a, b):
        return self.torch.logical_or(a, b)

    def logical_not(self, a):
        return self.torch.logical_not(a)

    def zeros_like(self, x):
        return self.torch.zeros_like(x, dtype=self.tdtype)

    def full_like(self, x, v):
        return self.torch.full_like(x, v, dtype=self.tdtype)

    def to_float(self, x):
        return x.to(self.tdtype)

    def inf(self):
        return self.c(float("inf"))


NUMPY = NumpyBackend(np.float32)
NUMPY64 = NumpyBackend(np.float64)

# --------------------------------------------------------------------------------------
# defaults.py:3-15, utils.py:12
# --------------------------------------------------------------------------------------

DEFAULT_ALPHA = 100.0
DEFAULT_PATCH = 0.0
DEFAULT_R_COEF = 0.5
DEFAULT_HEIGHT = 0.1
P0 = 100.0
SEG_TOL = 0.005  # geometry.py:89, not reachable from the sweep kwargs
DEFAULT_TOL = 1e-2  # geometry.py:915

# --------------------------------------------------------------------------------------
# logic.py
# --------------------------------------------------------------------------------------


def sigmoid(x, alpha, xp=NUMPY):
    """logic.py:218-235: jax.nn.sigmoid(alpha * x) = 1 / (1 + exp(-(alpha x)))."""
    z = xp.c(alpha) * x
    return xp.logistic(z)


def hard_sigmoid(x, alpha, xp=NUMPY):
    """logic.py:238-255: jax.nn.hard_sigmoid(alpha x) = relu6(alpha x + 3) / 6,
    relu6(y) = minimum(maximum(y, 0), 6)."""
    z = xp.c(alpha) * x
    return xp.minimum(xp.maximum(z + xp.c(3.0), xp.c(0.0)), xp.c(6.0)) / xp.c(6.0)


ACTIVATIONS = {"hard_sigmoid": hard_sigmoid, "sigmoid": sigmoid}


def activation(x, alpha=DEFAULT_ALPHA, function="hard_sigmoid", xp=NUMPY):
    """logic.py:258-312."""
    f = ACTIVATIONS[function] if isinstance(function, str) else function
    return f(x, alpha, xp=xp)


def logical_or(x, y, approx, xp=NUMPY):
    """logic.py:315-335."""
    return xp.maximum(x, y) if approx else xp.logical_or(x, y)


def logical_and(x, y, approx, xp=NUMPY):
    """logic.py:338-358."""
    return xp.minimum(x, y) if approx else xp.logical_and(x, y)


def logical_not(x, approx, xp=NUMPY):
    """logic.py:361-377."""
    return (xp.c(1.0) - x) if approx else xp.logical_not(x)


def greater(x, y, approx, xp=NUMPY, **kw):
    """logic.py:380-404."""
    return activation(x - y, xp=xp, **kw) if approx else (x > y)


def greater_equal(x, y, approx, xp=NUMPY, **kw):
    """logic.py:407-434."""
    return activation(x - y, xp=xp, **kw) if approx else (x >= y)


def less(x, y, approx, xp=NUMPY, **kw):
    """logic.py:436-460."""
    return activation(y - x, xp=xp, **kw) if approx else (x < y)


def less_equal(x, y, approx, xp=NUMPY, **kw):
    """logic.py:463-487."""
    return activation(y - x, xp=xp, **kw) if approx else (x <= y)


def logical_all(xs: Sequence[Any], approx, xp=NUMPY):
    """logic.py:490-512 (min / all over the listed operands, in order)."""
    out = xs[0]
    for x in xs[1:]:
        out = logical_and(out, x, approx, xp=xp)
    return out


def is_true(x, approx, tol=0.5):
    """logic.py:540-561."""
    return (x > 1.0 - tol) if approx else np.asarray(x)


def is_false(x, approx, tol=0.5):
    """logic.py:564-585."""
    return (x < tol) if approx else np.logical_not(x)


def true_value(approx, xp=NUMPY):
    """logic.py:588-601."""
    return xp.c(1.0) if approx else (xp.c(1.0) > xp.c(0.0))


def false_value(approx, xp=NUMPY):
    """logic.py:604-617."""
    return xp.c(0.0) if approx else (xp.c(1.0) < xp.c(0.0))


# --------------------------------------------------------------------------------------
# geometry.py -- free functions.  Points are arrays (..., 2); X(p), Y(p) take components.
# --------------------------------------------------------------------------------------


def X(p):
    return p[..., 0]


def Y(p):
    return p[..., 1]


def vec(x, y, xp=NUMPY):
    return xp.stack([x, y], axis=-1)


def dot(a, b):
    """jnp.dot on 2-vectors: a0*b0 + a1*b1."""
    return X(a) * X(b) + Y(a) * Y(b)


def norm2(v, xp=NUMPY):
    """jnp.linalg.norm(v) for a 2-vector: sqrt(v0*v0 + v1*v1)."""
    return xp.sqrt(X(v) * X(v) + Y(v) * Y(v))


def segments_intersect(P1, P2, P3, P4, tol=SEG_TOL, approx=False, xp=NUMPY, **kw):
    """geometry.py:82-173."""
    tol = xp.c(tol)
    A = P2 - P1
    B = P3 - P4
    C = P1 - P3
    a = Y(B) * X(C) - X(B) * Y(C)  # :157
    b = X(A) * Y(C) - Y(A) * X(C)  # :158
    d = Y(A) * X(B) - X(A) * Y(B)  # :159

    def test(num, den):  # :163-171
        den_is_zero = den == xp.c(0.0)
        den = xp.where(den_is_zero, xp.c(1.0), den)
        t = xp.where(den_is_zero, xp.inf(), num / den)
        return logical_and(
            greater_equal(t, -tol, approx, xp=xp, **kw),
            less_equal(t, xp.c(1.0) + tol, approx, xp=xp, **kw),
            approx,
            xp=xp,
        )

    return logical_and(test(a, d), test(b, d), approx, xp=xp)


def path_length(points: Sequence[Any], xp=NUMPY):
    """geometry.py:176-203.  ``points`` is a list of (..., 2) arrays."""
    total = None
    for p, q in zip(points[:-1], points[1:]):
        v = (q - p) + xp.eps  # :199-200, eps added to both components
        ln = norm2(v, xp)
        total = ln if total is None else total + ln
    return total


def normalize(v, xp=NUMPY):
    """geometry.py:206-230."""
    length = norm2(v, xp)
    length = xp.where(length == xp.c(0.0), xp.c(1.0), length)
    return vec(X(v) / length, Y(v) / length, xp), length


# --- Wall (geometry.py:542-680); a wall is an array (..., 2, 2) = [origin, dest] ---------


def wall_origin(w):
    return w[..., 0, :]


def wall_dest(w):
    return w[..., 1, :]


def wall_t(w):
    """geometry.py:479-487."""
    return wall_dest(w) - wall_origin(w)


def wall_normal(w, xp=NUMPY):
    """geometry.py:561-573: n = normalize((t_y, -t_x))."""
    t = wall_t(w)
    n, _ = normalize(vec(Y(t), -X(t), xp), xp)
    return n


def wall_parametric_to_cartesian(w, s):
    """geometry.py:581-587."""
    t = wall_t(w)
    return wall_origin(w) + s[..., None] * t


def wall_cartesian_to_parametric(w, p, xp=NUMPY):
    """geometry.py:589-598."""
    other = p - wall_origin(w)
    t = wall_t(w)
    sq = dot(t, t)
    sq = xp.where(sq == xp.c(0.0), xp.c(1.0), sq)
    return dot(t, other) / sq


def wall_contains_parametric(s, approx, xp=NUMPY, **kw):
    """geometry.py:600-621."""
    ge = greater_equal(s, xp.c(0.0), approx, xp=xp, **kw)
    le = less_equal(s, xp.c(1.0), approx, xp=xp, **kw)
    return logical_and(ge, le, approx, xp=xp)


def wall_intersects_cartesian(w, ray0, ray1, patch=DEFAULT_PATCH, approx=False, xp=NUMPY, **kw):
    """geometry.py:623-639."""
    t = wall_t(w)
    patch = xp.c(patch)
    return segments_intersect(
        wall_origin(w) - patch * t,
        wall_dest(w) + patch * t,
        ray0,
        ray1,
        approx=approx,
        xp=xp,
        **kw,
    )


def wall_evaluate_cartesian(w, p0, p1, p2, xp=NUMPY):
    """geometry.py:641-650."""
    i = p1 - p0
    r = p2 - p1
    n = wall_normal(w, xp)
    i, _ = normalize(i, xp)
    r, _ = normalize(r, xp)
    din = dot(i, n)
    e = r - (i - xp.c(2.0) * din[..., None] * n)
    return dot(e, e)


def wall_image_of(w, p, xp=NUMPY):
    """geometry.py:652-670."""
    i = p - wall_origin(w)
    n = wall_normal(w, xp)
    return p - xp.c(2.0) * dot(i, n)[..., None] * n


def ris_evaluate_cartesian(w, phi, p0, p1, p2, xp=NUMPY):
    """geometry.py:698-711."""
    r = p2 - p1
    n = wall_normal(w, xp)
    r, _ = normalize(r, xp)
    mr = -r
    sin_a = X(mr) * Y(n) - Y(mr) * X(n)  # jnp.cross on 2-vectors
    cos_a = dot(mr, n)
    sin_p = xp.sin(phi)
    cos_p = xp.cos(phi)
    ds = sin_a - sin_p
    dc = cos_a - cos_p
    return ds * ds + dc * dc


# --- Objects: kind 0 = Wall, 1 = RIS, 2 = Vertex (geometry.py:352-431, 542-721) ---------

WALL, RIS, VERTEX = 0, 1, 2


class Obj:
    """Minimal stand-in for the reference's Interactable objects (abc.py:129-256)."""

    def __init__(self, kind, xys, phi=math.pi / 4):
        self.kind = kind
        self.xys = xys  # Wall/RIS: (2,2) ; Vertex: (2,) point
        self.phi = phi

    def parameters_count(self):
        return 0 if self.kind == VERTEX else 1  # geometry.py:373-377, 575-579

    def parametric_to_cartesian(self, s, xp=NUMPY):
        if self.kind == VERTEX:
            return self.xys  # :381-385
        return wall_parametric_to_cartesian(self.xys, s)

    def contains(self, p, approx, xp=NUMPY, **kw):
        if self.kind == VERTEX:
            return true_value(approx, xp)  # :397-403
        s = wall_cartesian_to_parametric(self.xys, p, xp)
        return wall_contains_parametric(s, approx, xp=xp, **kw)

    def intersects(self, r0, r1, patch, approx, xp=NUMPY, **kw):
        if self.kind == VERTEX:
            return false_value(approx, xp)  # :407-414
        return wall_intersects_cartesian(self.xys, r0, r1, patch, approx, xp=xp, **kw)

    def evaluate(self, p0, p1, p2, xp=NUMPY):
        if self.kind == VERTEX:
            return xp.c(0.0)  # :418-419
        if self.kind == RIS:
            return ris_evaluate_cartesian(self.xys, xp.asarray(self.phi), p0, p1, p2, xp)
        return wall_evaluate_cartesian(self.xys, p0, p1, p2, xp)


def walls_to_objs(walls, xp=NUMPY):
    walls = xp.asarray(walls)
    return [Obj(WALL, walls[i]) for i in range(walls.shape[0])]


# --------------------------------------------------------------------------------------
# geometry.py -- Path / ImagePath / MinPath / FermatPath
# --------------------------------------------------------------------------------------


def path_loss(objs, pts, xp=NUMPY):
    """geometry.py:1077-1084: sum_k obj_k.evaluate_cartesian(xys[k:k+3])."""
    loss = xp.c(0.0)
    for i, o in enumerate(objs):
        loss = loss + o.evaluate(pts[i], pts[i + 1], pts[i + 2], xp)
    return loss


def image_path(tx, objs, rx, xp=NUMPY):
    """geometry.py:1013-1114.  Returns (list of points tx..rx, loss)."""
    n = len(objs)
    if n == 0:
        return [tx, rx], xp.c(0.0) * X(rx)  # :1071-1073 (loss 0, broadcast to batch)
    # forward scan of images, :1086-1091, 1109
    images = []
    image = tx
    for o in objs:
        image = wall_image_of(o.xys, image, xp)
        images.append(image)
    # backward scan, :1093-1110
    point = rx
    points = [None] * n
    for k in range(n - 1, -1, -1):
        w = objs[k].xys
        p = wall_origin(w)
        nrm = wall_normal(w, xp)
        u = point - images[k]
        v = p - point
        un = dot(u, nrm)
        vn = dot(v, nrm)
        un_zero = un == xp.c(0.0)
        # jnp.where(un == 0, 0, vn * u / un) -- evaluated left to right, (vn*u)/un, and divided by `un`
        # itself (not by a guarded copy): under autodiff the untaken 0/0 branch poisons the gradient with NaN,
        # exactly as in the reference (the classic `where` trap).
        incx = xp.where(un_zero, xp.c(0.0), xp.div((vn * X(u)), un))
        incy = xp.where(un_zero, xp.c(0.0), xp.div((vn * Y(u)), un))
        point = point + vec(incx, incy, xp)
        points[k] = point
    pts = [tx, *points, rx]
    return pts, path_loss(objs, pts, xp)


def midpoint_path(tx, objs, rx, xp=NUMPY):
    """geometry.py:752-809 (base Path: parametric 0.5 on each object, loss 0)."""
    pts = [o.parametric_to_cartesian(xp.c(0.5), xp) for o in objs]
    return [tx, *pts, rx], xp.c(0.0)


def parametric_to_cartesian(objs, theta, tx, rx, xp=NUMPY):
    """geometry.py:988-1010.  ``theta`` is a list of (...,) arrays, one per unknown."""
    pts = [tx]
    j = 0
    for o in objs:
        size = o.parameters_count()
        if size == 0:
            pts.append(o.parametric_to_cartesian(None, xp))
        else:
            pts.append(o.parametric_to_cartesian(theta[j], xp))
        j += size
    pts.append(rx)
    return pts


# Hyper-parameters of the solvers' Adam (tests of d2d_set_optimizer set them through `adam_hyper`; the reference's default is
# optax.adam(0.1), optimize.py:83).
_ADAM = dict(lr=0.1, b1=0.9, b2=0.999, eps=1e-8)


class adam_hyper:
    """``with adam_hyper(lr=0.05, b1=0.8): ...`` -- opt_path / opt_path_diff inside use these instead of the defaults."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = dict(_ADAM)
        _ADAM.update(self.kw)

    def __exit__(self, *exc):
        _ADAM.clear()
        _ADAM.update(self.old)


def adam_minimize(value_and_grad, x0, steps=100, lr=0.1, b1=0.9, b2=0.999, eps=1e-8, xp=NUMPY):
    """optimize.py:44-97 with optax.adam(0.1) (optax 0.2.4 defaults b1=.9 b2=.999 eps=1e-8,
    eps_root=0): returns (x_final, loss evaluated BEFORE the last update).

    ``x0`` is a list of arrays (one per unknown); ``value_and_grad(x) -> (loss, [grads])``.
    optax.scale_by_adam: mu = b1*mu + (1-b1)*g ; nu = b2*nu + (1-b2)*g*g ;
    mu_hat = mu / (1 - b1**t) ; nu_hat = nu / (1 - b2**t) ; u = -lr * mu_hat / (sqrt(nu_hat) + eps).
    """
    x = list(x0)
    mu = [xp.zeros_like(v) for v in x]
    nu = [xp.zeros_like(v) for v in x]
    loss = None
    for t in range(1, steps + 1):
        loss, g = value_and_grad(x)
        c1 = xp.c(1.0 - b1**t)  # bias corrections are computed in fp32 by optax; see DESIGN.md
        c2 = xp.c(1.0 - b2**t)
        for i in range(len(x)):
            mu[i] = xp.c(b1) * mu[i] + xp.c(1.0 - b1) * g[i]
            nu[i] = xp.c(b2) * nu[i] + xp.c(1.0 - b2) * (g[i] * g[i])
            mh = mu[i] / c1
            nh = nu[i] / c2
            x[i] = x[i] + xp.c(-lr) * (mh / (xp.sqrt(nh) + xp.c(eps)))
    return x, loss


# --- validity (geometry.py:821-963) ---------------------------------------------------


def on_objects(objs, pts, approx, xp=NUMPY, **kw):
    """geometry.py:821-854."""
    contains = true_value(approx, xp)
    for i, o in enumerate(objs):
        contains = logical_and(contains, o.contains(pts[i + 1], approx, xp=xp, **kw), approx, xp=xp)
    return contains


def intersects_with_objects(scene_objs, cand, pts, patch, approx, xp=NUMPY, **kw):
    """geometry.py:856-906."""
    idx = [-1, *[int(c) for c in cand], -1]
    intersects = false_value(approx, xp)
    for i in range(len(pts) - 1):
        for j, o in enumerate(scene_objs):
            ignore = (j == idx[i]) or (j == idx[i + 1])
            if ignore:
                continue  # jnp.where(ignore, intersects, ...) with a concrete (python) `ignore`
            hit = o.intersects(pts[i], pts[i + 1], patch, approx, xp=xp, **kw)
            intersects = logical_or(intersects, hit, approx, xp=xp)
    return intersects


def is_valid(scene_objs, cand, inter_objs, pts, loss, tol=DEFAULT_TOL, patch=DEFAULT_PATCH,
             approx=False, xp=NUMPY, **kw):
    """geometry.py:908-963."""
    on = on_objects(inter_objs, pts, approx, xp=xp, **kw)
    hit = intersects_with_objects(scene_objs, cand, pts, patch, approx, xp=xp, **kw)
    ok = less(loss, xp.c(tol), approx, xp=xp, **kw)
    return xp.nan_to_num(logical_all([on, logical_not(hit, approx, xp=xp), ok], approx, xp=xp))


# --------------------------------------------------------------------------------------
# utils.py:17-54 and other built-in path functions
# --------------------------------------------------------------------------------------


def integer_pow(x, n, xp=NUMPY):
    """jax lax.integer_pow lowering (x**n with python int n >= 0), square-and-multiply."""
    if n == 0:
        return xp.c(1.0)
    acc = None
    while n > 0:
        if n & 1:
            acc = x if acc is None else acc * x
        n >>= 1
        if n > 0:
            x = x * x
    return acc


def received_power(pts, r_coef=DEFAULT_R_COEF, height=DEFAULT_HEIGHT, xp=NUMPY):
    """utils.py:17-54.  r_coef / height enter the jitted function as fp32 scalars."""
    r = path_length(pts, xp)
    n = len(pts) - 2
    h = xp.c(height)
    return integer_pow(xp.c(r_coef), n, xp) / (h * h + r * r)


def length_squared(pts, xp=NUMPY):
    """tests/test_scene.py:444-445: ``path.length() ** 2``."""
    r = path_length(pts, xp)
    return r * r


FUNS = {
    "received_power": received_power,
    "length_squared": length_squared,
    "length": lambda pts, xp=NUMPY: path_length(pts, xp),
    "one": lambda pts, xp=NUMPY: xp.c(1.0) + xp.c(0.0) * X(pts[-1]),
}

# --------------------------------------------------------------------------------------
# scene.py
# --------------------------------------------------------------------------------------


def all_path_candidates(num_nodes, min_order=0, max_order=1, order=None, filter_nodes=None):
    """scene.py:122-175.  differt-core 0.0.31 (Rust, not in the tree) enumerates, for each
    order k ascending, every node sequence of length k with no two consecutive equal nodes,
    lexicographically (recorded order: docs/source/notebooks/cost20120_helsinki_model.ipynb:607-648);
    ``filter_nodes`` are disconnected, i.e. never visited (tests/test_scene.py:381-399)."""
    if order is not None:
        min_order = max_order = order
    allowed = [i for i in range(num_nodes) if not (filter_nodes and i in filter_nodes)]
    out = []

    def rec(prefix, k):
        if k == 0:
            out.append(np.asarray(prefix, dtype=np.int32))
            return
        for w in allowed:
            if prefix and prefix[-1] == w:
                continue
            rec(prefix + [w], k - 1)

    for k in range(min_order, max_order + 1):
        rec([], k)
    return out


def square_scene_walls():
    """scene.py:829-834."""
    return np.array(
        [[[0, 0], [1, 0]], [[1, 0], [1, 1]], [[1, 1], [0, 1]], [[0, 1], [0, 0]]], dtype=np.float32
    )


def square_scene_with_wall_walls(ratio=0.6):
    """scene.py:878-882."""
    extra = np.array([[[0.5, 0.5 * (1 - ratio)], [0.5, 0.5 * (1 + ratio)]]], dtype=np.float32)
    return np.concatenate([square_scene_walls(), extra])


def square_scene_with_obstacle_walls(ratio=0.1):
    """scene.py:923-935."""
    hl = 0.5 * ratio
    x0, x1 = 0.5 - hl, 0.5 + hl
    y0, y1 = 0.5 - hl, 0.5 + hl
    extra = np.array(
        [[[x0, y0], [x1, y0]], [[x1, y0], [x1, y1]], [[x1, y1], [x0, y1]], [[x0, y1], [x0, y0]]],
        dtype=np.float32,
    )
    return np.concatenate([square_scene_walls(), extra])


def basic_scene_walls():
    """scene.py:775-785."""
    return np.array(
        [
            [[0.0, 0.0], [1.0, 0.0]],
            [[1.0, 0.0], [1.0, 1.0]],
            [[1.0, 1.0], [0.0, 1.0]],
            [[0.0, 1.0], [0.0, 0.0]],
            [[0.4, 0.0], [0.4, 0.4]],
            [[0.4, 0.4], [0.3, 0.4]],
            [[0.1, 0.4], [0.0, 0.4]],
        ],
        dtype=np.float32,
    )


def solve_path(solver, tx, inter_objs, rx, xp=NUMPY, theta0=None, steps=100):
    """Dispatch on path class (scene.py:1896-1902)."""
    if solver == "image":
        return image_path(tx, inter_objs, rx, xp)
    if solver in ("min", "fermat"):
        return opt_path(solver, tx, inter_objs, rx, theta0, steps, xp)
    raise ValueError(solver)


def opt_path(solver, tx, objs, rx, theta0, steps, xp=NUMPY):
    """geometry.py:1117-1204 (FermatPath) and :1207-1288 (MinPath) with an explicit theta0
    (the reference draws theta0 ~ U[0,1) from a per-candidate Threefry key shared by all RX,
    scene.py:1887-1890, optimize.py:132; jax.random is unavailable so theta0 is an input).
    Gradients of the loss w.r.t. theta come from torch autograd (fp32) -- the reference uses
    jax.value_and_grad (optimize.py:85)."""
    import torch

    n = len(objs)
    if n == 0:
        return [tx, rx], xp.c(0.0) * X(rx)
    if getattr(xp, "diff_solver", False):
        return opt_path_diff(solver, tx, objs, rx, theta0, steps, xp)
    tb = TorchBackend("float32" if getattr(xp, "dtype", None) == np.float32 or xp.name == "torch" else "float64")
    n_unknowns = sum(o.parameters_count() for o in objs)
    batch = np.broadcast_shapes(np.shape(X(tx)), np.shape(X(rx)))
    t_tx, t_rx = tb.asarray(_np(tx)), tb.asarray(_np(rx))
    t_objs = [Obj(o.kind, tb.asarray(_np(o.xys)), o.phi) for o in objs]

    def loss_of(pts):
        if solver == "fermat":
            return path_length(pts, tb)
        return path_loss(t_objs, pts, tb)

    def vg(x):
        xs = [v.detach().clone().requires_grad_(True) for v in x]
        pts = parametric_to_cartesian(t_objs, xs, t_tx, t_rx, tb)
        loss = loss_of(pts)
        loss = loss + 0 * sum(xs) if n_unknowns else loss
        g = torch.autograd.grad(loss.sum(), xs) if n_unknowns else []
        return loss.detach(), [gi.detach() for gi in g]

    x0 = [torch.full(batch, float(theta0[i]), dtype=tb.tdtype) for i in range(n_unknowns)]
    x, last_loss = adam_minimize(vg, x0, steps=steps, xp=tb, **_ADAM)
    pts = parametric_to_cartesian(t_objs, x, t_tx, t_rx, tb)
    if solver == "fermat":
        loss = path_loss(t_objs, pts, tb)  # geometry.py:1204
    else:
        loss = last_loss  # geometry.py:1284-1288 (loss before the last update)
    pts = [xp.asarray(p.detach().numpy()) if xp.name == "numpy" else p for p in pts]
    pts = [_bcast(p, batch, xp) for p in pts]
    loss = xp.asarray(loss.detach().numpy()) if xp.name == "numpy" else loss
    return pts, loss


def opt_path_diff(solver, tx, objs, rx, theta0, steps, tb):
    """opt_path with the autograd graph kept through every Adam step: jax.value_and_grad inside the scan becomes
    torch.autograd.grad(..., create_graph=True), so that reverse mode through the whole solve gives what the reference's
    reverse mode through lax.scan gives (optimize.py:83-97).  ``tx`` / ``rx`` / ``objs`` hold torch tensors."""
    import torch

    n_unknowns = sum(o.parameters_count() for o in objs)
    batch = torch.broadcast_shapes(X(tx).shape, X(rx).shape)

    def loss_of(pts):
        return path_length(pts, tb) if solver == "fermat" else path_loss(objs, pts, tb)

    def vg(x):
        pts = parametric_to_cartesian(objs, x, tx, rx, tb)
        loss = loss_of(pts)
        loss = loss + 0 * sum(x) if n_unknowns else loss
        g = torch.autograd.grad(loss.sum(), x, create_graph=True) if n_unknowns else []
        return loss, list(g)

    x0 = [torch.full(tuple(batch), float(theta0[i]), dtype=tb.tdtype).requires_grad_(True) for i in range(n_unknowns)]
    x, last_loss = adam_minimize(vg, x0, steps=steps, xp=tb, **_ADAM)
    pts = parametric_to_cartesian(objs, x, tx, rx, tb)
    loss = path_loss(objs, pts, tb) if solver == "fermat" else last_loss  # geometry.py:1204 / :1284-1288
    return [p.expand(*batch, 2) for p in pts], loss


def _np(x):
    return x.detach().numpy() if hasattr(x, "detach") else np.asarray(x)


def _bcast(p, batch, xp):
    if xp.name == "numpy":
        return np.broadcast_to(p, (*batch, 2)).astype(p.dtype, copy=False)
    return p.expand(*batch, 2)


def accumulate_candidate(tx, scene_objs, cand, rx, fun="received_power", fun_kwargs=None,
                         solver="image", approx=False, xp=NUMPY, theta0=None, steps=100, **kw):
    """One iteration of the loop body of scene.py:1894-1916: returns (valid, fun value)."""
    fun_kwargs = fun_kwargs or {}
    inter = [scene_objs[int(i)] for i in cand]  # scene.py:1136-1154
    pts, loss = solve_path(solver, tx, inter, rx, xp, theta0=theta0, steps=steps)
    valid = is_valid(scene_objs, cand, inter, pts, loss, approx=approx, xp=xp, **kw)
    f = FUNS[fun] if isinstance(fun, str) else fun
    return valid, f(pts, xp=xp, **fun_kwargs), pts, loss


def facc(tx, scene_objs, cands, rx, fun="received_power", fun_kwargs=None, solver="image",
         approx=False, xp=NUMPY, theta0s=None, steps=100, **kw):
    """scene.py:1892-1918: acc = 0; for cand: acc = acc + valid * fun  (sequential fp32 sum)."""
    acc = xp.c(0.0) * X(rx)
    for ci, cand in enumerate(cands):
        th = None if theta0s is None else theta0s[ci]
        valid, val, _, _ = accumulate_candidate(
            tx, scene_objs, cand, rx, fun, fun_kwargs, solver, approx, xp, theta0=th, steps=steps, **kw
        )
        acc = acc + xp.to_float(valid) * val
    return acc


def power_map(walls, tx, Xg, Yg, min_order=0, max_order=1, order=None, fun="received_power",
              fun_kwargs=None, approx=False, objs=None, filter_nodes=None, solver="image",
              theta0s=None, steps=100, xp=NUMPY, grid_role="rx", **kw):
    """Scene.accumulate_on_receivers_grid_over_paths for ONE transmitter (scene.py:1803-1953):
    grid = dstack((X, Y)); Z = vmap(vmap(facc))(tx, grid).  With grid_role="tx" the grid cells are the
    transmitters and ``tx`` is the fixed receiver (accumulate_on_transmitters_grid_over_paths, scene.py:1489-1648)."""
    scene_objs = objs if objs is not None else walls_to_objs(walls, xp)
    cands = all_path_candidates(len(scene_objs), min_order, max_order, order, filter_nodes)
    grid = vec(xp.asarray(Xg), xp.asarray(Yg), xp)
    fixed = xp.asarray(tx)
    a, b = (fixed, grid) if grid_role == "rx" else (grid, fixed)
    return facc(a, scene_objs, cands, b, fun, fun_kwargs, solver, approx, xp, theta0s=theta0s, steps=steps, **kw)


def grid(bbox, m=50, n=None):
    """abc.py:57-81: x = linspace(xmin, xmax, m), y = linspace(ymin, ymax, n); meshgrid 'xy'."""
    if n is None:
        n = m
    x = np.linspace(bbox[0][0], bbox[1][0], m).astype(np.float32)
    y = np.linspace(bbox[0][1], bbox[1][1], n).astype(np.float32)
    return np.meshgrid(x, y)


# --------------------------------------------------------------------------------------
# Reverse-mode gradients through the torch backend (scene.py:1920-1925 and user-side
# jax.value_and_grad over scene parameters, examples/plot_power_optimize.py:78-93)
# --------------------------------------------------------------------------------------


def power_map_value_and_grads(walls, tx, Xg, Yg, cotangent=None, dtype="float32", **kwargs):
    """Returns dict(value[m,n], grad_rx[m,n,2], tx_bar[2], walls_bar[N,2,2]).

    grad_rx is the per-cell gradient (each cell's facc depends on its own rx only, so one
    backward pass of sum(facc) yields it); tx_bar / walls_bar are the VJP with ``cotangent``
    (default ones, i.e. the gradient of sum(Z))."""
    import torch

    tb = TorchBackend(dtype)
    w = tb.asarray(np.asarray(walls)).clone().requires_grad_(True)
    t = tb.asarray(np.asarray(tx)).clone().requires_grad_(True)
    gx = tb.asarray(np.asarray(Xg)).clone().requires_grad_(True)
    gy = tb.asarray(np.asarray(Yg)).clone().requires_grad_(True)
    Z = power_map(w, t, gx, gy, xp=tb, **kwargs)
    ct = torch.ones_like(Z) if cotangent is None else tb.asarray(np.asarray(cotangent))
    gw, gt = torch.autograd.grad((Z * ct).sum(), [w, t], retain_graph=True, allow_unused=True)
    ggx, ggy = torch.autograd.grad(Z.sum(), [gx, gy], allow_unused=True)
    z = lambda g, ref: (torch.zeros_like(ref) if g is None else g).detach().numpy()
    return {
        "value": Z.detach().numpy(),
        "grad_rx": np.stack([z(ggx, gx), z(ggy, gy)], axis=-1),
        "tx_bar": z(gt, t),
        "walls_bar": z(gw, w),
    }


# --------------------------------------------------------------------------------------
# The same sweep with the candidates of one order evaluated side by side (a leading candidate
# axis that broadcasts through every function above, exactly as the cells' axis does).  Same op
# chain per (cell, candidate); only the Python loop over candidates is gone, which makes reverse
# mode over scenes of 50+ walls tractable (tests/golden/cfg3_grad_*.npz).  tests/test_oracle_batched.py
# checks it against the candidate-by-candidate functions above.
# --------------------------------------------------------------------------------------


def intersects_with_objects_batched(scene_objs, cands, pts, patch, approx, xp=NUMPY, **kw):
    """geometry.py:856-906 for a batch of candidates ``cands[Ck, k]``: ``ignore`` is now an array, applied with
    ``where(ignore, intersects, or(intersects, hit))`` exactly as the reference writes it (:892-904)."""
    Ck, k = cands.shape
    idx = np.concatenate([np.full((Ck, 1), -1), cands, np.full((Ck, 1), -1)], axis=1)
    intersects = false_value(approx, xp)
    for i in range(k + 1):
        for j, o in enumerate(scene_objs):
            ignore = (idx[:, i] == j) | (idx[:, i + 1] == j)
            if ignore.all():
                continue  # where(True, intersects, ...) == intersects
            hit = o.intersects(pts[i], pts[i + 1], patch, approx, xp=xp, **kw)
            new = logical_or(intersects, hit, approx, xp=xp)
            if not ignore.any():
                intersects = new
            elif xp.name == "torch":
                intersects = xp.torch.where(xp.torch.as_tensor(ignore[:, None]), intersects, new)
            else:
                intersects = np.where(ignore[:, None], intersects, new)  # keeps bool (hard) / float (approx) dtype
    return intersects


def facc_batched(tx, walls, cands_by_order, rx, fun="received_power", fun_kwargs=None, approx=False, xp=NUMPY,
                 grid_role="rx", sequential_sum=True, **kw):
    """scene.py:1892-1918 with the candidates of each order side by side.  ``walls``: (N, 2, 2) array / tensor;
    ``cands_by_order``: list of (k, int array [Ck, k]); ``rx``: (cells, 2) -- the grid cells (receivers, or transmitters
    with grid_role="tx"); ``tx``: (2,) the fixed end point.  The contributions are added in candidate order (fp32 sum)."""
    fun_kwargs = fun_kwargs or {}
    patch = kw.pop("patch", DEFAULT_PATCH)
    tol = kw.pop("tol", DEFAULT_TOL)
    f = FUNS[fun] if isinstance(fun, str) else fun
    N = walls.shape[0]
    scene_objs = [Obj(WALL, walls[j]) for j in range(N)]
    acc = xp.c(0.0) * X(rx)
    for k, cands in cands_by_order:
        if k == 0:
            a, b = (tx, rx) if grid_role == "rx" else (rx, tx)
            valid, val, _, _ = accumulate_candidate(a, scene_objs, cands[0], b, fun, fun_kwargs, "image", approx, xp,
                                                    patch=patch, tol=tol, **kw)
            acc = acc + xp.to_float(valid) * val
            continue
        inter = [Obj(WALL, walls[cands[:, i]][:, None]) for i in range(k)]  # xys: (Ck, 1, 2, 2)
        a, b = (tx, rx) if grid_role == "rx" else (rx, tx)
        pts, loss = image_path(a, inter, b, xp)
        on = on_objects(inter, pts, approx, xp=xp, **kw)
        hit = intersects_with_objects_batched(scene_objs, cands, pts, patch, approx, xp=xp, **kw)
        ok = less(loss, xp.c(tol), approx, xp=xp, **kw)
        valid = xp.nan_to_num(logical_all([on, logical_not(hit, approx, xp=xp), ok], approx, xp=xp))
        contrib = xp.to_float(valid) * f(pts, xp=xp, **fun_kwargs)  # (Ck, cells)
        if sequential_sum:
            for c in range(contrib.shape[0]):
                acc = acc + contrib[c]
        else:
            acc = acc + contrib.sum(0)
    return acc


def candidates_by_order(num_nodes, min_order=0, max_order=1, order=None, filter_nodes=None):
    if order is not None:
        min_order = max_order = order
    out = []
    for k in range(min_order, max_order + 1):
        c = all_path_candidates(num_nodes, k, k, None, filter_nodes)
        out.append((k, np.stack(c).astype(np.int64).reshape(len(c), k) if c else np.zeros((0, k), np.int64)))
    return [(k, c) for k, c in out if len(c)]


def power_map_batched(walls, tx, Xg, Yg, min_order=0, max_order=1, order=None, filter_nodes=None, xp=NUMPY, **kw):
    """power_map() through facc_batched: same values bit for bit (fp32 NumPy backend)."""
    w = xp.asarray(walls)
    shape = np.shape(Xg)
    cells = vec(xp.asarray(Xg).reshape(-1), xp.asarray(Yg).reshape(-1), xp)
    cb = candidates_by_order(w.shape[0], min_order, max_order, order, filter_nodes)
    return facc_batched(xp.asarray(tx), w, cb, cells, xp=xp, **kw).reshape(shape)


def power_map_value_and_grads_batched(walls, tx, Xg, Yg, cotangent=None, dtype="float32", chunk=128, min_order=0,
                                      max_order=1, order=None, filter_nodes=None, **kwargs):
    """power_map_value_and_grads() with the candidates side by side and the cells in chunks of ``chunk`` (bounds the
    autograd tape: ~3750 saved tensors of Ck x chunk elements at 50 walls, order 2).  Same dict of results."""
    import torch

    tb = TorchBackend(dtype)
    shape = np.shape(Xg)
    Xf, Yf = np.asarray(Xg).reshape(-1), np.asarray(Yg).reshape(-1)
    ct_all = np.ones(Xf.shape, np.float64) if cotangent is None else np.asarray(cotangent, np.float64).reshape(-1)
    walls = np.asarray(walls)
    cb = candidates_by_order(walls.shape[0], min_order, max_order, order, filter_nodes)
    value = np.zeros(Xf.shape, np.float64)
    grad = np.zeros((*Xf.shape, 2), np.float64)
    tx_bar = np.zeros(2, np.float64)
    walls_bar = np.zeros(walls.shape, np.float64)
    for lo in range(0, Xf.size, chunk):
        sl = slice(lo, min(lo + chunk, Xf.size))
        w = tb.asarray(walls).clone().requires_grad_(True)
        t = tb.asarray(np.asarray(tx)).clone().requires_grad_(True)
        cells = tb.asarray(np.stack([Xf[sl], Yf[sl]], -1)).clone().requires_grad_(True)
        Z = facc_batched(t, w, cb, cells, xp=tb, sequential_sum=False, **kwargs)
        ct = tb.asarray(ct_all[sl])
        gw, gt = torch.autograd.grad((Z * ct).sum(), [w, t], retain_graph=True, allow_unused=True)
        (gc,) = torch.autograd.grad(Z.sum(), [cells], allow_unused=True)
        value[sl] = Z.detach().numpy()
        if gc is not None:
            grad[sl] = gc.detach().numpy()
        if gt is not None:
            tx_bar += gt.detach().numpy()
        if gw is not None:
            walls_bar += gw.detach().numpy()
    return {"value": value.reshape(shape), "grad_rx": grad.reshape(*shape, 2), "tx_bar": tx_bar, "walls_bar": walls_bar}


def opt_value_and_grads(kinds, xys, phis, tx, Xg, Yg, cands, theta0s, solver="min", steps=100, cotangent=None,
                        dtype="float64", grid_role="rx", **kwargs):
    """Value map and reverse-mode gradients of a MinPath / FermatPath sweep over a scene of Wall / RIS / Vertex objects
    (BASELINE.json configs[4]), differentiating THROUGH the Adam loop like the reference does (optimize.py:83-97).

    ``kinds[N]`` (WALL / RIS / VERTEX), ``xys[N, 2, 2]`` (a Vertex keeps its point in row 0), ``phis[N]``; ``cands``: list of
    index arrays; ``theta0s``: one list of initial guesses per candidate.  Returns dict(value[m, n], grad_cell[m, n, 2],
    fixed_bar[2], xys_bar[N, 2, 2], phi_bar[N]) -- the last three contracted with ``cotangent`` (default ones)."""
    import torch

    tb = TorchBackend(dtype, diff_solver=True)
    w = tb.asarray(np.asarray(xys)).clone().requires_grad_(True)
    ph = tb.asarray(np.asarray(phis)).clone().requires_grad_(True)
    t = tb.asarray(np.asarray(tx)).clone().requires_grad_(True)
    gx = tb.asarray(np.asarray(Xg)).clone().requires_grad_(True)
    gy = tb.asarray(np.asarray(Yg)).clone().requires_grad_(True)
    objs = [Obj(int(k), w[j, 0] if int(k) == VERTEX else w[j], ph[j]) for j, k in enumerate(kinds)]
    grid = vec(gx, gy, tb)
    a, b = (t, grid) if grid_role == "rx" else (grid, t)
    Z = facc(a, objs, cands, b, solver=solver, xp=tb, theta0s=theta0s, steps=steps, **kwargs)
    ct = torch.ones_like(Z) if cotangent is None else tb.asarray(np.asarray(cotangent))
    gw, gp, gt = torch.autograd.grad((Z * ct).sum(), [w, ph, t], retain_graph=True, allow_unused=True)
    ggx, ggy = torch.autograd.grad(Z.sum(), [gx, gy], allow_unused=True)
    z = lambda g, ref: (torch.zeros_like(ref) if g is None else g).detach().numpy()
    return {"value": Z.detach().numpy(), "grad_cell": np.stack([z(ggx, gx), z(ggy, gy)], axis=-1), "fixed_bar": z(gt, t),
            "xys_bar": z(gw, w), "phi_bar": z(gp, ph)}
