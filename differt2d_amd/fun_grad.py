"""Derivatives of a user-supplied path function ``fun(tx, rx, path, interacting_objects, *args, **kwargs)`` on the host.

The reference differentiates whatever JAX callable it is given (scene.py:1892-1923: ``jax.grad`` of
``sum_c valid_c * fun_c`` w.r.t. the grid cell).  Here the validity and the image method are differentiated on the GPU by the
hand-derived adjoint (``d2d_power_map_vg_launch`` with ``D2D_FUN_CUSTOM``, include/d2d.h); what the kernel needs from the host
is ``fun`` itself and its derivative w.r.t. the path's points, per (candidate, cell).  Two ways to obtain them:

* ``fun.value_and_grad(tx, rx, path, interacting_objects, *args, **kwargs)`` -- supplied by the user -- returning
  ``(value, d value / d path.xys)`` or ``(value, d value / d path.xys, d value / d tx.xy, d value / d rx.xy)``;
* otherwise ``fun`` is called on :class:`TapeArray` operands -- NumPy arrays that record the operations applied to them
  (arithmetic operators, ``path.length()``, NumPy ufuncs and the common NumPy functions: ``np.sqrt``, ``np.sum``,
  ``np.where``, ``np.linalg.norm`` ..., indexing, array methods) -- and the derivative is read off that record in reverse
  order with JAX's conventions (``minimum`` / ``maximum`` split a tie evenly, ``abs'(0) = 0``, ``sqrt'(0) = inf``, ``where``
  passes nothing to the branch not taken).  A few dozen lines of NumPy: no PyTorch, no JAX, nothing imported on demand; it
  sees the user's function only, batched over (cells) once per candidate.  An operation the record does not know raises,
  and the sweep is refused (``D2DUnsupported``) rather than differentiated wrongly.
"""

from __future__ import annotations

import numpy as np

from . import _lib as L

F = np.float32
EPS = float(np.finfo(np.float32).eps)


# --------------------------------------------------------------------------------------------------------------- the tape
def _unbroadcast(g, shape):
    """Sum ``g`` down to ``shape`` (the adjoint of NumPy broadcasting)."""
    g = np.asarray(g)
    if g.shape == tuple(shape):
        return g
    while g.ndim > len(shape):
        g = g.sum(axis=0)
    for ax, n in enumerate(shape):
        if n == 1 and g.shape[ax] != 1:
            g = g.sum(axis=ax, keepdims=True)
    return g


class TapeError(TypeError):
    """An operation the tape cannot differentiate."""


class TapeArray:
    """A NumPy array that records how it was computed: ``value`` and, per operand that is itself recorded, the function that
    maps this node's cotangent to the operand's."""

    __array_priority__ = 1000.0
    __slots__ = ("value", "parents")

    def __init__(self, value, parents=()):
        self.value = np.asarray(value)
        self.parents = parents  # tuple of (TapeArray, cotangent -> cotangent of that parent)

    # -- array protocol
    shape = property(lambda self: self.value.shape)
    ndim = property(lambda self: self.value.ndim)
    dtype = property(lambda self: self.value.dtype)
    size = property(lambda self: self.value.size)

    def __len__(self):
        return len(self.value)

    def __array__(self, *a, **k):
        raise TapeError("a recording array was converted to a plain NumPy array (np.asarray / float() / an unsupported function): "
                        "the derivative would be lost")

    def __float__(self):
        raise TapeError("float() of a recording array: the derivative would be lost")

    __bool__ = __int__ = __float__

    def __repr__(self):
        return f"TapeArray({self.value!r})"

    # -- operators -> ufuncs
    def __add__(self, o): return np.add(self, o)
    def __radd__(self, o): return np.add(o, self)
    def __sub__(self, o): return np.subtract(self, o)
    def __rsub__(self, o): return np.subtract(o, self)
    def __mul__(self, o): return np.multiply(self, o)
    def __rmul__(self, o): return np.multiply(o, self)
    def __truediv__(self, o): return np.true_divide(self, o)
    def __rtruediv__(self, o): return np.true_divide(o, self)
    def __pow__(self, o): return np.power(self, o)
    def __rpow__(self, o): return np.power(o, self)
    def __neg__(self): return np.negative(self)
    def __pos__(self): return self
    def __abs__(self): return np.absolute(self)
    def __lt__(self, o): return self.value < _val(o)
    def __le__(self, o): return self.value <= _val(o)
    def __gt__(self, o): return self.value > _val(o)
    def __ge__(self, o): return self.value >= _val(o)
    def __eq__(self, o): return self.value == _val(o)  # noqa: PLR0124
    def __ne__(self, o): return self.value != _val(o)
    __hash__ = None

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        if method != "__call__" or kwargs.get("out") is not None:
            raise TapeError(f"np.{ufunc.__name__}.{method} on a recording array")
        rule = _UFUNCS.get(ufunc)
        vals = [_val(x) for x in inputs]
        if rule is None:
            if ufunc in _PLAIN_UFUNCS:  # comparisons, isfinite, sign ...: piecewise constant, plain arrays out
                return ufunc(*vals, **kwargs)
            raise TapeError(f"np.{ufunc.__name__} has no derivative rule on the tape")
        out = ufunc(*vals, **kwargs)
        parents = []
        for i, x in enumerate(inputs):
            if isinstance(x, TapeArray):
                parents.append((x, _bind(rule, i, vals, out, x.shape)))
        return TapeArray(out, tuple(parents))

    def __array_function__(self, func, types, args, kwargs):
        impl = _FUNCTIONS.get(func)
        if impl is None:
            raise TapeError(f"np.{getattr(func, '__name__', func)} has no derivative rule on the tape")
        return impl(*args, **kwargs)

    # -- indexing, shapes, methods (NumPy's and, for functions written against tensors, the common tensor spellings)
    def __getitem__(self, idx):
        out = self.value[idx]
        shape, dtype = self.shape, self.dtype

        def back(g):
            z = np.zeros(shape, dtype if dtype.kind == "f" else np.float64)
            np.add.at(z, idx, g)
            return z

        return TapeArray(out, ((self, back),))

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        old = self.shape
        return TapeArray(self.value.reshape(shape), ((self, lambda g: np.reshape(g, old)),))

    def astype(self, dtype, **_):
        if np.dtype(dtype).kind != "f":
            raise TapeError("astype to a non-float type: the derivative would be lost")
        return TapeArray(self.value.astype(dtype), ((self, lambda g: g),))

    @property
    def T(self):
        return TapeArray(self.value.T, ((self, lambda g: np.asarray(g).T),))

    def sum(self, axis=None, keepdims=False, **_):
        return _sum(self, axis=axis, keepdims=keepdims)

    def mean(self, axis=None, keepdims=False, **_):
        return _mean(self, axis=axis, keepdims=keepdims)

    def sqrt(self): return np.sqrt(self)
    def exp(self): return np.exp(self)
    def log(self): return np.log(self)
    def sin(self): return np.sin(self)
    def cos(self): return np.cos(self)
    def tanh(self): return np.tanh(self)
    def abs(self): return np.absolute(self)
    def square(self): return np.square(self)
    def pow(self, o): return np.power(self, o)
    def clip(self, a_min=None, a_max=None, **_): return _clip(self, a_min, a_max)


def _val(x):
    return x.value if isinstance(x, TapeArray) else x


def _bind(rule, i, vals, out, shape):
    return lambda g: _unbroadcast(rule(i, g, vals, out), shape)


def _tie_half(sel, tie):
    """JAX's rule for minimum / maximum: the selected argument takes the cotangent, a tie splits it evenly."""
    return np.where(tie, 0.5, np.where(sel, 1.0, 0.0))


def _pow_rule(i, g, v, out):
    a, b = v
    with np.errstate(all="ignore"):
        if i == 0:  # lax.pow / integer_pow: b a^(b-1), and 0 for b == 0 whatever a
            return g * np.where(np.asarray(b) == 0, 0.0, b * np.power(a, np.asarray(b) - 1))
        return g * out * np.log(np.where(np.asarray(a) == 0, 1.0, a))  # (0 at a == 0)


_UFUNCS = {
    np.add: lambda i, g, v, out: g,
    np.subtract: lambda i, g, v, out: g if i == 0 else -g,
    np.multiply: lambda i, g, v, out: g * v[1 - i],
    np.true_divide: lambda i, g, v, out: g / v[1] if i == 0 else -g * out / v[1],
    np.negative: lambda i, g, v, out: -g,
    np.positive: lambda i, g, v, out: g,
    np.power: _pow_rule,
    np.square: lambda i, g, v, out: g * 2 * v[0],
    np.sqrt: lambda i, g, v, out: _quiet(lambda: g * (0.5 / out)),  # sqrt'(0) = inf, as jnp.sqrt's rule has it
    np.reciprocal: lambda i, g, v, out: -g * out * out,
    np.exp: lambda i, g, v, out: g * out,
    np.expm1: lambda i, g, v, out: g * (out + 1),
    np.log: lambda i, g, v, out: _quiet(lambda: g / v[0]),
    np.log1p: lambda i, g, v, out: _quiet(lambda: g / (1 + v[0])),
    np.log2: lambda i, g, v, out: _quiet(lambda: g / (v[0] * np.log(2.0))),
    np.log10: lambda i, g, v, out: _quiet(lambda: g / (v[0] * np.log(10.0))),
    np.sin: lambda i, g, v, out: g * np.cos(v[0]),
    np.cos: lambda i, g, v, out: -g * np.sin(v[0]),
    np.tan: lambda i, g, v, out: g * (1 + out * out),
    np.tanh: lambda i, g, v, out: g * (1 - out * out),
    np.arctan: lambda i, g, v, out: g / (1 + np.square(v[0])),
    np.arctan2: lambda i, g, v, out: _quiet(lambda: g * (v[1] if i == 0 else -v[0]) / (np.square(v[0]) + np.square(v[1]))),
    np.hypot: lambda i, g, v, out: _quiet(lambda: g * v[i] / out),
    np.absolute: lambda i, g, v, out: g * np.sign(v[0]),  # jnp.abs: sign(0) = 0
    np.minimum: lambda i, g, v, out: g * _tie_half(v[i] < v[1 - i], v[0] == v[1]),
    np.maximum: lambda i, g, v, out: g * _tie_half(v[i] > v[1 - i], v[0] == v[1]),
}
_UFUNCS[np.divide] = _UFUNCS[np.true_divide]
_UFUNCS[np.abs] = _UFUNCS[np.absolute]
_PLAIN_UFUNCS = {np.greater, np.greater_equal, np.less, np.less_equal, np.equal, np.not_equal, np.isfinite, np.isnan, np.isinf,
                 np.sign, np.signbit, np.floor, np.ceil, np.rint, np.trunc}


def _quiet(f):
    with np.errstate(all="ignore"):
        return f()


def _axes(axis, ndim):
    if axis is None:
        return tuple(range(ndim))
    return tuple(a % ndim for a in (axis if isinstance(axis, (tuple, list)) else (axis,)))


def _sum(x, axis=None, keepdims=False, **kw):
    if kw.get("out") is not None or kw.get("where", True) is not True:
        raise TapeError("np.sum(out= / where=) on a recording array")
    x = _as_tape(x)
    shape, axes = x.shape, _axes(axis, x.ndim)
    out = x.value.sum(axis=axis, keepdims=keepdims, dtype=kw.get("dtype"))

    def back(g):
        g = np.asarray(g)
        if not keepdims:
            g = np.expand_dims(g, axes) if axes else g
        return np.broadcast_to(g, shape)

    return TapeArray(out, ((x, back),))


def _mean(x, axis=None, keepdims=False, **kw):
    x = _as_tape(x)
    n = int(np.prod([x.shape[a] for a in _axes(axis, x.ndim)])) if x.ndim else 1
    return _sum(x, axis=axis, keepdims=keepdims, **kw) / x.dtype.type(n)


def _as_tape(x):
    return x if isinstance(x, TapeArray) else TapeArray(np.asarray(x))


def _where(cond, a, b):
    cond = np.asarray(_val(cond), bool)
    out = np.where(cond, _val(a), _val(b))
    parents = []
    for x, keep in ((a, cond), (b, ~cond)):
        if isinstance(x, TapeArray):
            parents.append((x, (lambda keep, shape: lambda g: _unbroadcast(np.where(keep, g, 0.0), shape))(keep, x.shape)))
    return TapeArray(out, tuple(parents))


def _stack_like(np_func):
    def impl(arrays, axis=0, **kw):
        if kw.get("out") is not None:
            raise TapeError(f"np.{np_func.__name__}(out=) on a recording array")
        arrays = list(arrays)
        vals = [np.asarray(_val(a)) for a in arrays]
        out = np_func(vals, axis=axis)
        parents = []
        if np_func is np.stack:
            for i, a in enumerate(arrays):
                if isinstance(a, TapeArray):
                    parents.append((a, (lambda i: lambda g: np.take(g, i, axis=axis))(i)))
        else:
            ax = axis % out.ndim
            off = 0
            for a, v in zip(arrays, vals):
                n = v.shape[ax]
                if isinstance(a, TapeArray):
                    parents.append((a, (lambda lo, hi: lambda g: np.take(g, np.arange(lo, hi), axis=ax))(off, off + n)))
                off += n
        return TapeArray(out, tuple(parents))

    return impl


def _norm(x, ord=None, axis=None, keepdims=False):  # noqa: A002
    if ord not in (None, 2, "fro"):
        raise TapeError("np.linalg.norm: only the 2-norm has a rule on the tape")
    return np.sqrt(_sum(_as_tape(x) * x, axis=axis, keepdims=keepdims))


def _clip(x, a_min=None, a_max=None, **kw):
    if a_min is not None:
        x = np.maximum(x, a_min)
    if a_max is not None:
        x = np.minimum(x, a_max)
    return x


def _dot_last(a, b):
    return _sum(_as_tape(a) * b, axis=-1)


_FUNCTIONS = {
    np.sum: _sum,
    np.mean: _mean,
    np.where: _where,
    np.stack: _stack_like(np.stack),
    np.concatenate: _stack_like(np.concatenate),
    np.linalg.norm: _norm,
    np.clip: _clip,
    np.reshape: lambda x, *shape, **kw: _as_tape(x).reshape(*shape, **kw),
    np.squeeze: lambda x, axis=None: (lambda x: TapeArray(np.squeeze(x.value, axis), ((x, lambda g: np.reshape(g, x.shape)),)))(_as_tape(x)),
    np.expand_dims: lambda x, axis: (lambda x: TapeArray(np.expand_dims(x.value, axis), ((x, lambda g: np.reshape(g, x.shape)),)))(_as_tape(x)),
    np.broadcast_to: lambda x, shape, **kw: (lambda x: TapeArray(np.broadcast_to(x.value, shape), ((x, lambda g: _unbroadcast(g, x.shape)),)))(_as_tape(x)),
    np.shape: lambda x: _val(x).shape,
    np.ndim: lambda x: _val(x).ndim,
    np.size: lambda x, axis=None: np.size(_val(x), axis),
    np.zeros_like: lambda x, **kw: np.zeros_like(_val(x), **kw),
    np.ones_like: lambda x, **kw: np.ones_like(_val(x), **kw),
    np.diff: lambda x, n=1, axis=-1: _diff(x, n, axis),
    np.dot: lambda a, b: _dot_1d(a, b),
    np.vdot: lambda a, b: _dot_1d(a, b),
}


def _dot_1d(a, b):
    """``np.dot`` of scalars / vectors (what an objective of ``optimize.minimize`` uses: ``jnp.dot(x, x)``)."""
    if np.ndim(_val(a)) > 1 or np.ndim(_val(b)) > 1:
        raise TapeError("np.dot of matrices is not recorded; write it with * and sum")
    return _sum(_as_tape(a) * b)


def _diff(x, n, axis):
    x = _as_tape(x)
    for _ in range(n):
        hi = [slice(None)] * x.ndim
        lo = [slice(None)] * x.ndim
        hi[axis], lo[axis] = slice(1, None), slice(None, -1)
        x = x[tuple(hi)] - x[tuple(lo)]
    return x


def backward(out: TapeArray, leaves):
    """Cotangents of ``leaves`` for the cotangent ``ones`` of ``out`` (every batch entry's value depends on its own entries of
    the leaves only, so the derivative of the batch sum IS the per-entry derivative)."""
    order, seen = [], set()
    stack = [(out, False)]
    while stack:  # iterative post-order: fun may build long chains
        node, done = stack.pop()
        if done:
            order.append(node)
            continue
        if id(node) in seen:
            continue
        seen.add(id(node))
        stack.append((node, True))
        for parent, _ in node.parents:
            if id(parent) not in seen:
                stack.append((parent, False))
    grads = {id(out): np.ones(out.shape, out.dtype if out.dtype.kind == "f" else np.float64)}
    for node in reversed(order):
        g = grads.get(id(node))
        if g is None:
            continue
        for parent, fn in node.parents:
            with np.errstate(all="ignore"):
                contrib = fn(g)
            grads[id(parent)] = contrib if id(parent) not in grads else grads[id(parent)] + contrib
    return [grads.get(id(leaf)) for leaf in leaves]


# ------------------------------------------------------------------------------------------------- what `fun` is handed
class TapePoint:
    """What ``fun`` receives for ``tx`` / ``rx`` on the tape route: ``xy`` is a recording array."""

    def __init__(self, xy):
        self.xy = xy


class TapePath:
    """What ``fun`` receives for ``path`` on the tape route: ``xys`` [..., k + 2, 2] is a recording array.  ``loss`` is not
    available there: the reference differentiates THROUGH it (it depends on the path's points and the objects), the traced
    number is a constant -- a function that reads it must supply ``fun.value_and_grad``."""

    def __init__(self, xys):
        self.xys = xys

    @property
    def loss(self):
        raise TapeError("fun reads path.loss, which the tape route holds as a traced constant (the reference differentiates "
                        "through it); supply fun.value_and_grad")

    def length(self):
        """Path length with the reference's guard (geometry.py:176-203: eps added to both components of every segment)."""
        v = (self.xys[..., 1:, :] - self.xys[..., :-1, :]) + F(EPS)
        return np.sqrt((v * v).sum(-1)).sum(-1)


def value_and_xys_bar(fun, fixed_xy, grid_xy, grid_is_rx, xys, loss, interacting, fun_args, fun_kwargs, point_cls, path_cls):
    """``fun`` and its derivative on one candidate's traced paths.

    fixed_xy [2], grid_xy [..., 2], xys [..., k + 2, 2], loss [...] -> (value [...], xys_bar [..., k + 2, 2]) in fp32, the
    derivative w.r.t. the end points as arguments of ``fun`` folded into rows 0 and k + 1."""
    fun_kwargs = dict(fun_kwargs or {})
    batch = xys.shape[:-2]
    user = getattr(fun, "value_and_grad", None)
    if user is not None:
        moving = point_cls(xy=grid_xy)
        fixed = point_cls(xy=fixed_xy)
        a, b = (fixed, moving) if grid_is_rx else (moving, fixed)
        out = user(a, b, path_cls(xys=xys, loss=loss), interacting, *fun_args, **fun_kwargs)
        if not isinstance(out, tuple) or len(out) not in (2, 4):
            raise TypeError("fun.value_and_grad must return (value, d/d path.xys) or (value, d/d path.xys, d/d tx.xy, d/d rx.xy)")
        val = np.broadcast_to(np.asarray(out[0], F), batch)
        bar = np.array(np.broadcast_to(np.asarray(out[1], F), xys.shape), F)
        if len(out) == 4:
            bar[..., 0, :] += np.asarray(out[2], F)
            bar[..., -1, :] += np.asarray(out[3], F)
        return val, bar
    t_xys = TapeArray(np.ascontiguousarray(xys, F))
    t_grid = TapeArray(np.ascontiguousarray(grid_xy, F))
    # (one row per cell for the fixed end point as well: its derivative is wanted per cell, not summed over the batch)
    t_fixed = TapeArray(np.ascontiguousarray(np.broadcast_to(np.asarray(fixed_xy, F), np.shape(grid_xy))))
    moving, fixed = TapePoint(t_grid), TapePoint(t_fixed)
    a, b = (fixed, moving) if grid_is_rx else (moving, fixed)
    try:
        with np.errstate(all="ignore"):
            val = fun(a, b, TapePath(t_xys), interacting, *fun_args, **fun_kwargs)
            if isinstance(val, TapeArray):
                if val.shape != tuple(batch):
                    val = np.broadcast_to(val, tuple(batch))
                g_xys, g_grid, g_fixed = backward(val, [t_xys, t_grid, t_fixed])
                val = val.value
            else:  # a constant
                val = np.broadcast_to(np.asarray(val, F), batch)
                g_xys = g_grid = g_fixed = None
    except Exception as e:
        raise L.D2DUnsupported(-4, f"fun={fun!r} could not be evaluated on recording arrays ({type(e).__name__}: {e}); write it "
                                   "with arithmetic operators / path.length() / NumPy functions, or supply "
                                   "fun.value_and_grad") from e
    bar = np.zeros(xys.shape, F) if g_xys is None else np.asarray(g_xys, F).copy()
    first, last = (g_fixed, g_grid) if grid_is_rx else (g_grid, g_fixed)
    if first is not None:
        bar[..., 0, :] += np.asarray(first, F)
    if last is not None:
        bar[..., -1, :] += np.asarray(last, F)
    return np.asarray(val, F), bar
