"""Derivatives of a user-supplied path function ``fun(tx, rx, path, interacting_objects, *args, **kwargs)`` on the host.

The reference differentiates whatever JAX callable it is given (scene.py:1892-1923: ``jax.grad`` of
``sum_c valid_c * fun_c`` w.r.t. the grid cell).  Here the validity and the image method are differentiated on the GPU by the
hand-derived adjoint (``d2d_power_map_vg_launch`` with ``D2D_FUN_CUSTOM``, include/d2d.h); what the kernel needs from the host
is ``fun`` itself and its derivative w.r.t. the path's points, per (candidate, cell).  Two ways to obtain them:

* ``fun.value_and_grad(tx, rx, path, interacting_objects, *args, **kwargs)`` -- supplied by the user -- returning
  ``(value, d value / d path.xys)`` or ``(value, d value / d path.xys, d value / d tx.xy, d value / d rx.xy)``;
* otherwise ``fun`` is called on tensors that record their operations (``torch.autograd``; arithmetic operators,
  ``path.length()``, tensor methods and ``torch.*`` functions work, ``numpy`` functions do not) and its derivative is read
  off the tape.  This is the only use of an autodiff tape in the package, it concerns the user's function alone and it is
  imported on demand.
"""

from __future__ import annotations

import numpy as np

from . import _lib as L

F = np.float32
EPS = float(np.finfo(np.float32).eps)


class TapePoint:
    """What ``fun`` receives for ``tx`` / ``rx`` on the tape route: ``xy`` is a tensor."""

    def __init__(self, xy):
        self.xy = xy


class TapePath:
    """What ``fun`` receives for ``path`` on the tape route: ``xys`` [..., k + 2, 2] and ``loss`` are tensors."""

    def __init__(self, xys, loss):
        self.xys = xys
        self.loss = loss

    def length(self):
        """Path length with the reference's guard (geometry.py:176-203: eps added to both components of every segment)."""
        v = (self.xys[..., 1:, :] - self.xys[..., :-1, :]) + EPS
        return (v * v).sum(-1).sqrt().sum(-1)


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
    try:
        import torch
    except ImportError as e:  # pragma: no cover
        raise L.D2DUnsupported(-4, "the gradient of a path function that is not fused natively needs either "
                                   "fun.value_and_grad or torch (to record fun's operations)") from e
    t_xys = torch.tensor(np.ascontiguousarray(xys, F), requires_grad=True)
    t_grid = torch.tensor(np.ascontiguousarray(grid_xy, F), requires_grad=True)
    # (one row per cell for the fixed end point as well: its derivative is wanted per cell, not summed over the batch)
    t_fixed = torch.tensor(np.ascontiguousarray(np.broadcast_to(np.asarray(fixed_xy, F), np.shape(grid_xy))), requires_grad=True)
    moving, fixed = TapePoint(t_grid), TapePoint(t_fixed)
    a, b = (fixed, moving) if grid_is_rx else (moving, fixed)
    try:
        val = fun(a, b, TapePath(t_xys, torch.tensor(np.ascontiguousarray(loss, F))), interacting, *fun_args, **fun_kwargs)
    except Exception as e:
        raise L.D2DUnsupported(-4, f"fun={fun!r} could not be evaluated on recording tensors ({type(e).__name__}: {e}); write it "
                                   "with arithmetic operators / path.length() / torch functions, or supply "
                                   "fun.value_and_grad") from e
    if not isinstance(val, torch.Tensor):
        val = torch.as_tensor(val, dtype=torch.float32)
    val = val.to(torch.float32).expand(batch) if val.shape != tuple(batch) else val.to(torch.float32)
    if val.requires_grad:
        g_xys, g_grid, g_fixed = torch.autograd.grad(val.sum(), [t_xys, t_grid, t_fixed], allow_unused=True)
    else:  # a constant
        g_xys = g_grid = g_fixed = None
    bar = np.zeros(xys.shape, F) if g_xys is None else g_xys.numpy().astype(F)
    first, last = (g_fixed, g_grid) if grid_is_rx else (g_grid, g_fixed)
    if first is not None:
        bar[..., 0, :] += first.numpy()
    if last is not None:
        bar[..., -1, :] += last.numpy()
    return val.detach().numpy().astype(F), bar
