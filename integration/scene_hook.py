"""The early return a DiffeRT2d maintainer adds to ``Scene.accumulate_on_receivers_grid_over_paths``
(``differt2d/scene.py``, right after the argument normalisation at line 1874; the TX-grid twin at line 1571 is the same
with ``grid_role=_d2d.D2D_GRID_TX`` and ``self.receivers``), as a real function so that it can be executed:
``tests/test_gpu_integration.py`` runs it on this repository's Scene mirror with ``xp = numpy``; inside the reference
``xp`` is ``jax.numpy`` and ``hard_sigmoid`` / ``sigmoid`` / ``received_power`` / ``ImagePath`` / ``Wall`` are the
reference's own objects.

In the reference:

    def accumulate_on_receivers_grid_over_paths(self, X, Y, fun, fun_args=(), fun_kwargs=None, *, reduce_all=False, ...):
        ...                                                     # scene.py:1864-1874 unchanged
        if os.environ.get("DIFFERT2D_BACKEND") == "mi355x":
            from ._d2d_hook import mi355x_sweep, NOT_HANDLED
            out = mi355x_sweep(self, X, Y, fun, fun_args, fun_kwargs, reduce_all=reduce_all, grad=grad,
                               value_and_grad=value_and_grad, path_cls=path_cls, min_order=min_order, max_order=max_order,
                               order=order, filter_objects=filter_objects, kwargs=kwargs, xp=jnp,
                               names=dict(hard_sigmoid=hard_sigmoid, sigmoid=sigmoid, received_power=received_power,
                                          ImagePath=ImagePath, Wall=Wall, enable_approx=ENABLE_APPROX))
            if out is not NOT_HANDLED:
                return out
        ...                                                     # the JAX implementation, unchanged
"""

NOT_HANDLED = object()


def mi355x_sweep(scene, X, Y, fun, fun_args, fun_kwargs, *, reduce_all, grad, value_and_grad, path_cls, min_order,
                 max_order, order, filter_objects, kwargs, xp, names, binding=None, grid_role=None):
    """Returns what the reference's method returns (scene.py:1934-1953) or NOT_HANDLED (then the caller falls through
    to the JAX implementation: arbitrary Python ``fun``, non-Wall objects, optimiser-based path classes, custom
    activations)."""
    import numpy as np

    _d2d = binding
    if _d2d is None:
        from . import _d2d  # differt2d/_d2d.py in the reference
    fun_kwargs = fun_kwargs or {}
    acts = {names["hard_sigmoid"]: _d2d.D2D_ACT_HARD_SIGMOID, names["sigmoid"]: _d2d.D2D_ACT_SIGMOID}
    function = kwargs.get("function", names["hard_sigmoid"])
    if (path_cls is not names["ImagePath"] or fun is not names["received_power"] or fun_args or function not in acts
            or not all(type(o) is names["Wall"] for o in scene.objects)
            or set(kwargs) - {"approx", "alpha", "function", "tol", "patch"} or set(fun_kwargs) - {"r_coef", "height"}):
        return NOT_HANDLED
    approx = kwargs.get("approx")
    approx = names["enable_approx"] if approx is None else approx  # logic.py:58, 333-334
    lo, hi = (order, order) if order is not None else (min_order, max_order)  # scene.py:162-164
    role = _d2d.D2D_GRID_RX if grid_role is None else grid_role
    p = _d2d.make_params(lo, hi, approx, acts[function], kwargs.get("alpha", 100.0), kwargs.get("tol", 1e-2),
                         kwargs.get("patch", 0.0), fun_kwargs.get("r_coef", 0.5), fun_kwargs.get("height", 0.1), role)
    walls = np.stack([np.asarray(o.xys, np.float32) for o in scene.objects]) if scene.objects else np.zeros((0, 2, 2), np.float32)
    allowed = None if filter_objects is None else [bool(filter_objects(o)) for o in scene.objects]  # scene.py:1089-1134
    shape = _d2d.set_problem(walls, allowed, np.asarray(X), np.asarray(Y))
    want_grad = bool(grad or value_and_grad)
    fixed = scene.transmitters if role == _d2d.D2D_GRID_RX else scene.receivers

    def wrap(out):
        if value_and_grad:  # wins over grad, scene.py:1920-1923
            return tuple(xp.asarray(o) for o in out)
        return xp.asarray(out[1] if grad else out)

    if reduce_all:
        # scene.py:1939-1952: the per-transmitter maps are summed in dict order; on the device that is D2D_OUT_ADD into the
        # resident map(s), fetched once
        if not fixed:
            zero = np.zeros(shape, np.float32)
            return wrap((zero, np.zeros((*shape, 2), np.float32)) if want_grad else zero)
        for i, pt in enumerate(fixed.values()):
            _d2d.sweep(p, np.asarray(pt.xy), add=i > 0, grad=want_grad)
        return wrap(_d2d.fetch(shape, want_grad))

    def one(pt):
        _d2d.sweep(p, np.asarray(pt.xy), add=False, grad=want_grad)
        return wrap(_d2d.fetch(shape, want_grad))

    return ((name, one(pt)) for name, pt in fixed.items())  # scene.py:1934-1937: a generator of (name, array)
