"""
Optimization toolbox (reference: differt2d/optimize.py).

Two layers, as in the reference:

* The MinPath / FermatPath solvers.  The reference hands the path classes' objective -- the path's length or its interaction
  losses as a function of the parametric coordinates -- to :func:`minimize` (geometry.py:1172-1204, 1256-1288); here that loop IS
  the GPU kernel (`d2d::power_opt_kernel` and the reverse sweep behind it, differt2d_amd/csrc) for those objectives and for the
  optimiser the reference defaults to, Adam -- with any hyper-parameters: pass ``optimizer=adam(...)`` in ``path_cls_kwargs``
  where the reference takes ``optimizer=optax.adam(...)``.  No sweep ever calls the functions below.
* :func:`minimize`, :func:`minimize_random_uniform`, :func:`minimize_many_random_uniform` as CALLABLES for a user's own Python
  objective (optimize.py:44-182): a host utility, like the reference's (which is plain JAX on whatever backend), with the same
  signatures, the same update rule (``optax.adam``, fp32, optax's order of operations -- the one `oracle/ref.py:616-638` and the
  kernels follow), the same initial guesses (``jax.random.uniform`` on the Threefry key: differt2d_amd/random.py) and the
  reference's return convention (the loss evaluated BEFORE the last update, optimize.py:86-97).  The objective's gradient comes
  from the NumPy recording tape of differt2d_amd/fun_grad.py (JAX's conventions at ties, ``where``, ``sqrt'(0)``), or from
  ``fun.value_and_grad(x, *args)`` when the user supplies one.  The reference's known answers (tests/test_optimize.py:27-74 and
  the doctests) run against these in tests/test_host_api.py.
"""

from dataclasses import dataclass
from typing import Any, Callable, Optional

import numpy as np

__all__ = ["Adam", "adam", "default_optimizer", "minimize", "minimize_random_uniform", "minimize_many_random_uniform"]

F = np.float32


@dataclass(frozen=True)
class Adam:
    """``optax.adam(learning_rate, b1, b2, eps)`` (optax 0.2: ``eps_root = 0``, no Nesterov momentum)."""

    learning_rate: float = 0.1
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8


def adam(learning_rate: float = 0.1, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> Adam:
    """Same signature as ``optax.adam`` for the arguments the native solver has."""
    return Adam(float(learning_rate), float(b1), float(b2), float(eps))


def default_optimizer() -> Adam:
    """The reference's default (optimize.py:83): ``optax.adam(learning_rate=0.1)``."""
    return Adam()


def _value_and_grad(fun: Callable, x: np.ndarray, args: tuple):
    """``jax.value_and_grad(fun)(x, *args)`` on the host: the user's own ``fun.value_and_grad`` or the recording tape."""
    user = getattr(fun, "value_and_grad", None)
    if user is not None:
        v, g = user(x, *args)
        return F(v), np.asarray(g, F).reshape(x.shape)
    from .fun_grad import TapeArray, backward

    t = TapeArray(x)
    out = fun(t, *args)
    if not isinstance(out, TapeArray):  # a constant objective: zero gradient, like jax.grad
        return F(out), np.zeros_like(x)
    if out.shape != ():
        raise TypeError(f"the objective must return a scalar, got shape {out.shape}")
    (g,) = backward(out, [t])
    return F(out.value), (np.zeros_like(x) if g is None else np.asarray(g, F).reshape(x.shape))


def minimize(fun: Callable, x0, args: tuple = (), steps: int = 100, optimizer: Optional[Adam] = None):
    """Minimizes a scalar function of one or more variables (reference optimize.py:44-97).

    Returns ``(x, loss)``: the solution after ``steps`` updates and the loss evaluated before the last of them (the
    reference's ``losses[-1]`` of a ``lax.scan`` whose body evaluates, then updates).  ``optimizer``: an :class:`Adam`
    (:func:`adam`); anything else is refused -- the reference takes any ``optax.GradientTransformation``, this library has Adam.

    >>> import numpy as np
    >>> def f(x, offset=1.0):
    ...     x = x - offset
    ...     return np.dot(x, x)
    >>> x, y = minimize(f, np.zeros(10))
    >>> bool(np.allclose(x, 1.0, rtol=1e-2)) and bool(abs(y) <= 1e-4)
    True
    >>> x, y = minimize(f, np.zeros(10), args=(2.0,))
    >>> bool(np.allclose(x, 2.0, rtol=1e-2)) and bool(abs(y) <= 1e-3)
    True
    """
    opt = optimizer or default_optimizer()
    if not isinstance(opt, Adam):
        from ._lib import D2DUnsupported

        raise D2DUnsupported(-4, f"optimizer {type(opt).__name__}: only differt2d_amd.optimize.adam(...) is implemented")
    x = np.array(x0, F)
    mu, nu = np.zeros_like(x), np.zeros_like(x)
    # optax.scale_by_adam + scale(-lr) in fp32; the constants 1 - b and 1 - b**t are formed in double and cast, this order of
    # operations: oracle/ref.py:616-638 (which tests/test_oracle_opt_c.py pins the C oracle and the solver kernels to)
    b1, b2 = float(opt.b1), float(opt.b2)
    eps, lr = F(opt.eps), F(-float(opt.learning_rate))
    loss = F(np.nan)
    for t in range(1, int(steps) + 1):
        loss, g = _value_and_grad(fun, x, tuple(args))
        with np.errstate(all="ignore"):
            mu = F(b1) * mu + F(1.0 - b1) * g
            nu = F(b2) * nu + F(1.0 - b2) * (g * g)
            mh = mu / F(1.0 - b1**t)
            nh = nu / F(1.0 - b2**t)
            x = (x + lr * (mh / (np.sqrt(nh) + eps))).astype(F)
    return x, loss


def minimize_random_uniform(fun: Callable, key, n: int, **kwargs: Any):
    """:func:`minimize` from ``x0 = jax.random.uniform(key, (n,))`` (reference optimize.py:100-135).

    >>> import numpy as np
    >>> from differt2d_amd.random import PRNGKey
    >>> def f(x):
    ...     x = x - 1.0
    ...     return np.dot(x, x)
    >>> x, y = minimize_random_uniform(f, PRNGKey(1234), 10)
    >>> bool(np.allclose(x, 1.0, rtol=1e-2)) and bool(abs(y) <= 1e-3)
    True
    """
    from .random import uniform

    return minimize(fun, uniform(key, (int(n),)), **kwargs)


def minimize_many_random_uniform(fun: Callable, key, n: int, many: int = 10, **kwargs: Any):
    """The best of ``many`` runs of :func:`minimize_random_uniform` on ``jax.random.split(key, many)`` (reference
    optimize.py:138-182; ``many == 1`` uses ``key`` itself, as there).

    >>> import numpy as np
    >>> from differt2d_amd.random import PRNGKey
    >>> def f(x):
    ...     x = x - 1.0
    ...     return np.dot(x, x)
    >>> x, y = minimize_many_random_uniform(f, PRNGKey(1234), 10)
    >>> bool(np.allclose(x, 1.0, rtol=1e-2)) and bool(abs(y) <= 1e-4)
    True
    """
    if many == 1:
        return minimize_random_uniform(fun, key, n, **kwargs)
    from .random import split

    runs = [minimize_random_uniform(fun, k, n, **kwargs) for k in split(key, int(many))]
    losses = np.array([r[1] for r in runs], F)
    i_min = int(np.argmin(losses))  # (jnp.argmin: the first of equal minima; NaN counts as the minimum there and here)
    if np.isnan(losses).any():
        i_min = int(np.flatnonzero(np.isnan(losses))[0])
    return runs[i_min]
