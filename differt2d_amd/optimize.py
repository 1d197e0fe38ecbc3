"""
The optimiser of the MinPath / FermatPath solvers (reference: differt2d/optimize.py).

The reference's ``minimize(fun, x0, args, steps, optimizer)`` runs any optax ``GradientTransformation`` on any JAX
objective (optimize.py:44-97); the path classes hand their objective -- the path's length or its interaction losses as a
function of the parametric coordinates -- to it (geometry.py:1172-1204, 1256-1288).  Here that loop IS the GPU kernel
(`d2d::power_opt_kernel` and the reverse sweep behind it, differt2d_amd/csrc), for the objectives of the two path classes and
for the optimiser the reference defaults to, Adam -- with any hyper-parameters: pass ``optimizer=adam(...)`` in
``path_cls_kwargs`` where the reference takes ``optimizer=optax.adam(...)``.  A general-purpose ``minimize`` for arbitrary
Python objectives is host-side autodiff and not part of this library.
"""

from dataclasses import dataclass

__all__ = ["Adam", "adam", "default_optimizer"]


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
