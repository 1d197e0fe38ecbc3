"""
Abstract protocols with the reference's names (``differt2d/abc.py``): :class:`Plottable`,
:class:`Interactable` and :class:`Object`.  ``Interactable`` is the reference's object plug-in
interface (abc.py:129-256); the native kernels implement it for ``Wall``, ``RIS`` and ``Vertex``.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Literal, Optional

import numpy as np

Loc = Literal["N", "E", "S", "W", "C", "NE", "NW", "SE", "SW"]
F = np.float32


def key_to_generator(key) -> np.random.Generator:
    """A NumPy generator for callers that ask for NumPy's PRNG explicitly (a ``Generator``, or ``None`` = fresh entropy).
    The reference's ``key`` arguments -- int seeds, Threefry keys -- are drawn with differt2d_amd/random.py instead
    (``random_uniform``)."""
    if isinstance(key, np.random.Generator):
        return key
    if key is None:
        return np.random.default_rng()
    return np.random.default_rng(np.asarray(key, dtype=np.uint32).reshape(-1).tolist())


def random_uniform(key, shape) -> np.ndarray:
    """``jax.random.uniform(key, shape)`` for an int seed / Threefry key (the reference's numbers: differt2d_amd/random.py), or
    the same shape from a ``numpy.random.Generator``."""
    if isinstance(key, np.random.Generator):
        return key.random(shape, dtype=F)
    if key is None:
        raise TypeError("a `key` is needed to draw random numbers")
    from . import random as jr

    return jr.uniform(jr.as_key(key), shape)


def linspace_f32(start, stop, num: int) -> np.ndarray:
    """fp32 linspace the way ``jnp.linspace`` evaluates it (recalled from jax/_src/numpy/lax_numpy.py, not
    verifiable here): ``start * (1 - i/div) + stop * (i/div)`` for ``i < div = num - 1``, then ``stop`` itself."""
    start, stop = F(start), F(stop)
    if num <= 0:
        return np.empty(0, F)
    if num == 1:
        return np.array([start], F)
    div = num - 1
    step = np.arange(div, dtype=F) / F(div)
    out = start * (F(1.0) - step) + stop * step
    return np.concatenate([out, np.array([stop], F)]).astype(F)


class Plottable(ABC):
    """Anything that can be drawn with matplotlib and has a bounding box (reference abc.py:29-126)."""

    @abstractmethod
    def plot(self, ax, *args: Any, **kwargs: Any):
        """Draws this object on ``ax`` and returns the artists."""

    @abstractmethod
    def bounding_box(self) -> np.ndarray:
        """``[[min_x, min_y], [max_x, max_y]]``."""

    def grid(self, m: int = 50, n: Optional[int] = None):
        """Mesh grid overlaying the object: ``m`` samples along x, ``n`` (default ``m``) along y;
        'xy' indexing, so both outputs have shape ``(n, m)`` (reference abc.py:57-81)."""
        bbox = self.bounding_box()
        if n is None:
            n = m
        x = linspace_f32(bbox[0, 0], bbox[1, 0], m)
        y = linspace_f32(bbox[0, 1], bbox[1, 1], n)
        X, Y = np.meshgrid(x, y)
        # immutable, like the reference's JAX arrays (docs/source/jax_and_jaxtyping.md:52-57): a grid passed to a sweep again
        # is then recognised by identity and not uploaded a second time (engine.Context.set_grid)
        X.setflags(write=False)
        Y.setflags(write=False)
        return X, Y

    def center(self) -> np.ndarray:
        """Centre of the bounding box (reference abc.py:83-96)."""
        bbox = self.bounding_box()
        return (F(0.5) * (bbox[0, :] + bbox[1, :])).astype(F)

    def get_location(self, location: Loc) -> np.ndarray:
        """Compass location within the bounding box (reference abc.py:98-126)."""
        (xmin, ymin), (xmax, ymax) = self.bounding_box()
        xavg, yavg = F(0.5) * (xmin + xmax), F(0.5) * (ymin + ymax)
        table = {
            "N": (xavg, ymax), "E": (xmax, yavg), "S": (xavg, ymin), "W": (xmin, yavg), "C": (xavg, yavg),
            "NE": (xmax, ymax), "NW": (xmin, ymax), "SE": (xmax, ymin), "SW": (xmin, ymin),
        }
        return np.array(table[location], dtype=F)


class Interactable(ABC):
    """Anything a ray path can interact with (reference abc.py:129-256)."""

    @staticmethod
    @abstractmethod
    def parameters_count() -> int:
        """Number of parametric coordinates (``Wall``: 1, ``Vertex``: 0)."""

    @abstractmethod
    def parametric_to_cartesian(self, param_coords) -> np.ndarray:
        """Parametric -> cartesian coordinates."""

    def sample(self, key) -> np.ndarray:
        """A random point on the object (uniform parametric coordinates in [0, 1))."""
        return self.parametric_to_cartesian(random_uniform(key, (self.parameters_count(),)))  # reference abc.py:176-178

    @abstractmethod
    def cartesian_to_parametric(self, carte_coords) -> np.ndarray:
        """Cartesian -> parametric coordinates."""

    @abstractmethod
    def contains_parametric(self, param_coords, approx: Optional[bool] = None, **kwargs: Any):
        """Whether the parametric coordinates lie on the object."""

    @abstractmethod
    def intersects_cartesian(self, ray, patch: float = 0.0, approx: Optional[bool] = None, **kwargs: Any):
        """Whether the ray ``[[x0, y0], [x1, y1]]`` hits the object (stretched by ``patch``)."""

    @abstractmethod
    def evaluate_cartesian(self, ray_path) -> np.ndarray:
        """Interaction residual of a 3-point ray path; 0 means a perfect interaction."""


class Object(Plottable, Interactable):
    """Both :class:`Plottable` and :class:`Interactable` (reference abc.py:259-266)."""
