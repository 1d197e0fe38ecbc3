"""
Geometrical objects with the reference's names, constructors and method signatures
(``differt2d/geometry.py``), backed by NumPy arrays (fp32) instead of JAX arrays.

Division of labour
------------------
* Per-object accessors (``Wall.normal``, ``image_of``, ``cartesian_to_parametric`` ...) and the free
  helpers (``segments_intersect``, ``path_length``, ``normalize``) are small host-side NumPy functions:
  they exist so that user code written against the reference keeps working.
* Everything on the hot path -- solving a path (``ImagePath.from_tx_objects_rx``), ``Path.on_objects``,
  ``Path.intersects_with_objects``, ``Path.is_valid`` and every ``Scene`` sweep -- runs on the GPU through
  the C ABI (``d2d_trace_paths`` / ``d2d_power_map_*``).  There is no CPU implementation of those to fall
  back to: without ``libd2d.so`` and an MI355X they raise.
"""

from __future__ import annotations

import dataclasses
import math
from typing import Any, Optional, Sequence, Union

import numpy as np

from . import _lib as L
from . import logic
from .abc import Interactable, Object, Plottable
from .defaults import DEFAULT_PATCH

F = np.float32
ArrayLike = Any


def _xy(p) -> np.ndarray:
    """Coordinates of a Point or of an array-like, as fp32."""
    return p.xy if isinstance(p, Point) else np.asarray(p, dtype=F)


# --------------------------------------------------------------------------------------
# free functions (reference geometry.py:82-267)
# --------------------------------------------------------------------------------------


def segments_intersect(P1, P2, P3, P4, tol=0.005, approx: Optional[bool] = None, **kwargs):
    """Whether segments P1-P2 and P3-P4 intersect, within ``[-tol, 1 + tol]`` (reference geometry.py:82-173)."""
    P1, P2, P3, P4 = (np.asarray(p, dtype=F) for p in (P1, P2, P3, P4))
    tol = F(tol)
    A, B, C = P2 - P1, P3 - P4, P1 - P3
    a = B[..., 1] * C[..., 0] - B[..., 0] * C[..., 1]
    b = A[..., 0] * C[..., 1] - A[..., 1] * C[..., 0]
    d = A[..., 1] * B[..., 0] - A[..., 0] * B[..., 1]

    def test(num, den):
        zero = den == F(0.0)
        safe = np.where(zero, F(1.0), den).astype(F)
        with np.errstate(all="ignore"):
            t = np.where(zero, F(np.inf), num / safe).astype(F)
        return logic.logical_and(
            logic.greater_equal(t, -tol, approx=approx, **kwargs),
            logic.less_equal(t, F(1.0) + tol, approx=approx, **kwargs),
            approx=approx,
        )

    return logic.logical_and(test(a, d), test(b, d), approx=approx)


def path_length(points) -> np.ndarray:
    """Length of a polyline ``(..., N, 2)``; ``eps`` is added to every difference (reference geometry.py:176-203)."""
    points = np.asarray(points, dtype=F)
    vectors = np.diff(points, axis=-2) + np.finfo(F).eps
    lengths = np.sqrt(vectors[..., 0] * vectors[..., 0] + vectors[..., 1] * vectors[..., 1])
    total = lengths[..., 0]
    for i in range(1, lengths.shape[-1]):
        total = total + lengths[..., i]
    return total.astype(F)


def normalize(vector):
    """Returns ``(vector / length, length)`` with the zero vector mapped to itself and length 1
    (reference geometry.py:206-230)."""
    vector = np.asarray(vector, dtype=F)
    length = np.sqrt(vector[..., 0] * vector[..., 0] + vector[..., 1] * vector[..., 1])
    length = np.where(length == F(0.0), F(1.0), length).astype(F)
    return (vector / length[..., None]).astype(F), length


def closest_point(points, target):
    """Index of the point closest to ``target`` and its distance (reference geometry.py:233-267)."""
    points = np.asarray(points, dtype=F).reshape(-1, 2)
    diff = points - np.asarray(target, dtype=F).reshape(-1, 2)
    distances = np.sqrt(diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1])
    i_min = int(np.argmin(distances))
    return np.int32(i_min), distances[i_min]


def stack_leaves(objects: Sequence["Object"]):
    """Stacks homogeneous objects into one object whose arrays have a new leading axis
    (reference geometry.py:42-64). Raises ``ValueError`` on mixed types, like the reference."""
    objects = list(objects)
    if not objects:
        raise ValueError("need at least one object")
    cls = type(objects[0])
    if any(type(o) is not cls for o in objects):
        raise ValueError("all objects must be of the same type")
    fields = {f.name: np.stack([np.asarray(getattr(o, f.name)) for o in objects]) for f in dataclasses.fields(cls)}
    out = object.__new__(cls)
    for k, v in fields.items():
        object.__setattr__(out, k, v)
    return out


def unstack_leaves(stacked) -> list:
    """Reciprocal of :func:`stack_leaves` (reference geometry.py:67-79)."""
    cls = type(stacked)
    names = [f.name for f in dataclasses.fields(cls)]
    n = len(np.asarray(getattr(stacked, names[0])))
    return [cls(**{k: np.asarray(getattr(stacked, k))[i] for k in names}) for i in range(n)]


# --------------------------------------------------------------------------------------
# Point / Vertex / Ray / Wall / RIS
# --------------------------------------------------------------------------------------


@dataclasses.dataclass(frozen=True, eq=False)
class Point(Plottable):
    """A point defined by its coordinates (reference geometry.py:270-349)."""

    xy: np.ndarray = dataclasses.field(default_factory=lambda: np.zeros(2, F))

    def __post_init__(self):
        object.__setattr__(self, "xy", np.asarray(self.xy, dtype=F))

    def bounding_box(self) -> np.ndarray:
        return np.vstack([self.xy, self.xy]).astype(F)

    def plot(self, ax, *args, annotate: Optional[str] = None, annotate_offset=(0.0, 0.0), annotate_kwargs=None, **kwargs):
        kwargs.setdefault("marker", "o")
        kwargs.setdefault("color", "red")
        x, y = self.xy
        artists = [ax.scatter(x, y, *args, **kwargs)]
        if annotate:
            off = self.xy + np.asarray(annotate_offset, dtype=float)
            artists.append(ax.annotate(annotate, xy=(x, y), xytext=(off[0], off[1]), **(annotate_kwargs or {})))
        return artists


@dataclasses.dataclass(frozen=True, eq=False)
class Vertex(Point, Object):
    """A diffraction vertex: zero parameters, never occludes (reference geometry.py:352-431)."""

    kind = L.D2D_VERTEX

    @staticmethod
    def parameters_count() -> int:
        return 0

    def parametric_to_cartesian(self, param_coords=None) -> np.ndarray:
        return self.xy

    def cartesian_to_parametric(self, carte_coords) -> np.ndarray:
        return np.empty(0, F)

    def contains_parametric(self, param_coords=None, approx: Optional[bool] = None, **kwargs):
        return logic.true_value(approx=approx)

    def intersects_cartesian(self, ray, patch=DEFAULT_PATCH, approx: Optional[bool] = None, **kwargs):
        return logic.false_value(approx=approx)

    def evaluate_cartesian(self, ray_path) -> np.ndarray:
        return F(0.0)

    def as_rows(self) -> np.ndarray:
        return np.vstack([self.xy, self.xy]).astype(F)

    def plot(self, ax, *args, **kwargs):
        kwargs.setdefault("edgecolors", "black")
        kwargs.setdefault("facecolors", (1.0, 1.0, 0.0, 0.5))
        kwargs.setdefault("linestyle", "dashed")
        return super().plot(ax, *args, **kwargs)


@dataclasses.dataclass(frozen=True, eq=False)
class Ray(Plottable):
    """A segment with origin and destination (reference geometry.py:434-539)."""

    xys: np.ndarray = dataclasses.field(default_factory=lambda: np.array([[0.0, 0.0], [1.0, 1.0]], F))

    def __post_init__(self):
        object.__setattr__(self, "xys", np.asarray(self.xys, dtype=F))

    def origin(self) -> np.ndarray:
        return self.xys[..., 0, :]

    def dest(self) -> np.ndarray:
        return self.xys[..., 1, :]

    def t(self) -> np.ndarray:
        return self.dest() - self.origin()

    def rotate(self, angle, around=None):
        """Rotated copy (reference geometry.py:489-528)."""
        center = np.zeros(2, F) if around is None else _xy(around)
        c, s = F(math.cos(angle)), F(math.sin(angle))
        rot = np.array([[c, -s], [s, c]], dtype=F)
        xys = (rot @ (self.xys - center[None, :]).T).T + center[None, :]
        return dataclasses.replace(self, xys=xys.astype(F))

    def plot(self, ax, *args, **kwargs):
        kwargs.setdefault("color", "blue")
        x, y = self.xys.T
        return ax.plot(x, y, *args, **kwargs)

    def bounding_box(self) -> np.ndarray:
        return np.vstack([np.min(self.xys, axis=0), np.max(self.xys, axis=0)]).astype(F)


@dataclasses.dataclass(frozen=True, eq=False)
class Wall(Ray, Object):
    """A reflecting wall (reference geometry.py:542-680)."""

    kind = L.D2D_WALL

    def normal(self) -> np.ndarray:
        t = self.t()
        n, _ = normalize(np.stack([t[..., 1], -t[..., 0]], axis=-1))
        return n

    @staticmethod
    def parameters_count() -> int:
        return 1

    def parametric_to_cartesian(self, param_coords) -> np.ndarray:
        return (self.origin() + np.asarray(param_coords, dtype=F) * self.t()).astype(F)

    def cartesian_to_parametric(self, carte_coords) -> np.ndarray:
        other = np.asarray(carte_coords, dtype=F) - self.origin()
        t = self.t()
        sq = t[0] * t[0] + t[1] * t[1]
        sq = F(1.0) if sq == 0.0 else sq
        return ((t[0] * other[..., 0] + t[1] * other[..., 1]).reshape(-1) / sq).astype(F)

    def contains_parametric(self, param_coords, approx: Optional[bool] = None, **kwargs):
        s = np.asarray(param_coords, dtype=F)[0]
        ge = logic.greater_equal(s, F(0.0), approx=approx, **kwargs)
        le = logic.less_equal(s, F(1.0), approx=approx, **kwargs)
        return logic.logical_and(ge, le, approx=approx)

    def intersects_cartesian(self, ray, patch=DEFAULT_PATCH, approx: Optional[bool] = None, **kwargs):
        ray = np.asarray(ray, dtype=F)
        patch = F(patch)
        return segments_intersect(
            self.origin() - patch * self.t(), self.dest() + patch * self.t(), ray[0, :], ray[1, :], approx=approx, **kwargs
        )

    def evaluate_cartesian(self, ray_path) -> np.ndarray:
        ray_path = np.asarray(ray_path, dtype=F)
        i, _ = normalize(ray_path[1, :] - ray_path[0, :])
        r, _ = normalize(ray_path[2, :] - ray_path[1, :])
        n = self.normal()
        din = i[0] * n[0] + i[1] * n[1]
        e = r - (i - F(2.0) * din * n)
        return F(e[0] * e[0] + e[1] * e[1])

    def image_of(self, point) -> np.ndarray:
        point = np.asarray(point, dtype=F)
        i = point - self.origin()
        n = self.normal()
        return (point - F(2.0) * (i[..., 0] * n[0] + i[..., 1] * n[1])[..., None] * n).astype(F)

    def get_vertices(self):
        return Vertex(xy=self.xys[0, :]), Vertex(xy=self.xys[1, :])

    def as_rows(self) -> np.ndarray:
        return self.xys


@dataclasses.dataclass(frozen=True, eq=False)
class RIS(Wall):
    """Reflective intelligent surface with a fixed reflection angle (reference geometry.py:683-721)."""

    phi: np.ndarray = dataclasses.field(default_factory=lambda: F(math.pi / 4))
    kind = L.D2D_RIS

    def __post_init__(self):
        super().__post_init__()
        object.__setattr__(self, "phi", np.asarray(self.phi, dtype=F))

    def evaluate_cartesian(self, ray_path) -> np.ndarray:
        ray_path = np.asarray(ray_path, dtype=F)
        r, _ = normalize(ray_path[2, :] - ray_path[1, :])
        n = self.normal()
        mr = -r
        sin_a = mr[0] * n[1] - mr[1] * n[0]
        cos_a = mr[0] * n[0] + mr[1] * n[1]
        ds, dc = sin_a - np.sin(self.phi), cos_a - np.cos(self.phi)
        return F(ds * ds + dc * dc)

    def plot(self, ax, *args, **kwargs):
        kwargs.setdefault("color", "green")
        return super().plot(ax, *args, **kwargs)


# --------------------------------------------------------------------------------------
# scene tables for the C ABI
# --------------------------------------------------------------------------------------


def objects_to_tables(objects: Sequence[Object]):
    """(xys[N,2,2], kind[N], phi[N]) for ``d2d_set_scene``."""
    n = len(objects)
    try:
        rows = [o.as_rows() for o in objects]
        kinds = [o.kind for o in objects]
    except AttributeError:
        bad = next(o for o in objects if not hasattr(o, "as_rows") or not hasattr(o, "kind"))
        raise L.D2DUnsupported(-4, f"object {bad!r} is not a native Wall / RIS / Vertex") from None
    xys = np.array(rows, F).reshape(n, 2, 2) if n else np.zeros((0, 2, 2), F)  # (one conversion for the whole scene: every sweep call passes here)
    kind = np.array(kinds, np.uint8) if n else np.zeros(0, np.uint8)
    phi = np.full(n, math.pi / 4, F)
    if L.D2D_RIS in kinds:
        for i, o in enumerate(objects):
            if isinstance(o, RIS):
                phi[i] = o.phi
    return xys, kind, phi


def _validity_kwargs(tol=1e-2, patch=DEFAULT_PATCH, approx=None, alpha=None, function=None, **extra):
    if extra:
        raise TypeError(f"unexpected keyword arguments: {sorted(extra)}")
    kw = dict(tol=tol, patch=patch, approx=logic._resolve(approx), function=logic.native_activation_name(function))
    if alpha is not None:
        kw["alpha"] = alpha
    return kw


# --------------------------------------------------------------------------------------
# Path and solvers
# --------------------------------------------------------------------------------------


@dataclasses.dataclass(frozen=True, eq=False)
class Path(Plottable):
    """A ray path: ``xys`` of shape ``(num_points, 2)`` (leading batch dimensions allowed) and the solver's
    ``loss`` (reference geometry.py:724-973)."""

    xys: np.ndarray
    loss: np.ndarray = dataclasses.field(default_factory=lambda: F(0.0))

    def __post_init__(self):
        object.__setattr__(self, "xys", np.asarray(self.xys, dtype=F))
        object.__setattr__(self, "loss", np.asarray(self.loss, dtype=F))

    solver = None  # base class: parametric mid points, no solve

    @classmethod
    def from_tx_objects_rx(cls, tx, objects: Sequence[Interactable], rx, *, key=None, **kwargs):
        """Path through the parametric mid point of every object (reference geometry.py:752-809)."""
        pts = [o.parametric_to_cartesian(np.array([0.5], F)) for o in objects]
        return cls(xys=np.vstack([_xy(tx), *pts, _xy(rx)]))

    def length(self) -> np.ndarray:
        """Path length (reference geometry.py:811-819)."""
        return path_length(self.xys)

    # -- validity: GPU (d2d_trace_paths in "given points" mode) -------------------------
    def _validate(self, objects, path_candidate, **kwargs):
        from .engine import default_context, make_params

        cand = np.asarray(path_candidate, dtype=np.int32).reshape(-1)
        if self.xys.ndim != 2 or self.xys.shape[0] != cand.size + 2:
            raise ValueError(f"a path with {cand.size} interactions needs {cand.size + 2} points, got {self.xys.shape}")
        if cand.size > L.D2D_MAX_ORDER:
            raise L.D2DError(-1, f"paths with more than D2D_MAX_ORDER={L.D2D_MAX_ORDER} interactions are not supported")
        ctx = default_context()
        ctx.set_scene(*objects_to_tables(objects))
        NP = L.D2D_MAX_ORDER + 2
        xin = np.full((1, 1, NP, 2), np.nan, F)
        xin[0, 0, : self.xys.shape[0]] = self.xys
        p = make_params(max_order=L.D2D_MAX_ORDER, **_validity_kwargs(**kwargs))
        return ctx.trace_paths(p, self.xys[0], self.xys[-1], [cand], xys_in=xin, loss_in=np.asarray(self.loss, F).reshape(1, 1))

    def on_objects(self, objects: Sequence[Interactable], approx: Optional[bool] = None, **kwargs):
        """Whether point ``i+1`` lies on object ``i`` (reference geometry.py:821-854)."""
        out = self._validate(objects, np.arange(len(objects)), approx=approx, **kwargs)["on"][0, 0]
        return out if logic._resolve(approx) else np.bool_(out != 0)

    def intersects_with_objects(self, objects, path_candidate, patch=DEFAULT_PATCH, approx: Optional[bool] = None, **kwargs):
        """Whether any scene object (other than the two a segment touches) blocks the path
        (reference geometry.py:856-906)."""
        out = self._validate(objects, path_candidate, patch=patch, approx=approx, **kwargs)["hit"][0, 0]
        return out if logic._resolve(approx) else np.bool_(out != 0)

    def is_valid(self, objects, path_candidate, interacting_objects=None, tol=1e-2, patch=DEFAULT_PATCH,
                 approx: Optional[bool] = None, **kwargs):
        """``nan_to_num(all(on_objects, not intersects, loss < tol))`` (reference geometry.py:908-963)."""
        out = self._validate(objects, path_candidate, tol=tol, patch=patch, approx=approx, **kwargs)["valid"][0, 0]
        return out if logic._resolve(approx) else np.bool_(out != 0)

    def plot(self, ax, *args, **kwargs):
        kwargs.setdefault("color", "orange")
        x, y = self.xys.T
        return ax.plot(x, y, *args, **kwargs)

    def bounding_box(self) -> np.ndarray:
        return np.vstack([np.min(self.xys, axis=0), np.max(self.xys, axis=0)]).astype(F)


class ImagePath(Path):
    """Path obtained with the image method (reference geometry.py:1013-1114); solved on the GPU."""

    solver = "image"

    @classmethod
    def from_tx_objects_rx(cls, tx, objects: Sequence[Wall], rx, *, key=None, **kwargs):
        from .engine import default_context, make_params

        objects = list(objects)
        if len({type(o) for o in objects}) > 1:
            # the reference stacks the objects' leaves and fails on heterogeneous lists
            raise ValueError("ImagePath needs objects of one type (reference: stack_leaves raises)")
        k = len(objects)
        ctx = default_context()
        ctx.set_scene(*objects_to_tables(objects))
        out = ctx.trace_paths(make_params(max_order=L.D2D_MAX_ORDER), _xy(tx), _xy(rx), [np.arange(k, dtype=np.int32)])
        return cls(xys=out["xys"][0, 0, : k + 2], loss=out["loss"][0, 0])


def _theta0_rows(key, count: int, many: int):
    """``minimize_many_random_uniform``'s draws for one path (reference optimize.py:132, 174-178): ``uniform(key, (n,))``, or,
    ``many > 1``, one draw per key of ``split(key, many)``."""
    from . import random as jr

    if many == 1:
        return [jr.uniform(key, (count,))]
    return [jr.uniform(k, (count,)) for k in jr.split(key, many)]


def draw_theta0(objects_per_candidate, key, theta0=None, many: int = 1, per_candidate_keys: bool = False):
    """Initial parametric guesses: ``many`` consecutive rows per candidate, ``U[0, 1)`` per unknown (reference
    optimize.py:132, 174-178).

    ``key``: an int seed or a raw Threefry key (``differt2d_amd.random.PRNGKey``) -- the draws are then the reference's own
    (``jax.random``'s Threefry, restated in differt2d_amd/random.py): with ``per_candidate_keys`` the key is first split into
    one key per candidate, as the grid sweeps do (scene.py:1585, 1888); else it is the key of the single path
    (``from_tx_objects_rx``).  A ``numpy.random.Generator`` draws the rows in order from NumPy's PRNG instead.  Pass
    ``theta0`` to skip the draw."""
    from . import random as jr

    counts = [sum(o.parameters_count() for o in objs) for objs in objects_per_candidate]
    if theta0 is not None:
        rows = [np.asarray(r, F).reshape(-1) for r in theta0]
        need = [c for c in counts for _ in range(many)]
        if len(rows) != len(need) or any(r.size < c for r, c in zip(rows, need)):
            raise ValueError("theta0 must hold `many` rows per candidate with at least as many values as unknowns")
        return rows
    if key is None:
        raise TypeError("this path class needs a `key` (or explicit `theta0`) to draw its initial guess")
    if isinstance(key, np.random.Generator):
        return [key.random(c, dtype=F) for c in counts for _ in range(many)]
    key = jr.as_key(key)
    keys = jr.split(key, len(counts)) if per_candidate_keys else [key] * len(counts)
    if not per_candidate_keys and len(counts) != 1:
        raise ValueError("one key per path: split it per candidate (per_candidate_keys)")
    rows = []
    for k, c in zip(keys, counts):
        rows.extend(_theta0_rows(k, c, many))
    return rows


def _opt_kwargs(kwargs):
    """(steps, many, theta0, optimizer) from ``path_cls_kwargs`` (reference optimize.py:44-52, 136-143).  ``optimizer``: None
    (the reference's default, ``optax.adam(0.1)``) or ``differt2d_amd.optimize.adam(learning_rate, b1, b2, eps)``."""
    from .optimize import Adam

    kw = dict(kwargs)
    steps = int(kw.pop("steps", 100))
    many = int(kw.pop("many", 1))
    theta0 = kw.pop("theta0", None)
    if many < 1:
        raise ValueError("many must be >= 1")
    optimizer = kw.pop("optimizer", None)
    if optimizer is not None and not isinstance(optimizer, Adam):
        raise L.D2DUnsupported(-4, f"optimizer {optimizer!r} is not native: differt2d_amd.optimize.adam(learning_rate, b1, b2, eps) is "
                                   "(the reference's default is optax.adam(0.1))")
    if kw:
        raise TypeError(f"unexpected keyword arguments: {sorted(kw)}")
    return steps, many, theta0, optimizer


class _OptPath(Path):
    """Shared driver of the optimiser-based solvers (GPU: d2d_trace_paths with solver != image)."""

    @classmethod
    def from_tx_objects_rx(cls, tx, objects, rx, *, key=None, **kwargs):
        from .engine import default_context, make_params

        objects = list(objects)
        k = len(objects)
        steps, many, theta0, optimizer = _opt_kwargs(kwargs)
        if theta0 is not None and many == 1 and np.ndim(theta0) == 1:
            theta0 = [theta0]
        th = draw_theta0([objects], key, theta0, many) if k else [np.zeros(0, F)] * many
        ctx = default_context()
        ctx.set_optimizer(optimizer)
        ctx.set_scene(*objects_to_tables(objects))
        p = make_params(max_order=L.D2D_MAX_ORDER, solver=cls.solver, steps=steps, many=many)
        out = ctx.trace_paths(p, _xy(tx), _xy(rx), [np.arange(k, dtype=np.int32)], theta0=th)
        return cls(xys=out["xys"][0, 0, : k + 2], loss=out["loss"][0, 0])


class FermatPath(_OptPath):
    """Path minimising its length (reference geometry.py:1117-1204): Adam on the parametric coordinates, on the GPU."""

    solver = "fermat"


class MinPath(_OptPath):
    """Path minimising the sum of interaction losses, Min-Path-Tracing (reference geometry.py:1207-1288)."""

    solver = "min"
