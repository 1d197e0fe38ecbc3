"""
``Scene`` with the reference's constructors, container semantics and accumulate entry points
(``differt2d/scene.py``), NumPy in / NumPy out, computing on an MI355X through ``libd2d.so``.

What runs where
---------------
* container methods (``with_*``, ``add_objects``, canned scenes ...) are plain host code;
* candidate enumeration is host-native C++ (``d2d_enumerate_candidates``), as it is host-native Rust in
  the reference;
* ``accumulate_on_receivers_grid_over_paths`` with a natively fused ``fun`` (``utils.received_power``,
  ``utils.path_length_squared`` ...) is ONE fused kernel launch per transmitter (``d2d_power_map_launch``);
* ``all_paths`` / ``all_valid_paths`` / ``accumulate_over_paths`` and grid sweeps with an arbitrary Python
  ``fun`` trace every (pair, candidate) on the GPU (``d2d_trace_paths``) and call ``fun`` on the host.

No CPU implementation of the path solve / validity exists in this package.
"""

from __future__ import annotations

import dataclasses
import weakref
import json
import operator
from collections.abc import Iterator, Mapping, Sequence
from itertools import groupby, product
from typing import Any, Callable, Optional, Union

import numpy as np

from . import _lib as L
from . import logic
from .abc import Object, Plottable, random_uniform
from .engine import Context, default_context, make_params
from .geometry import (
    FermatPath, ImagePath, MinPath, Path, Point, RIS, Vertex, Wall, _opt_kwargs, _validity_kwargs, closest_point,
    draw_theta0, objects_to_tables, stack_leaves, unstack_leaves,
)

__all__ = ("Scene", "SceneName", "PyTreeDict", "all_path_candidates")

F = np.float32
SceneName = str
PathFun = Callable[..., Any]

#: grid cells x candidates above which a non-native ``fun`` is refused (host memory / time)
EMIT_LIMIT = 8_000_000


class PyTreeDict(Mapping):
    """Immutable ordered mapping (reference scene.py:72-119); indexing is linear in its size."""

    def __init__(self, _keys=(), _values=()):
        self._keys = tuple(_keys)
        self._values = tuple(_values)
        if len(self._keys) != len(self._values):
            raise ValueError(
                f"Number of keys must match number of values, got {len(self._keys)} and {len(self._values)}."
            )

    @classmethod
    def from_mapping(cls, mapping: Mapping) -> "PyTreeDict":
        return cls(_keys=mapping.keys(), _values=mapping.values())

    def __getitem__(self, key):
        try:
            return self._values[self._keys.index(key)]
        except ValueError as e:
            raise KeyError(key) from e

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def __repr__(self):
        return f"PyTreeDict({dict(self)!r})"


def all_path_candidates(num_nodes: int, min_order: int = 0, max_order: int = 1, *, order: Optional[int] = None,
                        filter_nodes: Optional[Sequence[int]] = None) -> list:
    """All path candidates as a list of int32 arrays (reference scene.py:122-175): for each order ascending,
    every tuple of object indices with no two equal neighbours, lexicographic; ``filter_nodes`` are never
    visited."""
    if order is not None:
        min_order = max_order = order
    allowed = None
    if filter_nodes is not None:
        allowed = np.ones(num_nodes, np.uint8)
        allowed[list(filter_nodes)] = 0
    return L.enumerate_candidates(num_nodes, min_order, max_order, allowed)


def _probe_paths():
    """Synthetic (transmitter, receiver, path, interacting objects) tuples of orders 0..3 with incommensurate lengths -- each
    path TWICE: once with plain end points and unit walls through its interaction points, once with end points that are not the
    path's, and objects of other kinds, lengths and orientations.  A path function that is one of the closed forms depends on
    the path alone and returns the same number for both."""
    from .geometry import RIS, Vertex, Wall

    rng = np.random.default_rng(20260)
    out = []
    for k in (0, 1, 1, 2, 3, 0, 2, 1, 3):
        pts = (rng.random((k + 2, 2)) * F(3.0) - F(1.0)).astype(F)
        walls = [Wall(xys=np.stack([pts[i + 1] - F(0.5), pts[i + 1] + F(0.5)])) for i in range(k)]
        other = []
        for i in range(k):
            d = (rng.random(2) * F(2.0) - F(1.0)).astype(F) * F(0.25 + 2.0 * rng.random())
            kind = (RIS, Vertex, Wall)[i % 3]
            other.append(Vertex(xy=pts[i + 1]) if kind is Vertex else kind(xys=np.stack([pts[i + 1] - d, pts[i + 1] + F(0.3) * d])))
        shift = (rng.random(2) * F(0.7) + F(0.1)).astype(F)
        out.append(((Point(xy=pts[0]), Point(xy=pts[-1]), Path(xys=pts), walls),
                    (Point(xy=pts[0] + shift), Point(xy=pts[-1] - shift), Path(xys=pts), other)))
    return out


_FUN_VERDICTS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _recognise_fun(fun, fun_args, fun_kwargs):
    """A user callable that IS one of the natively fused closed forms -- the reference's own tests pass a local
    ``fun`` returning ``path.length() ** 2`` (tests/test_scene.py:444, 558-560) -- is recognised by what it computes: it is
    evaluated on a handful of synthetic paths and compared with ``1``, ``length``, ``length ** 2`` and the
    ``r_coef ** n / (height ** 2 + length ** 2)`` family (utils.py:17-54; the two constants are fitted from the first probes
    and verified on the others).  Returns (native name, make_params kwargs) or None; a callable that raises on the
    probes, depends on anything but the path's length and order -- the end points it is handed, the kinds, lengths or
    orientations of the interacting objects: every path is probed with two different sets of them -- or matches nothing is
    left to the host.  The verdict is kept per callable (and arguments), so the full probe set runs once; a cached POSITIVE
    verdict is re-checked on every call against three of the probes (orders 0, 1, 2 -- enough to pin both fitted constants): a
    callable that reads a global, a closure cell or a mutable attribute (a parameter sweep over ``height``) and changed since
    is probed afresh instead of being run with the stale fit.  ``fun._d2d_native = False`` opts a callable out."""
    try:
        ckey = (tuple(fun_args), tuple(sorted((fun_kwargs or {}).items())))
        hash(ckey)
        cached = _FUN_VERDICTS.get(fun, {}).get(ckey, "?")
        if cached is None or (cached != "?" and _verdict_still_holds(fun, fun_args, fun_kwargs, cached)):
            return cached
    except TypeError:  # unhashable arguments, or a callable that cannot be weakly referenced: probe every time
        ckey = None
    verdict = _recognise_fun_uncached(fun, fun_args, fun_kwargs)
    if ckey is not None:
        try:
            _FUN_VERDICTS.setdefault(fun, {})[ckey] = verdict
        except TypeError:
            pass
    return verdict


_PROBES = None


def _closed_form(name, extra, r, k):
    if name == "one":
        return 1.0
    if name == "length":
        return r
    if name == "length_squared":
        return r * r
    return extra.get("r_coef", 0.5) ** k / (extra.get("height", 0.1) ** 2 + r * r)


def _verdict_still_holds(fun, fun_args, fun_kwargs, verdict) -> bool:
    global _PROBES
    if _PROBES is None:
        _PROBES = _probe_paths()
    name, extra = verdict
    seen = set()
    try:
        for pair in _PROBES:
            a, b, path, inter = pair[0]
            k = path.xys.shape[0] - 2
            if k in seen or k > 2:
                continue
            seen.add(k)
            v = float(np.asarray(fun(a, b, path, inter, *fun_args, **(fun_kwargs or {})), dtype=np.float64))
            want = _closed_form(name, extra, float(path.length()), k)
            if not np.isfinite(v) or abs(v - want) > 2e-6 * abs(want):
                return False
    except Exception:  # noqa: BLE001
        return False
    return True


def _recognise_fun_uncached(fun, fun_args, fun_kwargs):
    probes = _probe_paths()
    try:
        both = [[np.asarray(fun(a, b, path, inter, *fun_args, **(fun_kwargs or {})), dtype=np.float64) for a, b, path, inter in pair]
                for pair in probes]
    except Exception:  # noqa: BLE001 -- whatever the callable needs, the probes do not provide it
        return None
    if any(v.shape != () or not np.isfinite(v) for pair in both for v in pair):
        return None
    if any(float(u) != float(v) for u, v in both):
        return None  # depends on the end points or on the objects, not on the path alone
    probes = [pair[0] for pair in probes]
    vals = np.array([float(pair[0]) for pair in both])
    r = np.array([float(path.length()) for _, _, path, _ in probes])
    ks = np.array([path.xys.shape[0] - 2 for _, _, path, _ in probes])
    close = lambda want: bool(np.allclose(vals, want, rtol=2e-6, atol=0.0))
    if close(np.ones_like(r)):
        return "one", {}
    if close(r):
        return "length", {}
    if close(r * r):
        return "length_squared", {}
    # r_coef ** k / (h2 + r * r): h2 from the first order-0 probe, r_coef from the first order-1 probe
    i0, i1 = int(np.argmax(ks == 0)), int(np.argmax(ks == 1))
    if vals[i0] > 0.0:
        h2 = 1.0 / vals[i0] - r[i0] ** 2
        if h2 > 0.0:
            r_coef = vals[i1] * (h2 + r[i1] ** 2)
            if np.isfinite(r_coef) and close(r_coef ** ks / (h2 + r * r)):
                # the fit carries the probes' fp32 round-off: prefer the short decimals the caller almost certainly wrote
                height = float(np.sqrt(h2))
                for digits in (2, 3, 4, 5):
                    rc, hh = float(f"{r_coef:.{digits}g}"), float(f"{height:.{digits}g}")
                    if close(rc ** ks / (hh * hh + r * r)):
                        return "received_power", {"r_coef": rc, "height": hh}
                return "received_power", {"r_coef": float(r_coef), "height": height}
    return None


def _native_fun(fun, fun_args, fun_kwargs):
    """(fun name, kwargs for make_params) if ``fun`` is fused natively, else None."""
    name = getattr(fun, "_d2d_native", None)
    if name is False:
        return None  # opted out: the host evaluates it
    if name is None:
        return _recognise_fun(fun, fun_args, fun_kwargs) if callable(fun) else None
    if fun_args:
        return None
    extra = dict(fun_kwargs or {})
    if name != "received_power" and extra:
        return None
    if set(extra) - {"r_coef", "height"}:
        return None
    return name, extra


@dataclasses.dataclass(frozen=True, eq=False)
class Scene(Plottable):
    """2-D scene: named transmitters, named receivers and a sequence of objects (reference scene.py:178-192)."""

    transmitters: Mapping = dataclasses.field(default_factory=lambda: PyTreeDict())
    receivers: Mapping = dataclasses.field(default_factory=lambda: PyTreeDict())
    objects: Sequence = ()

    def __post_init__(self):
        object.__setattr__(self, "transmitters", PyTreeDict.from_mapping(self.transmitters))
        object.__setattr__(self, "receivers", PyTreeDict.from_mapping(self.receivers))
        object.__setattr__(self, "objects", tuple(self.objects))

    # ---------------------------------------------------------------- container methods
    def with_transmitters(self, **transmitters: Point) -> "Scene":
        return dataclasses.replace(self, transmitters=transmitters)

    def with_receivers(self, **receivers: Point) -> "Scene":
        return dataclasses.replace(self, receivers=receivers)

    def with_objects(self, *objects: Object) -> "Scene":
        return dataclasses.replace(self, objects=tuple(objects))

    def filter_objects(self, filter_spec: Callable[[Object], bool]) -> "Scene":
        return dataclasses.replace(self, objects=tuple(filter(filter_spec, self.objects)))

    def update_transmitters(self, **transmitters: Point) -> "Scene":
        return dataclasses.replace(self, transmitters={**self.transmitters, **transmitters})

    def update_receivers(self, **receivers: Point) -> "Scene":
        return dataclasses.replace(self, receivers={**self.receivers, **receivers})

    def add_objects(self, *objects: Object) -> "Scene":
        return self.with_objects(*self.objects, *objects)

    def get_object(self, index) -> Object:
        """Object at ``index`` (clamped like ``lax.switch``); homogeneous scenes only (reference scene.py:330-345)."""
        if any(type(o) is not type(self.objects[0]) for o in self.objects):
            raise TypeError("get_object needs objects of one type (reference: lax.switch raises)")
        return self.objects[int(np.clip(int(index), 0, len(self.objects) - 1))]

    def stacked_objects(self):
        return stack_leaves(self.objects)

    def rename_transmitters(self, **names: str) -> "Scene":
        return self.with_transmitters(**{names.get(k, k): v for k, v in self.transmitters.items()})

    def rename_receivers(self, **names: str) -> "Scene":
        return self.with_receivers(**{names.get(k, k): v for k, v in self.receivers.items()})

    @classmethod
    def from_stacked_objects(cls, objects) -> "Scene":
        return cls(transmitters={}, receivers={}, objects=unstack_leaves(objects))

    @classmethod
    def from_walls_array(cls, walls) -> "Scene":
        """Empty scene from an array ``[num_walls, 2, 2]`` (reference scene.py:413-426)."""
        return cls(transmitters={}, receivers={}, objects=[Wall(xys=xys) for xys in np.asarray(walls, dtype=F)])

    @classmethod
    def from_geojson(cls, s_or_fp, tx_loc: str = "NW", rx_loc: str = "SE") -> "Scene":
        """Scene from a GeoJSON document (string-like or file-like): one ``Wall`` per consecutive pair of points of
        every ``Polygon`` feature's outer ring, wrapping around (so a closed ring also yields one zero-length wall,
        as in the reference); TX / RX sit on corners of the bounding box (reference scene.py:428-668)."""
        if hasattr(s_or_fp, "read"):
            s_or_fp = s_or_fp.read()
        if not isinstance(s_or_fp, (str, bytes, bytearray)):
            raise NotImplementedError(f"Unsupported type {type(s_or_fp)}")
        document = json.loads(s_or_fp)
        walls = []
        for feature in document.get("features", []):
            geometry = feature.get("geometry", None)
            if not geometry or geometry["type"] != "Polygon":
                continue
            ring = geometry["coordinates"][0]
            for i in range(len(ring)):
                walls.append(Wall(xys=np.array([ring[i - 1], ring[i]], dtype=F)))
        scene = cls(objects=walls)
        if walls:
            return scene.with_transmitters(tx=Point(xy=scene.get_location(tx_loc))).with_receivers(
                rx=Point(xy=scene.get_location(rx_loc)))
        return scene.with_transmitters(tx=Point(xy=[0.0, 0.0])).with_receivers(rx=Point(xy=[1.0, 1.0]))

    @classmethod
    def from_scene_name(cls, scene_name: SceneName, *args, **kwargs) -> "Scene":
        return getattr(cls, scene_name)(*args, **kwargs)

    # ---------------------------------------------------------------------- canned scenes
    @classmethod
    def random_uniform_scene(cls, n_transmitters: int = 1, n_walls: int = 1, n_receivers: int = 1, *, key) -> "Scene":
        """Random scene with the reference's layout (scene.py:718-733): one uniform draw of
        ``n_transmitters + 2 n_walls + n_receivers`` points -- the reference's own numbers for an int seed / Threefry key
        (``differt2d_amd.random.PRNGKey(1234)`` = ``jax.random.PRNGKey(1234)``)."""
        pts = random_uniform(key, (n_transmitters + 2 * n_walls + n_receivers, 2))
        txs = {f"tx_{i}": Point(xy=pts[i, :]) for i in range(n_transmitters)}
        rxs = {f"rx_{i}": Point(xy=pts[-(i + 1), :]) for i in range(n_receivers)}
        walls = [Wall(xys=pts[2 * i + n_transmitters : 2 * i + 2 + n_transmitters, :]) for i in range(n_walls)]
        return cls(transmitters=txs, receivers=rxs, objects=walls)

    @classmethod
    def basic_scene(cls, tx_coords=(0.1, 0.1), rx_coords=(0.302, 0.2147)) -> "Scene":
        """Main room plus an inner room with a small entrance (reference scene.py:735-787)."""
        coords = [
            [[0.0, 0.0], [1.0, 0.0]], [[1.0, 0.0], [1.0, 1.0]], [[1.0, 1.0], [0.0, 1.0]], [[0.0, 1.0], [0.0, 0.0]],
            [[0.4, 0.0], [0.4, 0.4]], [[0.4, 0.4], [0.3, 0.4]], [[0.1, 0.4], [0.0, 0.4]],
        ]
        return cls(transmitters={"tx": Point(xy=tx_coords)}, receivers={"rx": Point(xy=rx_coords)},
                   objects=[Wall(xys=c) for c in coords])

    @classmethod
    def square_scene(cls, tx_coords=(0.2, 0.2), rx_coords=(0.5, 0.6)) -> "Scene":
        """One square room (reference scene.py:789-836)."""
        coords = [[[0.0, 0.0], [1.0, 0.0]], [[1.0, 0.0], [1.0, 1.0]], [[1.0, 1.0], [0.0, 1.0]], [[0.0, 1.0], [0.0, 0.0]]]
        return cls(transmitters={"tx": Point(xy=tx_coords)}, receivers={"rx": Point(xy=rx_coords)},
                   objects=[Wall(xys=c) for c in coords])

    @classmethod
    def square_scene_with_wall(cls, ratio: float = 0.6, tx_coords=(0.2, 0.5), rx_coords=(0.8, 0.5)) -> "Scene":
        """Square room with a vertical wall in the middle (reference scene.py:838-882)."""
        scene = cls.square_scene(tx_coords=tx_coords, rx_coords=rx_coords)
        return scene.add_objects(Wall(xys=[[0.5, 0.5 * (1 - ratio)], [0.5, 0.5 * (1 + ratio)]]))

    @classmethod
    def square_scene_with_obstacle(cls, ratio: float = 0.1, **kwargs) -> "Scene":
        """Square room with a square obstacle in its centre (reference scene.py:884-935)."""
        scene = cls.square_scene(**kwargs)
        hl = 0.5 * ratio
        x0, x1, y0, y1 = 0.5 - hl, 0.5 + hl, 0.5 - hl, 0.5 + hl
        return scene.add_objects(
            Wall(xys=[[x0, y0], [x1, y0]]), Wall(xys=[[x1, y0], [x1, y1]]),
            Wall(xys=[[x1, y1], [x0, y1]]), Wall(xys=[[x0, y1], [x0, y0]]),
        )

    # ------------------------------------------------------------------- plotting / extents
    def plot(self, ax, *args, transmitters: bool = True, transmitters_args=(), transmitters_kwargs=None,
             objects: bool = True, objects_args=(), objects_kwargs=None, receivers: bool = True, receivers_args=(),
             receivers_kwargs=None, annotate: bool = True, **kwargs):
        """Draws transmitters, objects and receivers (reference scene.py:937-1021)."""
        transmitters_kwargs = {**kwargs, **(transmitters_kwargs or {})}
        objects_kwargs = {**kwargs, **(objects_kwargs or {})}
        receivers_kwargs = {**kwargs, **(receivers_kwargs or {})}
        artists = []
        if transmitters:
            for name, tx in self.transmitters.items():
                kw = dict(transmitters_kwargs)
                if annotate:
                    kw.setdefault("annotate", name)
                artists.extend(tx.plot(ax, *args, *transmitters_args, **kw))
        if objects:
            for obj in self.objects:
                artists.extend(obj.plot(ax, *args, *objects_args, **objects_kwargs))
        if receivers:
            for name, rx in self.receivers.items():
                kw = dict(receivers_kwargs)
                if annotate:
                    kw.setdefault("annotate", name)
                artists.extend(rx.plot(ax, *args, *receivers_args, **kw))
        return artists

    def bounding_box(self) -> np.ndarray:
        boxes = ([t.bounding_box() for t in self.transmitters.values()] + [r.bounding_box() for r in self.receivers.values()]
                 + [o.bounding_box() for o in self.objects])
        boxes = np.stack(boxes)
        return np.vstack([np.min(boxes[:, 0, :], axis=0), np.max(boxes[:, 1, :], axis=0)]).astype(F)

    def get_closest_transmitter(self, coords):
        items = list(self.transmitters.items())
        i, d = closest_point(np.vstack([p.xy for _, p in items]), coords)
        return items[int(i)][0], d

    def get_closest_receiver(self, coords):
        items = list(self.receivers.items())
        i, d = closest_point(np.vstack([p.xy for _, p in items]), coords)
        return items[int(i)][0], d

    def all_transmitter_receiver_pairs(self):
        return product(self.transmitters.items(), self.receivers.items())

    # ------------------------------------------------------------------- path candidates
    def _allowed_mask(self, filter_objects):
        if filter_objects is None:
            return None
        return np.array([1 if filter_objects(o) else 0 for o in self.objects], np.uint8)

    def all_path_candidates(self, min_order: int = 0, max_order: int = 1, *, order: Optional[int] = None,
                            filter_objects: Optional[Callable[[Object], bool]] = None) -> list:
        """Reference scene.py:1089-1134."""
        if order is not None:
            min_order = max_order = order
        return L.enumerate_candidates(len(self.objects), min_order, max_order, self._allowed_mask(filter_objects))

    def get_interacting_objects(self, path_candidate) -> list:
        """Objects a candidate visits, in order (reference scene.py:1136-1154)."""
        return [self.objects[int(i)] for i in path_candidate]

    # --------------------------------------------------------------------------- GPU glue
    @staticmethod
    def _ctx(device: int = 0) -> Context:
        return default_context(device)

    def _upload(self, ctx: Context, filter_objects=None):
        ctx.set_scene(*objects_to_tables(self.objects))
        mask = self._allowed_mask(filter_objects)
        if mask is not None:
            ctx.set_candidate_mask(mask)

    @staticmethod
    def _solver_of(path_cls) -> str:
        solver = getattr(path_cls, "solver", None)
        if solver not in ("image", "min", "fermat"):
            raise L.D2DUnsupported(-4, f"path_cls={getattr(path_cls, '__name__', path_cls)} has no native solver "
                                       "(ImagePath, MinPath and FermatPath have)")
        return solver

    def _solver_setup(self, path_cls, path_cls_kwargs, candidates, key):
        """(extra make_params kwargs, theta0 rows or None) for a path class."""
        solver = self._solver_of(path_cls)
        if solver == "image":
            if path_cls_kwargs:
                raise TypeError(f"ImagePath takes no path_cls_kwargs, got {sorted(path_cls_kwargs)}")
            return dict(solver=solver), None
        steps, many, theta0, optimizer = _opt_kwargs(path_cls_kwargs or {})
        rows = draw_theta0([self.get_interacting_objects(c) for c in candidates], key, theta0, many, per_candidate_keys=True)
        self._ctx().set_optimizer(optimizer)  # (every optimiser-based call says which: nothing stale from an earlier one)
        return dict(solver=solver, steps=steps, many=many), rows

    def _trace(self, pairs_tx, pairs_rx, candidates, path_cls, path_cls_kwargs, key, validity):
        """GPU trace of every candidate for every (tx, rx) pair -> dict of arrays, leading shape (P, C).

        Optimiser-based path classes with a Threefry key: the reference hands every (pair, candidate) its own key from a chain
        of splits -- ``key, key_path = split(key, 2)`` in pair-major order (scene.py:1204-1219) -- so every pair is traced with
        its own initial guesses."""
        from . import random as jr
        from .geometry import _theta0_rows

        ctx = self._ctx()
        self._upload(ctx)
        solver = self._solver_of(path_cls)
        kw = dict(path_cls_kwargs or {})
        chain = solver != "image" and key is not None and not isinstance(key, np.random.Generator) and kw.get("theta0") is None
        if not chain:
            extra, theta0 = self._solver_setup(path_cls, path_cls_kwargs, candidates, key)
            params = make_params(max_order=L.D2D_MAX_ORDER, **extra, **validity)
            return ctx.trace_paths(params, pairs_tx, pairs_rx, candidates, theta0=theta0)
        steps, many, _, optimizer = _opt_kwargs(kw)
        ctx.set_optimizer(optimizer)
        params = make_params(max_order=L.D2D_MAX_ORDER, solver=solver, steps=steps, many=many, **validity)
        counts = [sum(o.parameters_count() for o in self.get_interacting_objects(c)) for c in candidates]
        key = jr.as_key(key)
        outs = []
        for p in range(len(pairs_tx)):
            rows = []
            for c in counts:
                key, key_path = jr.split(key, 2)
                rows.extend(_theta0_rows(key_path, c, many))
            outs.append(ctx.trace_paths(params, pairs_tx[p : p + 1], pairs_rx[p : p + 1], candidates, theta0=rows))
        return {k: np.concatenate([o[k] for o in outs], axis=0) for k in outs[0]}

    # ------------------------------------------------------------------- individual paths
    def all_paths(self, path_cls: type = ImagePath, path_cls_kwargs: Optional[Mapping] = None, min_order: int = 0,
                  max_order: int = 1, order: Optional[int] = None, filter_objects=None, *, key=None, **kwargs) -> Iterator:
        """Yields ``(tx name, rx name, valid, path, path_candidate)`` for every pair and candidate
        (reference scene.py:1156-1228)."""
        validity = _validity_kwargs(**kwargs)
        candidates = self.all_path_candidates(min_order=min_order, max_order=max_order, order=order, filter_objects=filter_objects)
        pairs = list(self.all_transmitter_receiver_pairs())
        if not pairs or not candidates:
            return
        txs = np.stack([t.xy for (_, t), _ in pairs])
        rxs = np.stack([r.xy for _, (_, r) in pairs])
        out = self._trace(txs, rxs, candidates, path_cls, path_cls_kwargs, key, validity)
        hard = not validity["approx"]
        for p, ((tx_key, _), (rx_key, _)) in enumerate(pairs):
            for c, cand in enumerate(candidates):
                k = len(cand)
                valid = out["valid"][p, c]
                path = path_cls(xys=out["xys"][p, c, : k + 2], loss=out["loss"][p, c])
                yield tx_key, rx_key, (np.bool_(valid != 0) if hard else valid), path, cand

    def all_valid_paths(self, approx: Optional[bool] = None, **kwargs) -> Iterator:
        """Only the paths for which ``is_true(valid)`` (reference scene.py:1230-1248)."""
        for tx_key, rx_key, valid, path, cand in self.all_paths(approx=approx, **kwargs):
            if logic.is_true(valid, approx=approx):
                yield tx_key, rx_key, path, cand

    def _pairwise_fused(self, fun, fun_args, fun_kwargs, kwargs):
        """The arguments of a fused pairwise sweep -- the receivers of the scene as a 1 x R grid, one launch per transmitter --
        or None when ``fun`` is not one of the natively fused closed forms (then: GPU trace + host ``fun``).

        ``"launches"``: (tx name, tx, receiver names, X, Y, theta0 rows or None) per launch.  Optimiser-based path classes with
        a Threefry key and no explicit ``theta0``: the reference hands every (pair, candidate) its own key from a chain of
        splits in pair-major order (scene.py:1204-1219, what ``_trace`` reproduces), so every pair is its own 1 x 1 launch with
        its own initial guesses -- the same numbers whether or not ``fun`` is recognised.  No candidates at all (order = 2 in
        a scene of one object): no launches, the iterator is empty as the reference's groupby over no paths is."""
        from . import random as jr
        from .geometry import _theta0_rows

        kwargs = dict(kwargs)
        path_cls = kwargs.pop("path_cls", ImagePath)
        path_cls_kwargs = kwargs.pop("path_cls_kwargs", None)
        min_order, max_order, order = kwargs.pop("min_order", 0), kwargs.pop("max_order", 1), kwargs.pop("order", None)
        filter_objects, key = kwargs.pop("filter_objects", None), kwargs.pop("key", None)
        native, common = self._sweep_params(fun, fun_args, fun_kwargs, path_cls, path_cls_kwargs, min_order, max_order, order, kwargs)
        if native is None or not self.receivers or not self.transmitters:
            return None
        name, extra = native
        solver = self._solver_of(path_cls)
        if solver == "image":
            # (ImagePath draws nothing per candidate: only WHETHER there are candidates matters here -- counted in the library
            # instead of building N^K index arrays in Python per call)
            lo, hi = (order, order) if order is not None else (min_order, max_order)
            cands = [None] * min(1, L.count_candidates(len(self.objects), lo, hi, self._allowed_mask(filter_objects)))
        else:
            cands = self.all_path_candidates(min_order, max_order, order=order, filter_objects=filter_objects)
        rx_keys = list(self.receivers)
        rx = np.stack([r.xy for r in self.receivers.values()]).astype(F)
        pkw = dict(path_cls_kwargs or {})
        chain = solver != "image" and key is not None and not isinstance(key, np.random.Generator) and pkw.get("theta0") is None
        launches = []
        if not cands:
            sextra = dict(solver=solver)
        elif chain:
            steps, many, _, optimizer = _opt_kwargs(pkw)
            self._ctx().set_optimizer(optimizer)
            sextra = dict(solver=solver, steps=steps, many=many)
            counts = [sum(o.parameters_count() for o in self.get_interacting_objects(c)) for c in cands]
            key = jr.as_key(key)
            for tx_key, tx in self.transmitters.items():  # (all_transmitter_receiver_pairs: transmitter-major)
                for j, rx_key in enumerate(rx_keys):
                    rows = []
                    for c in counts:
                        key, key_path = jr.split(key, 2)
                        rows.extend(_theta0_rows(key_path, c, many))
                    launches.append((tx_key, tx, [rx_key], np.ascontiguousarray(rx[None, j : j + 1, 0]),
                                     np.ascontiguousarray(rx[None, j : j + 1, 1]), rows))
        else:
            sextra, theta0 = self._solver_setup(path_cls, path_cls_kwargs, cands if solver != "image" else [], key)
            X, Y = np.ascontiguousarray(rx[None, :, 0]), np.ascontiguousarray(rx[None, :, 1])
            launches = [(tx_key, tx, rx_keys, X, Y, theta0) for tx_key, tx in self.transmitters.items()]
        return dict(params={"fun": name, **extra, **sextra, **common}, filter_objects=filter_objects, launches=launches)

    def accumulate_over_paths(self, fun: PathFun, fun_args: tuple = (), fun_kwargs: Optional[Mapping] = None, *,
                              reduce_all: bool = False, **kwargs):
        """Sum of ``valid * fun(...)`` over all candidates, per (tx, rx) pair (reference scene.py:1272-1334).  A natively
        fused ``fun`` runs as one fused launch per transmitter over the receivers (a 1 x R grid): the reference's sequential
        fp32 sum in candidate order, as everywhere; any other callable is evaluated on the host from the traced paths."""
        fun_kwargs = dict(fun_kwargs or {})
        fused = self._pairwise_fused(fun, fun_args, fun_kwargs, kwargs)

        def fused_results():
            ctx = self._ctx()
            params = make_params(**fused["params"])
            for tx_key, tx, rx_keys, X, Y, theta0 in fused["launches"]:
                self._upload(ctx, fused["filter_objects"])
                ctx.set_grid(X, Y)
                if theta0 is not None:
                    ctx.set_theta0(theta0)
                ctx.launch(params, tx.xy)
                row = ctx.get_map()[0]
                for j, rx_key in enumerate(rx_keys):
                    yield tx_key, rx_key, F(row[j])

        def results():
            for (tx_key, rx_key), group in groupby(self.all_paths(**kwargs), operator.itemgetter(slice(2))):
                acc = F(0.0)
                tx, rx = self.transmitters[tx_key], self.receivers[rx_key]
                for _, _, valid, path, cand in group:
                    inter = self.get_interacting_objects(cand)
                    acc = F(acc + F(valid) * F(fun(tx, rx, path, inter, *fun_args, **fun_kwargs)))
                yield tx_key, rx_key, acc

        gen = fused_results if fused is not None else results
        if reduce_all:
            Z = F(0.0)
            for _, _, p in gen():
                Z = F(Z + p)
            return Z
        return gen()

    def accumulate_over_paths_value_and_vjp(self, fun: PathFun, fun_args: tuple = (), fun_kwargs: Optional[Mapping] = None, *,
                                            cotangent=None, **kwargs):
        """``accumulate_over_paths`` with its reverse-mode derivatives -- what users of the reference obtain by wrapping the call
        in ``jax.value_and_grad`` (examples/plot_power_optimize.py:78-93: ``loss(tx_coords, scene)`` over
        ``scene.accumulate_over_paths(...)``).  ``fun`` must be natively fused.

        ``cotangent``: the derivative of the caller's scalar objective w.r.t. each accumulated value -- a mapping
        ``{(tx name, rx name): weight}`` (missing pairs: 0), or a callable that receives the values' mapping and returns such a
        mapping (so that ``objective(values)``'s own derivative can be formed in between: one forward launch, one reverse
        launch); default: ones, i.e. the gradient of the ``reduce_all`` sum.

        Returns ``(values, vjp)``: ``values[(tx name, rx name)]`` as ``accumulate_over_paths`` yields them, and ``vjp`` with
        ``"transmitters"[tx name]`` = d objective / d tx.xy, ``"receivers"[rx name]`` = d objective / d rx.xy, ``"objects"`` =
        d objective / d xys ``[N, 2, 2]`` and ``"phi"`` ``[N]`` (RIS angles; MinPath / FermatPath sweeps)."""
        fun_kwargs = dict(fun_kwargs or {})
        fused = self._pairwise_fused(fun, fun_args, fun_kwargs, kwargs)
        if fused is None:
            raise L.D2DUnsupported(-4, "the VJP needs transmitters, receivers and a natively fused fun (differt2d_amd.utils): an "
                                       "arbitrary Python callable cannot be differentiated by the hand-derived kernels")
        ctx = self._ctx()
        rx_keys = list(self.receivers)
        params = make_params(**fused["params"])
        values = {}
        if callable(cotangent) or cotangent is None:
            for tx_key, tx, keys, X, Y, theta0 in fused["launches"]:
                self._upload(ctx, fused["filter_objects"])
                ctx.set_grid(X, Y)
                if theta0 is not None:
                    ctx.set_theta0(theta0)
                ctx.launch(params, tx.xy)
                row = ctx.get_map()[0]
                values.update({(tx_key, k): F(row[j]) for j, k in enumerate(keys)})
        cot = cotangent(dict(values)) if callable(cotangent) else cotangent
        n = len(self.objects)
        vjp = {"transmitters": {k: np.zeros(2, F) for k in self.transmitters}, "receivers": {k: np.zeros(2, F) for k in rx_keys},
               "objects": np.zeros((n, 2, 2), F), "phi": np.zeros(n, F)}
        for tx_key, tx, keys, X, Y, theta0 in fused["launches"]:
            w = np.ones((1, len(keys)), F) if cot is None else np.array([[cot.get((tx_key, k), 0.0) for k in keys]], F)
            self._upload(ctx, fused["filter_objects"])
            ctx.set_grid(X, Y)
            ctx.set_cotangent(w)
            if theta0 is not None:
                ctx.set_theta0(theta0)
            ctx.launch_vg(params, tx.xy, scene_vjp=True)
            row, g = ctx.get_map()[0], ctx.get_grad_rx()[0]
            tx_bar, objects_bar, phi_bar = ctx.get_scene_vjp(with_phi=True)
            values.update({(tx_key, k): F(row[j]) for j, k in enumerate(keys)})
            vjp["transmitters"][tx_key] = (vjp["transmitters"][tx_key] + tx_bar).astype(F)
            for j, k in enumerate(keys):
                vjp["receivers"][k] = (vjp["receivers"][k] + w[0, j] * g[j]).astype(F)
            vjp["objects"] = (vjp["objects"] + objects_bar).astype(F)
            vjp["phi"] = (vjp["phi"] + phi_bar).astype(F)
        return values, vjp

    # ------------------------------------------------------------------------ grid sweeps
    def _sweep_params(self, fun, fun_args, fun_kwargs, path_cls, path_cls_kwargs, min_order, max_order, order, kwargs):
        validity = _validity_kwargs(**kwargs)
        native = _native_fun(fun, fun_args, fun_kwargs)
        self._solver_of(path_cls)
        common = dict(min_order=min_order, max_order=max_order, order=order, **validity)
        return native, common

    def _emit_grid(self, X, Y, fixed: Point, grid_is_rx: bool, point_cls, fun, fun_args, fun_kwargs, common,
                   filter_objects, path_cls, path_cls_kwargs=None, key=None):
        """Arbitrary Python ``fun`` on a grid: trace all (cell, candidate) on the GPU, call ``fun`` once per
        candidate on the batched paths, accumulate in candidate order (fp32)."""
        candidates = self.all_path_candidates(common["min_order"], common["max_order"], order=common.get("order"),
                                              filter_objects=filter_objects)
        cells = X.size
        if cells * max(len(candidates), 1) > EMIT_LIMIT:
            raise L.D2DUnsupported(-4, f"fun={fun!r} is not fused natively and {cells} cells x {len(candidates)} candidates "
                                       f"exceed the emit limit; use a function from differt2d_amd.utils")
        grid = np.stack([X.reshape(-1), Y.reshape(-1)], axis=-1).astype(F)
        other = np.broadcast_to(fixed.xy, grid.shape)
        txs, rxs = (other, grid) if grid_is_rx else (grid, other)
        ctx = self._ctx()
        self._upload(ctx)
        p = dict(common)
        p.pop("order", None)
        p["min_order"], p["max_order"] = 0, L.D2D_MAX_ORDER
        extra, theta0 = self._solver_setup(path_cls, path_cls_kwargs, candidates, key)
        out = ctx.trace_paths(make_params(**extra, **p), txs, rxs, candidates, theta0=theta0)
        acc = np.zeros(X.shape, F)
        for c, cand in enumerate(candidates):
            k = len(cand)
            path = path_cls(xys=out["xys"][:, c, : k + 2].reshape(*X.shape, k + 2, 2), loss=out["loss"][:, c].reshape(X.shape))
            inter = self.get_interacting_objects(cand)
            moving = point_cls(xy=grid.reshape(*X.shape, 2))
            a, b = (fixed, moving) if grid_is_rx else (moving, fixed)
            val = np.asarray(fun(a, b, path, inter, *fun_args, **(fun_kwargs or {})), dtype=F)
            acc = (acc + out["valid"][:, c].reshape(X.shape) * val).astype(F)
        return acc

    def _emit_grid_grad(self, X, Y, fixed: Point, grid_is_rx: bool, point_cls, fun, fun_args, fun_kwargs, common,
                        filter_objects, path_cls, path_cls_kwargs=None, key=None):
        """Value and per-cell gradient of a sweep with an arbitrary Python ``fun`` (reference scene.py:1892-1923 with any JAX
        callable): the paths of all (cell, candidate) are traced on the GPU, ``fun`` and its derivative w.r.t. the path points
        are evaluated on the host once per candidate (fun_grad.py: ``fun.value_and_grad`` or a tape of ``fun``'s operations),
        and the value+grad kernel chains them through the hand-derived adjoint of the validity and of the path method -- the
        image method, or the reverse pass over the MinPath / FermatPath solver's stored Adam trajectory, started from the same
        initial guesses the paths were traced with (include/d2d.h: d2d_set_path_fun_values, D2D_FUN_CUSTOM).  A function of
        ``path.loss`` is refused by the tape (fun_grad.py)."""
        from .fun_grad import value_and_xys_bar

        candidates = self.all_path_candidates(common["min_order"], common["max_order"], order=common.get("order"),
                                              filter_objects=filter_objects)
        cells = X.size
        if cells * max(len(candidates), 1) > EMIT_LIMIT:
            raise L.D2DUnsupported(-4, f"fun={fun!r} is not fused natively and {cells} cells x {len(candidates)} candidates "
                                       f"exceed the emit limit; use a function from differt2d_amd.utils")
        if not candidates or not cells:
            return np.zeros(X.shape, F), np.zeros(X.shape + (2,), F)
        grid = np.stack([X.reshape(-1), Y.reshape(-1)], axis=-1).astype(F)
        other = np.broadcast_to(fixed.xy, grid.shape)
        txs, rxs = (other, grid) if grid_is_rx else (grid, other)
        ctx = self._ctx()
        self._upload(ctx)
        p = dict(common)
        p.pop("order", None)
        p["min_order"], p["max_order"] = 0, L.D2D_MAX_ORDER
        sextra, theta0 = self._solver_setup(path_cls, path_cls_kwargs, candidates, key)
        out = ctx.trace_paths(make_params(**sextra, **p), txs, rxs, candidates, theta0=theta0)
        f = np.zeros((len(candidates),) + X.shape, F)
        bar = np.zeros((len(candidates),) + X.shape + (L.D2D_MAX_ORDER + 2, 2), F)
        for c, cand in enumerate(candidates):
            k = len(cand)
            val, xb = value_and_xys_bar(fun, fixed.xy, grid.reshape(*X.shape, 2), grid_is_rx,
                                        out["xys"][:, c, : k + 2].reshape(*X.shape, k + 2, 2), out["loss"][:, c].reshape(X.shape),
                                        self.get_interacting_objects(cand), fun_args, fun_kwargs, point_cls, path_cls)
            f[c] = val
            bar[c, ..., : k + 2, :] = xb
        self._upload(ctx, filter_objects)
        ctx.set_grid(X, Y)
        ctx.set_path_fun_values(f, bar)
        if theta0 is not None:
            ctx.set_theta0(theta0)
        params = make_params(fun="custom", grid_role=L.GRID_RX if grid_is_rx else L.GRID_TX, **sextra, **common)
        ctx.launch_vg(params, fixed.xy, scene_vjp=False)
        value, grad = ctx.get_map(), ctx.get_grad_rx()
        ctx.set_path_fun_values(None)
        return value, grad

    def _grid_sweep(self, X, Y, fixed_items, grid_is_rx, point_cls, fun, fun_args, fun_kwargs, reduce_all, grad,
                    value_and_grad, path_cls, path_cls_kwargs, min_order, max_order, order, filter_objects, key, kwargs):
        """Shared driver of the two grid sweeps: ``fixed_items`` are the named end points that stay put (transmitters
        for an RX grid, receivers for a TX grid); one fused launch per fixed point when ``fun`` is native."""
        X = np.ascontiguousarray(X, dtype=F)
        Y = np.ascontiguousarray(Y, dtype=F)
        native, common = self._sweep_params(fun, fun_args, fun_kwargs, path_cls, path_cls_kwargs, min_order, max_order,
                                            order, kwargs)
        want_grad = bool(grad or value_and_grad)

        if native is None:
            if want_grad:
                # (value_and_grad takes precedence over grad, reference scene.py:1920-1923)
                pick = (lambda vg: vg) if value_and_grad else (lambda vg: vg[1])
                gen = ((name, self._emit_grid_grad(X, Y, pt, grid_is_rx, point_cls, fun, fun_args, fun_kwargs, common,
                                                   filter_objects, path_cls, path_cls_kwargs, key)) for name, pt in fixed_items)
                if reduce_all:
                    Z, G = np.zeros(X.shape, F), np.zeros(X.shape + (2,), F)
                    for _, (v, g) in gen:
                        Z, G = (Z + v).astype(F), (G + g).astype(F)
                    return pick((Z, G))
                return ((name, pick(vg)) for name, vg in gen)
            gen = ((name, self._emit_grid(X, Y, pt, grid_is_rx, point_cls, fun, fun_args, fun_kwargs, common,
                                          filter_objects, path_cls, path_cls_kwargs, key)) for name, pt in fixed_items)
            if reduce_all:
                Z = F(0.0)
                for _, p in gen:
                    Z = (Z + p).astype(F)
                return Z
            return gen

        name, extra = native
        ctx = self._ctx()
        cands = None
        if self._solver_of(path_cls) != "image":
            # (grad / value_and_grad: the derivative through the solver's Adam loop, reference optimize.py:83-97)
            cands = self.all_path_candidates(min_order, max_order, order=order, filter_objects=filter_objects)
        sextra, theta0 = self._solver_setup(path_cls, path_cls_kwargs, cands or [], key)
        extra = {**extra, **sextra, "grid_role": L.GRID_RX if grid_is_rx else L.GRID_TX}

        def fetch():
            if value_and_grad:  # takes precedence over grad (reference scene.py:1920-1923)
                return ctx.get_map(), ctx.get_grad_rx()
            if grad:
                return ctx.get_grad_rx()
            return ctx.get_map()

        def launch(pt, out_mode):
            params = make_params(fun=name, out_mode=out_mode, **extra, **common)
            if theta0 is not None:
                ctx.set_theta0(theta0)
            if want_grad:
                ctx.launch_vg(params, pt.xy, scene_vjp=False)
            else:
                ctx.launch(params, pt.xy)

        if reduce_all:
            if not fixed_items:
                return (F(0.0), F(0.0)) if value_and_grad else F(0.0)
            self._upload(ctx, filter_objects)
            ctx.set_grid(X, Y)
            for i, (_, pt) in enumerate(fixed_items):
                launch(pt, L.OUT_ADD if i else L.OUT_OVERWRITE)
            return fetch()

        def results():
            for pt_name, pt in fixed_items:
                self._upload(ctx, filter_objects)
                ctx.set_grid(X, Y)
                launch(pt, L.OUT_OVERWRITE)
                yield pt_name, fetch()

        return results()

    def accumulate_on_receivers_grid_over_paths(
        self, X, Y, fun: PathFun, fun_args: tuple = (), fun_kwargs: Optional[Mapping] = None, *, reduce_all: bool = False,
        grad: bool = False, value_and_grad: bool = False, path_cls: type = ImagePath,
        path_cls_kwargs: Optional[Mapping] = None, receiver_cls: type = Point, min_order: int = 0, max_order: int = 1,
        order: Optional[int] = None, filter_objects: Optional[Callable[[Object], bool]] = None, key=None, **kwargs,
    ):
        """Power-map sweep: for every transmitter, ``Z[i, j] = sum_candidates valid * fun`` with the receiver at
        ``(X[i, j], Y[i, j])`` (reference scene.py:1803-1953). Returns an iterator of ``(tx name, Z)``, or their
        sum if ``reduce_all``; ``grad`` / ``value_and_grad`` add the per-cell gradient w.r.t. the receiver position
        (last axis ``(d/dx, d/dy)``). One fused kernel launch per transmitter when ``fun`` is native."""
        return self._grid_sweep(X, Y, list(self.transmitters.items()), True, receiver_cls, fun, fun_args, fun_kwargs,
                                reduce_all, grad, value_and_grad, path_cls, path_cls_kwargs, min_order, max_order, order,
                                filter_objects, key, kwargs)

    def receivers_grid_value_and_vjp(
        self, X, Y, fun: PathFun, fun_kwargs: Optional[Mapping] = None, *, cotangent=None, path_cls: type = ImagePath,
        path_cls_kwargs: Optional[Mapping] = None, min_order: int = 0, max_order: int = 1, order: Optional[int] = None,
        filter_objects: Optional[Callable[[Object], bool]] = None, key=None, **kwargs,
    ):
        """Reverse-mode sweep w.r.t. the SCENE parameters (what users of the reference obtain by wrapping the sweep
        in ``jax.value_and_grad``, e.g. examples/plot_power_optimize.py:78-93): for every transmitter returns
        ``(name, dict(value=Z, grad_rx=dZ/drx, tx_bar=<cot, dZ/dtx>, objects_bar=<cot, dZ/dxys>[N,2,2],
        phi_bar=<cot, dZ/dphi>[N]))`` where ``cotangent`` (default ones, i.e. the gradient of ``Z.sum()``) has the grid's
        shape.  ``objects_bar[j]`` of a ``Vertex`` carries the gradient w.r.t. its point in row 0; ``phi_bar`` is non-zero
        for ``RIS`` objects only (``path_cls=MinPath`` / ``FermatPath``: the derivative goes through the solver's Adam loop,
        reference optimize.py:83-97; pass ``path_cls_kwargs`` / ``key`` as for the sweep)."""
        X = np.ascontiguousarray(X, dtype=F)
        Y = np.ascontiguousarray(Y, dtype=F)
        native, common = self._sweep_params(fun, (), fun_kwargs, path_cls, path_cls_kwargs, min_order, max_order, order, kwargs)
        if native is None:
            raise L.D2DUnsupported(-4, "the scene VJP needs a natively fused fun (differt2d_amd.utils)")
        name, extra = native
        ctx = self._ctx()
        cands = None
        if self._solver_of(path_cls) != "image":
            cands = self.all_path_candidates(min_order, max_order, order=order, filter_objects=filter_objects)
        sextra, theta0 = self._solver_setup(path_cls, path_cls_kwargs, cands or [], key)
        extra = {**extra, **sextra}
        for tx_name, tx in self.transmitters.items():
            self._upload(ctx, filter_objects)
            ctx.set_grid(X, Y)
            ctx.set_cotangent(cotangent)
            if theta0 is not None:
                ctx.set_theta0(theta0)
            ctx.launch_vg(make_params(fun=name, **extra, **common), tx.xy, scene_vjp=True)
            tx_bar, objects_bar, phi_bar = ctx.get_scene_vjp(with_phi=True)
            yield tx_name, {"value": ctx.get_map(), "grad_rx": ctx.get_grad_rx(), "tx_bar": tx_bar, "objects_bar": objects_bar,
                            "phi_bar": phi_bar}

    def accumulate_on_transmitters_grid_over_paths(
        self, X, Y, fun: PathFun, fun_args: tuple = (), fun_kwargs: Optional[Mapping] = None, *, reduce_all: bool = False,
        grad: bool = False, value_and_grad: bool = False, path_cls: type = ImagePath,
        path_cls_kwargs: Optional[Mapping] = None, transmitter_cls: type = Point, min_order: int = 0, max_order: int = 1,
        order: Optional[int] = None, filter_objects: Optional[Callable[[Object], bool]] = None, key=None, **kwargs,
    ):
        """Transmitter-grid twin (reference scene.py:1489-1648): one map per receiver, the transmitter sits at
        ``(X[i, j], Y[i, j])``; gradients are w.r.t. the transmitter position (scene.py:1617-1620)."""
        return self._grid_sweep(X, Y, list(self.receivers.items()), False, transmitter_cls, fun, fun_args, fun_kwargs,
                                reduce_all, grad, value_and_grad, path_cls, path_cls_kwargs, min_order, max_order, order,
                                filter_objects, key, kwargs)
