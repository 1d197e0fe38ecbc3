"""
The reference's random numbers on the host: JAX's default PRNG -- Threefry-2x32, 20 rounds (Salmon et al., "Parallel random
numbers: as easy as 1, 2, 3", SC'11) -- with the key plumbing of ``jax.random`` (jax/_src/prng.py), restated in NumPy, in
BOTH of JAX's counter layouts:

* **partitionable** (``jax_threefry_partitionable=True``: the default since JAX 0.5.0, hence of the jax 0.5.2 the reference's
  ``uv.lock`` pins) -- the element's 64-bit flat index as the two counter words (high, low), 32 output bits = ``y0 ^ y1``,
  ``split`` = ``stack(y0, y1)`` per index.  The default here.
* **original** (JAX < 0.5, or the flag switched off) -- the counts split in two halves, outputs concatenated.  The reference's
  own doctest of a keyed draw (abc.py:168-174, marked ``+SKIP``) was recorded under this one.

``set_threefry_partitionable(False)`` / ``with threefry_partitionable(False):`` / the environment variable
``D2D_THREEFRY_PARTITIONABLE=0`` select the original layout.

The reference draws random scenes (scene.py:716-733) and the initial guesses of the MinPath / FermatPath solvers
(optimize.py:132, 174-178; one key per candidate, scene.py:1585, 1888; a chain of splits in ``all_paths``, scene.py:1210) from
``jax.random``; with this module a call with ``key=PRNGKey(1234)`` draws the same numbers here.  JAX itself cannot be
imported in this repository's containers, so the restatement is pinned by known answers instead (tests/test_random.py): the
Random123 / JAX test-suite vectors of the block function; original layout: the reference's own doctest of a keyed draw
(abc.py:168-174: Wall.sample(PRNGKey(1234)) = [0.88359046, 1.1781206]), ``random.split(PRNGKey(0))`` from JAX's PRNG design
note, ``random.uniform(PRNGKey(0))`` = 0.41845703, ``split(PRNGKey(42))`` and ``normal(PRNGKey(42))`` = -0.18471177 of JAX's
tutorial before 0.5; partitionable layout: ``split(key(42))`` = [1832780943 270669613], [64467757 2916123636] and
``normal(key(42))`` = -0.028304616 of the same tutorial since.  Host code by nature: draws are a few numbers per call.
"""

from __future__ import annotations

import contextlib
import os

import numpy as np

__all__ = ["PRNGKey", "as_key", "split", "uniform", "random_bits", "threefry2x32", "threefry_2x32",
           "set_threefry_partitionable", "threefry_partitionable"]

_PARTITIONABLE = os.environ.get("D2D_THREEFRY_PARTITIONABLE", "1").strip().lower() not in ("0", "false", "no", "off")


def set_threefry_partitionable(on: bool) -> bool:
    """``jax.config.update("jax_threefry_partitionable", on)``; returns the previous setting."""
    global _PARTITIONABLE
    old, _PARTITIONABLE = _PARTITIONABLE, bool(on)
    return old


@contextlib.contextmanager
def threefry_partitionable(on: bool = True):
    """``with jax.threefry_partitionable(on):``"""
    old = set_threefry_partitionable(on)
    try:
        yield
    finally:
        set_threefry_partitionable(old)

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def PRNGKey(seed: int) -> np.ndarray:
    """``jax.random.PRNGKey(seed)``: the 64-bit seed as two 32-bit words, high word first (``[0, seed]`` for seeds < 2^32)."""
    s = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.array([s >> 32, s & 0xFFFFFFFF], dtype=_U32)


def as_key(key) -> np.ndarray:
    """An int seed, or a raw key of two uint32 words (what ``PRNGKey`` / ``split`` return)."""
    if isinstance(key, (int, np.integer)):
        return PRNGKey(int(key))
    k = np.asarray(key)
    if k.shape != (2,):
        raise TypeError("a key is an int seed or two uint32 words (PRNGKey(seed), split(key)[i])")
    return k.astype(_U32)


def _rotl(x, r):
    return ((x << _U32(r)) | (x >> _U32(32 - r))).astype(_U32)


def threefry2x32(key, x0, x1):
    """The block function: 20 rounds of Threefry-2x32 on the counter words (x0[i], x1[i]) under ``key``."""
    with np.errstate(over="ignore"):
        k0, k1 = _U32(key[0]), _U32(key[1])
        ks = (k0, k1, _U32(k0 ^ k1 ^ _U32(0x1BD11BDA)))
        x0 = (np.asarray(x0, dtype=_U32) + ks[0]).astype(_U32)
        x1 = (np.asarray(x1, dtype=_U32) + ks[1]).astype(_U32)
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = (x0 + x1).astype(_U32)
                x1 = _rotl(x1, r) ^ x0
            x0 = (x0 + ks[(i + 1) % 3]).astype(_U32)
            x1 = (x1 + ks[(i + 2) % 3] + _U32(i + 1)).astype(_U32)
        return x0, x1


def threefry_2x32(key, count) -> np.ndarray:
    """``jax._src.prng.threefry_2x32``: the counts (any shape) are split in two halves -- an odd count padded with one zero --
    that become the two counter words; the output halves are concatenated back."""
    count = np.asarray(count, dtype=_U32)
    flat = count.ravel()
    odd = flat.size % 2
    if odd:
        flat = np.concatenate([flat, np.zeros(1, _U32)])
    h = flat.size // 2
    y0, y1 = threefry2x32(as_key(key), flat[:h], flat[h:])
    out = np.concatenate([y0, y1])
    return (out[:-1] if odd else out).reshape(count.shape)


def _iota_2x32(n: int):
    """``iota_2x32_shape``: the 64-bit flat index of each of ``n`` elements as (high, low) 32-bit words."""
    i = np.arange(n, dtype=np.uint64)
    return (i >> np.uint64(32)).astype(_U32), (i & np.uint64(0xFFFFFFFF)).astype(_U32)


def split(key, num: int = 2) -> np.ndarray:
    """``jax.random.split``: ``num`` new keys, shape (num, 2) (``_threefry_split_foldlike`` / ``_threefry_split_original``)."""
    num = int(num)
    if _PARTITIONABLE:
        hi, lo = _iota_2x32(num)
        y0, y1 = threefry2x32(as_key(key), hi, lo)
        return np.stack([y0, y1], axis=-1)
    return threefry_2x32(key, np.arange(2 * num, dtype=_U32)).reshape(num, 2)


def random_bits(key, shape) -> np.ndarray:
    """32 random bits per element (``_threefry_random_bits_partitionable``: counter = the element's 64-bit flat index, bits =
    ``y0 ^ y1``; ``_threefry_random_bits_original``: the flat indices split in two halves)."""
    shape = tuple(int(s) for s in (shape if np.ndim(shape) else (shape,)))
    n = int(np.prod(shape)) if shape else 1
    if n >= 2**32:
        raise ValueError("more than 2^32 - 1 elements per draw are not supported")
    if _PARTITIONABLE:
        hi, lo = _iota_2x32(n)
        y0, y1 = threefry2x32(as_key(key), hi, lo)
        return (y0 ^ y1).reshape(shape)
    return threefry_2x32(key, np.arange(n, dtype=_U32)).reshape(shape)


def uniform(key, shape=(), minval: float = 0.0, maxval: float = 1.0) -> np.ndarray:
    """``jax.random.uniform(key, shape, float32, minval, maxval)``: 23 mantissa bits under the exponent of 1.0, minus 1."""
    bits = random_bits(key, shape)
    f = ((bits >> _U32(9)) | _U32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, f * (hi - lo) + lo).astype(np.float32)
