"""
Multi-GPU sweep: one process per GPU, RX rows sharded over ranks, maps assembled with ONE all-gather (or one gather to
a root rank) per map and the scene VJP with one all-reduce.

The reference has no multi-device code (its only batching is ``jax.vmap`` over the grid,
scene.py:1927-1932).  Every RX cell is independent, so the partition needs no data-path exchange other
than assembling the final map:

* rows are dealt to ranks in blocks of ``BLOCK_ROWS`` (= the kernel's 8-row wave tile) round-robin,
  ``block b -> rank b % world``: neighbouring blocks have similar pruning rates, so interleaving
  balances the load, and a block keeps the 8x8 tile coherence the kernel's wave-level skips rely on;
* every rank pads its shard to the same number of rows (last row repeated) so that a plain
  ``ncclAllGather`` applies; padded rows are dropped on assembly;
* the scene (a few kB) and all parameters are replicated; candidates are enumerated on device.

Data plane AND barrier / max-over-ranks: RCCL directly on the library's device buffers (``d2d_comm_*``, xGMI on
an MI355X node).  Rendezvous (shipping the 128-byte unique id): a file in /tmp (:func:`file_rendezvous`), so the
GPU processes import no deep-learning framework -- PyTorch bundles its own libamdhip64 / librccl under the system ROCm's SONAMEs
and must stay out of a process that drives the system RCCL.  (The CPU tests of the partition logic move host arrays through a
gloo process group instead: ``tests/gloo_comm.py``, test infrastructure, not part of this package.)
"""

from __future__ import annotations

from typing import Callable, Optional

import numpy as np

BLOCK_ROWS = 8


class RowShards:
    """Index logic of the row-block round-robin partition (pure NumPy, no device)."""

    def __init__(self, m: int, world: int, block: int = BLOCK_ROWS):
        if m <= 0 or world <= 0 or block <= 0:
            raise ValueError("m, world and block must be positive")
        self.m, self.world, self.block = int(m), int(world), int(block)
        self.n_blocks = -(-self.m // self.block)
        self._rows = [self._rows_of(r) for r in range(self.world)]
        self.pad_rows = max(1, max(len(r) for r in self._rows))

    def _rows_of(self, rank: int) -> np.ndarray:
        rows = [np.arange(b * self.block, min((b + 1) * self.block, self.m)) for b in range(rank, self.n_blocks, self.world)]
        return np.concatenate(rows) if rows else np.zeros(0, np.int64)

    def rows(self, rank: int) -> np.ndarray:
        """Global row indices owned by ``rank`` (ascending)."""
        return self._rows[rank]

    def take(self, A: np.ndarray, rank: int) -> np.ndarray:
        """Rows of ``A`` owned by ``rank``, padded to ``pad_rows`` rows by repeating the last owned row
        (rank without rows: row 0), so that every rank sweeps a grid of the same shape."""
        rows = self._rows[rank]
        idx = np.empty(self.pad_rows, np.int64)
        idx[: len(rows)] = rows
        idx[len(rows):] = rows[-1] if len(rows) else 0
        return np.ascontiguousarray(A[idx])

    def cotangent_mask(self, rank: int, n_cols: int, cotangent: Optional[np.ndarray] = None) -> np.ndarray:
        """Cotangent of ``rank``'s padded shard: the rows of ``cotangent`` (``[m, n_cols]``, default ones) it owns and ZERO
        on the padding rows.  The padding repeats a real row, so without this mask a scene VJP summed over ranks
        (``d2d_comm_allreduce_vjp``) would count that row once per repetition."""
        rows = self._rows[rank]
        out = np.zeros((self.pad_rows, int(n_cols)), np.float32)
        if len(rows):
            out[: len(rows)] = 1.0 if cotangent is None else np.asarray(cotangent, np.float32)[rows]
        return out

    def padded(self, rank: int) -> bool:
        """Does ``rank``'s shard carry padding rows?"""
        return len(self._rows[rank]) < self.pad_rows

    def assemble(self, gathered: np.ndarray) -> np.ndarray:
        """``gathered[world, pad_rows, ...]`` -> ``[m, ...]`` (drops padding, undoes the interleave)."""
        if gathered.shape[0] != self.world or gathered.shape[1] != self.pad_rows:
            raise ValueError(f"expected leading shape ({self.world}, {self.pad_rows}), got {gathered.shape[:2]}")
        out = np.empty((self.m, *gathered.shape[2:]), gathered.dtype)
        for r in range(self.world):
            rows = self._rows[r]
            out[rows] = gathered[r, : len(rows)]
        return out


def file_rendezvous(rank: int, world: int, make_id: Callable[[], bytes], timeout: float = 300.0) -> bytes:
    """Ships rank 0's 128-byte RCCL unique id to the other ranks of ONE node through a directory in /tmp, so that
    the GPU processes need no PyTorch / MPI at run time (PyTorch's distributed launcher, if used at all, only starts the ranks).

    The directory is ``$D2D_RDZV_DIR`` or ``/tmp/d2d_rdzv_<MASTER_PORT>_<parent pid>`` (all workers of one
    launcher agent share the parent pid).  Rank 0 writes ``id.bin`` atomically; it removes the directory in
    :func:`file_rendezvous_cleanup` once every rank has initialised its communicator."""
    import os
    import time

    d = rendezvous_dir()
    path = os.path.join(d, "id.bin")
    if rank == 0:
        os.makedirs(d, exist_ok=True)
        uid = make_id()
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == 128:
                return uid
        except OSError:
            pass
        if time.time() - t0 > timeout:
            raise TimeoutError(f"rank {rank}: no unique id at {path} after {timeout} s")
        time.sleep(0.02)


def rendezvous_dir() -> str:
    import os

    return os.environ.get("D2D_RDZV_DIR") or f"/tmp/d2d_rdzv_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"


def file_rendezvous_cleanup(rank: int):
    import shutil

    if rank == 0:
        shutil.rmtree(rendezvous_dir(), ignore_errors=True)


class FileHostComm:
    """Control plane of last resort for ONE node: barrier and all-reduce of a few host doubles through small files in
    the rendezvous directory (busy-polled).  bench.py falls back to it when the RCCL communicator cannot be created,
    so that the row-sharded sweep -- which needs no data-path collective -- can still be timed over all ranks."""

    def __init__(self, rank: int, world: int, directory: Optional[str] = None):
        import os

        self.rank, self.world = int(rank), int(world)
        self.dir = directory or rendezvous_dir() + "_ctl"
        os.makedirs(self.dir, exist_ok=True)
        self.seq = 0

    def allreduce(self, values, op: str = "sum") -> np.ndarray:
        import os
        import time

        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        self.seq += 1
        mine = os.path.join(self.dir, f"r{self.seq}_{self.rank}.npy")
        with open(mine + ".tmp", "wb") as f:
            np.save(f, v)
        os.replace(mine + ".tmp", mine)
        parts = []
        t0 = time.time()
        for r in range(self.world):
            path = os.path.join(self.dir, f"r{self.seq}_{r}.npy")
            while True:
                try:
                    with open(path, "rb") as f:
                        parts.append(np.load(f))
                    break
                except (OSError, ValueError, EOFError):
                    if time.time() - t0 > 600.0:
                        raise TimeoutError(f"rank {self.rank}: rank {r} never reached step {self.seq}")
                    time.sleep(0)
        if self.seq > 2:  # everybody has read round seq-2 by now (it took part in round seq-1 after reading it)
            try:
                os.remove(os.path.join(self.dir, f"r{self.seq - 2}_{self.rank}.npy"))
            except OSError:
                pass
        stack = np.stack(parts)
        return stack.max(0) if op == "max" else stack.sum(0)

    def barrier(self):
        self.allreduce([1.0], "sum")


def sharded_map(X: np.ndarray, Y: np.ndarray, compute_shard: Callable[[np.ndarray, np.ndarray], np.ndarray], comm,
                block: int = BLOCK_ROWS) -> np.ndarray:
    """Generic driver: every rank computes its row shard with ``compute_shard(X_local, Y_local)`` and
    the full map is assembled on every rank with one all-gather."""
    shards = RowShards(X.shape[0], comm.world, block)
    local = compute_shard(shards.take(X, comm.rank), shards.take(Y, comm.rank))
    return shards.assemble(comm.allgather(local))


class ShardedSweep:
    """Resident multi-GPU power-map sweep on top of one :class:`~differt2d_amd.engine.Context` per rank.

    ``setup`` uploads this rank's row shard (and its cotangent, zero on padding rows) once; ``step`` = fused kernel +
    collectives, all asynchronous: the value map (and, with ``grad``, the per-cell gradient map) is gathered either to
    every rank (``gather="all"``, one ``ncclAllGather`` per map) or to ``root`` only (``gather="root"``,
    ``ncclSend``/``ncclRecv`` -- what the reference's single-process result needs, scene.py:1927-1953), and the scene
    VJP is summed over ranks with one ``ncclAllReduce`` in the same step; ``result`` / ``grad_result`` download and
    assemble the full maps on the ranks that hold them (``None`` elsewhere), ``scene_vjp`` returns the reduced VJP.

    ``ctx`` is duck-typed (the CPU tests drive this class with an oracle-backed stand-in over gloo)."""

    def __init__(self, ctx, rank: int, world: int, unique_id: Optional[bytes] = None):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        if unique_id is not None:
            ctx.comm_init(unique_id, rank, world)
        self.shards = None
        self._gathered = None  # (mode, root, grad) of the last step

    def setup(self, walls, X, Y, kind=None, phi=None, cotangent=None):
        self.shards = RowShards(X.shape[0], self.world)
        self.ctx.set_scene(walls, kind, phi)
        self.ctx.set_grid(self.shards.take(X, self.rank), self.shards.take(Y, self.rank))
        if self.world > 1 or cotangent is not None:
            # padding rows repeat a real row: their cotangent is zero, or the all-reduced VJP would count that row twice
            self.ctx.set_cotangent(self.shards.cotangent_mask(self.rank, X.shape[1], cotangent))

    def step(self, params, tx, grad: bool = False, scene_vjp: bool = False, gather: Optional[str] = "all", root: int = 0):
        if gather not in (None, "all", "root"):
            raise ValueError("gather must be None, 'all' or 'root'")
        if grad or scene_vjp:
            self.ctx.launch_vg(params, tx, scene_vjp=scene_vjp)
        else:
            self.ctx.launch(params, tx)
        self._gathered = None
        if self.world > 1:
            for g in ([False, True] if grad else [False]):
                if gather == "all":
                    self.ctx.comm_allgather_map(grad=g)
                elif gather == "root":
                    self.ctx.comm_gather_map(root=root, grad=g)
            if scene_vjp:
                self.ctx.comm_allreduce_vjp()
            if gather:
                self._gathered = (gather, int(root), bool(grad))

    def _holds_result(self) -> bool:
        if self.world == 1:
            return True
        if self._gathered is None:
            raise RuntimeError("the last step gathered nothing: call step(..., gather='all' or 'root')")
        mode, root, _ = self._gathered
        return mode == "all" or root == self.rank

    def result(self) -> Optional[np.ndarray]:
        """The assembled value map ``[m, n]`` on the ranks that hold it, ``None`` on the others."""
        if self.world == 1:
            return self.shards.assemble(self.ctx.get_map()[None])
        if not self._holds_result():
            return None
        return self.shards.assemble(self.ctx.comm_get_gathered(self.world))

    def grad_result(self) -> Optional[np.ndarray]:
        """The assembled per-cell gradient map ``[m, n, 2]`` (a step with ``grad=True``)."""
        if self.world == 1:
            return self.shards.assemble(self.ctx.get_grad_rx()[None])
        if not self._holds_result():
            return None
        if not self._gathered[2]:
            raise RuntimeError("the last step did not gather the gradient map: call step(..., grad=True)")
        return self.shards.assemble(self.ctx.comm_get_gathered(self.world, grad=True))

    def scene_vjp(self):
        """(tx_bar, xys_bar) summed over all ranks' cells (a step with ``scene_vjp=True``)."""
        return self.ctx.get_scene_vjp()
