"""N > 1 path on CPU: the row-block partition and the all-gather assembly, world_size 2 and 3 over gloo
(the per-shard compute is the oracle here; on GPUs it is the fused kernel + RCCL)."""

import os
import socket
import sys

import numpy as np
import pytest

from differt2d_amd.parallel import BLOCK_ROWS, RowShards

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("m,world", [(1, 1), (7, 2), (8, 2), (64, 8), (100, 3), (1024, 8), (13, 5), (2048, 8)])
def test_row_shards_partition(m, world):
    sh = RowShards(m, world)
    owned = np.concatenate([sh.rows(r) for r in range(world)])
    assert sorted(owned.tolist()) == list(range(m))  # a partition: every row exactly once
    assert max(len(sh.rows(r)) for r in range(world)) - min(len(sh.rows(r)) for r in range(world)) <= BLOCK_ROWS
    A = np.arange(m * 3, dtype=np.float32).reshape(m, 3)
    gathered = np.stack([sh.take(A, r) for r in range(world)])
    assert gathered.shape == (world, sh.pad_rows, 3)
    assert np.array_equal(sh.assemble(gathered), A)
    for r in range(world):  # whole 8-row blocks, dealt round-robin
        rows = sh.rows(r)
        assert all((row // BLOCK_ROWS) % world == r for row in rows)


def test_row_shards_rejects_bad_shapes():
    with pytest.raises(ValueError):
        RowShards(0, 2)
    sh = RowShards(10, 2)
    with pytest.raises(ValueError):
        sh.assemble(np.zeros((3, sh.pad_rows, 4), np.float32))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist

        from conftest import random_scene, unit_grid
        from differt2d_amd.parallel import sharded_map
        from gloo_comm import GlooHostComm
        from oracle import c_oracle as CO

        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = GlooHostComm()
        tx, walls = random_scene(9, seed=5)
        X, Y = unit_grid(21, 37)  # 37 rows: ragged last block, uneven shards
        kw = dict(min_order=0, max_order=2, approx=True)
        full = sharded_map(X, Y, lambda xs, ys: CO.power_map(walls, tx, xs, ys, nthreads=1, **kw), comm)
        want = CO.power_map(walls, tx, X, Y, nthreads=1, **kw)
        vjp = comm.allreduce_sum(np.array([rank + 1.0, 2.0]))
        comm.barrier()
        dist.destroy_process_group()
        q.put((rank, bool(np.array_equal(full, want)), vjp.tolist()))
    except Exception as e:  # pragma: no cover
        q.put((rank, False, repr(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_map_over_gloo(world):
    import multiprocessing as mp

    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, vjp in sorted(results):
        assert ok is True, f"rank {rank}: {vjp}"
        assert vjp == [world * (world + 1) / 2, 2.0 * world]


def _rdzv_worker(rank, world, d, q):
    sys.path.insert(0, ROOT)
    os.environ["D2D_RDZV_DIR"] = d
    from differt2d_amd.parallel import file_rendezvous

    uid = file_rendezvous(rank, world, lambda: bytes(range(128)), timeout=60)
    q.put((rank, uid))


def test_file_rendezvous_ships_the_unique_id(tmp_path):
    import multiprocessing as mp

    from differt2d_amd.parallel import file_rendezvous_cleanup

    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    d = str(tmp_path / "rdzv")
    procs = [mpctx.Process(target=_rdzv_worker, args=(r, 3, d, q)) for r in (2, 1, 0)]  # rank 0 starts last
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=30)
    assert all(got[r] == bytes(range(128)) for r in range(3))
    os.environ["D2D_RDZV_DIR"] = d
    try:
        file_rendezvous_cleanup(0)
    finally:
        del os.environ["D2D_RDZV_DIR"]
    assert not os.path.exists(d)


def _filecomm_worker(rank, world, d, q):
    sys.path.insert(0, ROOT)
    from differt2d_amd.parallel import FileHostComm

    comm = FileHostComm(rank, world, directory=d)
    out = []
    for it in range(6):  # several rounds: old round files are removed as it goes
        comm.barrier()
        out.append(float(comm.allreduce([rank + 10.0 * it], "max")[0]))
        out.append(float(comm.allreduce([rank + 1.0], "sum")[0]))
    q.put((rank, out, sorted(os.listdir(d))))


def test_file_host_comm_barrier_and_allreduce(tmp_path):
    """bench.py's control plane of last resort (no RCCL communicator): world_size 3 over files."""
    import multiprocessing as mp

    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    d = str(tmp_path / "ctl")
    world = 3
    procs = [mpctx.Process(target=_filecomm_worker, args=(r, world, d, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    want = []
    for it in range(6):
        want += [world - 1 + 10.0 * it, world * (world + 1) / 2]
    for rank, out, _ in res:
        assert out == want, (rank, out)
    assert max(len(files) for _, _, files in res) <= 3 * world  # at most the last few rounds are left behind


class _OracleContext:
    """Stand-in for engine.Context on CPU: the oracle computes this rank's shard, gloo moves the maps.  It implements
    exactly the methods ShardedSweep drives, with the library's semantics (separate value / gradient gather buffers,
    only the root of a gather holds the result, the VJP all-reduce sums what the local sweep produced)."""

    def __init__(self, comm):
        self.comm = comm
        self.cot = None
        self.gathered = {False: None, True: None}
        self.calls = []

    def set_scene(self, walls, kind=None, phi=None):
        self.walls = np.asarray(walls, np.float32)

    def set_grid(self, X, Y):
        self.X, self.Y = X, Y
        self.cot = None

    def set_cotangent(self, cot=None):
        self.cot = None if cot is None else np.asarray(cot, np.float32)

    def _kw(self, p):
        return dict(min_order=p.min_order, max_order=p.max_order, approx=bool(p.approx))

    def launch(self, params, tx):
        from oracle import c_oracle as CO

        self.calls.append("launch")
        self.value = CO.power_map(self.walls, tx, self.X, self.Y, nthreads=1, **self._kw(params))

    def launch_vg(self, params, tx, scene_vjp=False):
        from oracle import ref as R

        self.calls.append("launch_vg")
        g = R.power_map_value_and_grads(self.walls, tx, self.X, self.Y, cotangent=self.cot, dtype="float64", **self._kw(params))
        self.value, self.grad = g["value"].astype(np.float32), g["grad_rx"].astype(np.float32)
        self.vjp = (g["tx_bar"], g["walls_bar"]) if scene_vjp else None

    def comm_allgather_map(self, grad=False):
        self.calls.append(f"allgather{int(grad)}")
        self.gathered[grad] = self.comm.allgather(self.grad if grad else self.value)

    def comm_gather_map(self, root=0, grad=False):
        self.calls.append(f"gather{int(grad)}->{root}")
        self.gathered[grad] = self.comm.gather(self.grad if grad else self.value, root)

    def comm_get_gathered(self, world, grad=False):
        if self.gathered[grad] is None:
            raise RuntimeError("D2D_ERR_STATE: nothing gathered on this rank")
        return self.gathered[grad]

    def comm_allreduce_vjp(self):
        self.calls.append("allreduce_vjp")
        flat = np.concatenate([self.vjp[0].reshape(-1), self.vjp[1].reshape(-1)])
        red = self.comm.allreduce_sum(flat)
        self.vjp = (red[:2], red[2:].reshape(self.vjp[1].shape))

    def get_map(self):
        return self.value

    def get_grad_rx(self):
        return self.grad

    def get_scene_vjp(self):
        return self.vjp


def _sweep_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch
        import torch.distributed as dist

        torch.set_num_threads(1)
        from conftest import random_scene, unit_grid
        from differt2d_amd.engine import make_params
        from differt2d_amd.parallel import ShardedSweep
        from gloo_comm import GlooHostComm
        from oracle import ref as R

        dist.init_process_group("gloo", rank=rank, world_size=world)
        ctx = _OracleContext(GlooHostComm())
        tx, walls = random_scene(5, seed=5)
        X, Y = unit_grid(9, 21)  # 21 rows over 2 ranks: 16 + 5 -> rank 1's shard is padded with 11 repeated rows
        rng = np.random.default_rng(3)
        cot = (rng.random(X.shape) + 0.5).astype(np.float32)
        p = make_params(min_order=0, max_order=2, approx=True)
        want = R.power_map_value_and_grads(walls, tx, X, Y, cotangent=cot, dtype="float64", min_order=0, max_order=2, approx=True)
        sw = ShardedSweep(ctx, rank, world)
        sw.setup(walls, X, Y, cotangent=cot)
        problems = []
        # 1. forward map, all-gather: every rank assembles the full map
        sw.step(p, tx)
        from oracle import c_oracle as CO

        if not np.array_equal(sw.result(), CO.power_map(walls, tx, X, Y, nthreads=1, min_order=0, max_order=2, approx=True)):
            problems.append("all-gathered value map")
        # 2. value + grad + scene VJP, gathered to rank 1 only, VJP all-reduced in the same step
        sw.step(p, tx, grad=True, scene_vjp=True, gather="root", root=1)
        Z, G = sw.result(), sw.grad_result()
        if rank == 1:
            if not np.array_equal(Z, want["value"].astype(np.float32)):
                problems.append("root-gathered value map")
            if not np.allclose(G, want["grad_rx"], rtol=1e-6, atol=1e-6):
                problems.append("root-gathered gradient map")
        elif Z is not None or G is not None:
            problems.append("a non-root rank claims to hold the gathered maps")
        tb, wb = sw.scene_vjp()
        # padded rows carry a zero cotangent: the reduced VJP equals the single-process VJP (no row counted twice)
        if not (np.allclose(tb, want["tx_bar"], rtol=1e-9, atol=1e-9) and np.allclose(wb, want["walls_bar"], rtol=1e-9, atol=1e-9)):
            problems.append(f"all-reduced scene VJP: {tb} vs {want['tx_bar']}")
        order = [c for c in ctx.calls if c != "launch"]
        if order != ["allgather0", "launch_vg", "gather0->1", "gather1->1", "allreduce_vjp"]:
            problems.append(f"call order {order}")
        # 3. a step that gathers nothing must not hand out a stale map
        sw.step(p, tx, gather=None)
        try:
            sw.result()
            problems.append("result() after gather=None did not raise")
        except RuntimeError:
            pass
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, problems))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, [repr(e) + traceback.format_exc()]))


def test_sharded_sweep_end_to_end_over_gloo():
    """ShardedSweep (setup -> step -> gather / all-reduce -> assemble) at world_size 2 with an oracle-backed context:
    value map by all-gather, value + gradient maps gathered to a root, scene VJP all-reduced with padded rows masked."""
    import multiprocessing as mp

    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_sweep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, problems in sorted(results):
        assert problems == [], f"rank {rank}: {problems}"


def test_cotangent_mask_zeroes_padding_rows():
    sh = RowShards(21, 2)
    assert sh.padded(1) and not sh.padded(0)
    cot = np.arange(21 * 3, dtype=np.float32).reshape(21, 3)
    m1 = sh.cotangent_mask(1, 3, cot)
    assert m1.shape == (sh.pad_rows, 3)
    assert np.array_equal(m1[: len(sh.rows(1))], cot[sh.rows(1)]) and not m1[len(sh.rows(1)):].any()
    total = sum(sh.cotangent_mask(r, 3).sum() for r in range(2))
    assert total == 21 * 3  # every real cell exactly once


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without a launcher: the parent starts N children of itself with RANK /
    WORLD_SIZE / LOCAL_RANK set, never touches HIP, relays rank 0's output and exits with the worst child's code.  On this
    CPU box both children must get as far as creating their context (D2D_ERR_NO_DEVICE: there is no CPU fallback) -- not
    stop at a launcher check."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras"],
                         capture_output=True, text=True, timeout=300, env=env)
    from differt2d_amd import _lib as L

    if L.device_count() >= 2:  # a multi-GPU box: a full run, one JSON line from rank 0
        import json

        assert out.returncode == 0, out.stderr[-3000:]
        line = json.loads(out.stdout.strip().splitlines()[-1])
        assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and len(line["kernel_ms_per_rank"]) == 2
        return
    if L.device_count() == 1:  # one GPU: both ranks create their context on it, RCCL refuses two ranks on one device
        assert out.returncode == 3, (out.returncode, out.stderr[-3000:])
        for r in (0, 1):
            assert f"[bench rank {r}] RCCL communicator unavailable" in out.stderr, out.stderr[-3000:]
        return
    assert out.returncode == 4, (out.returncode, out.stderr[-3000:])
    for r in (0, 1):
        assert f"[bench rank {r}] D2D_ERR_NO_DEVICE" in out.stderr, out.stderr[-3000:]
    assert "must be launched with" not in out.stderr and out.stdout.strip() == ""
    # under torch.distributed.run's environment the script is a rank itself and does not spawn anything
    env.update(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-extras"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 4 and out.stderr.count("D2D_ERR_NO_DEVICE") == 1
