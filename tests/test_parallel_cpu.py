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
        from differt2d_amd.parallel import GlooHostComm, sharded_map
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
