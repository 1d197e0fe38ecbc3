"""RCCL plumbing on one GPU (world_size 1): librccl is dlopen'ed, a communicator is created, the all-gather
and all-reduce run on the context's stream and leave the data unchanged.  (Ranks > 1 need one GPU each; the
partition logic for N > 1 is covered on CPU over gloo in test_parallel_cpu.py.)"""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import random_scene, unit_grid

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world_size_one_roundtrip_in_fresh_process():
    """torch (imported by other tests for the autodiff oracle) bundles its own libamdhip64 / librccl under the
    same SONAMEs as the system ROCm; the RCCL data plane therefore runs in torch-free processes, like bench.py's."""
    code = "import sys; sys.path[:0] = [%r, %r]; import test_gpu_comm as t; t._roundtrip(); print('RCCL-OK')" % (
        ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def _roundtrip():
    from differt2d_amd.engine import Context, make_params
    from differt2d_amd.parallel import ShardedSweep

    tx, walls = random_scene(9, seed=5)
    X, Y = unit_grid(40, 27)
    with Context(0) as ctx:
        uid = Context.comm_unique_id()
        assert len(uid) == 128
        sweep = ShardedSweep(ctx, 0, 1, unique_id=uid)
        sweep.setup(walls, X, Y)
        p = make_params(max_order=2, approx=True)
        sweep.step(p, tx)
        direct = ctx.get_map()
        ctx.comm_allgather_map()
        gathered = ctx.comm_get_gathered(1)
        assert gathered.shape == (1, *X.shape) and np.array_equal(gathered[0], direct)
        assert np.array_equal(sweep.result(), direct)
        ctx.launch_vg(p, tx, scene_vjp=True)
        before = ctx.get_scene_vjp()
        ctx.comm_allreduce_vjp()
        after = ctx.get_scene_vjp()
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
        ctx.comm_allgather_map(grad=True)
        g = ctx.comm_get_gathered(1, grad=True)
        assert np.array_equal(g[0], ctx.get_grad_rx())
        # the all-gather of step k runs on its own stream while step k+1 sweeps: the gathered map must be step k's
        # even though the next sweep has already overwritten the context's value map
        txs = [np.array([0.3 + 0.05 * k, 0.4], np.float32) for k in range(4)]
        maps = []
        for t in txs:
            ctx.launch(p, t)
            maps.append(ctx.get_map())
        assert not np.array_equal(maps[0], maps[1])
        for k, t in enumerate(txs):
            ctx.launch(p, t)
            ctx.comm_allgather_map()              # sweep + all-gather, no host synchronisation
            if k + 1 < len(txs):
                ctx.launch(p, txs[k + 1])         # overwrites d_out while the gather may still be in flight
                assert np.array_equal(ctx.comm_get_gathered(1)[0], maps[k])
                assert np.array_equal(ctx.get_map(), maps[k + 1])
        for t in txs:                             # back-to-back steps, one synchronisation at the end
            ctx.launch(p, t)
            ctx.comm_allgather_map()
        ctx.synchronize()
        assert np.array_equal(ctx.comm_get_gathered(1)[0], maps[-1])
        # the value and the gradient map are gathered into separate buffers: the gradient gathered earlier is still there
        assert np.array_equal(ctx.comm_get_gathered(1, grad=True)[0], g[0])
        with pytest.raises(Exception):            # a buffer of the wrong size is refused, not overrun
            ctx._lib.d2d_comm_get_gathered  # noqa: B018
            bad = np.empty(7, np.float32)
            from differt2d_amd import _lib as L
            L.check(ctx._lib.d2d_comm_get_gathered(ctx._ctx, 0, bad, bad.size))
        # one step through ShardedSweep: value + gradient gathered to the root, scene VJP all-reduced, all asynchronous
        sweep.step(p, tx, grad=True, scene_vjp=True, gather="root", root=0)
        Z, G = sweep.result(), sweep.grad_result()
        tb, wb = sweep.scene_vjp()
        ref = ctx.value_and_grads(tx, X, Y, max_order=2, approx=True)
        assert np.array_equal(Z, ref["value"]) and np.array_equal(G, ref["grad_rx"], equal_nan=True)
        assert np.array_equal(tb, ref["tx_bar"]) and np.array_equal(wb, ref["walls_bar"])
        ctx.comm_gather_map(root=0)               # the ncclSend / ncclRecv path itself, at world size 1: a local copy
        ctx.comm_gather_map(root=0, grad=True)
        assert np.array_equal(ctx.comm_get_gathered(1)[0], ref["value"])
        assert np.array_equal(ctx.comm_get_gathered(1, grad=True)[0], ref["grad_rx"], equal_nan=True)
        with pytest.raises(Exception):            # gather to a rank that does not exist
            ctx.comm_gather_map(root=1)
        # a new grid invalidates what was gathered for the old one (and the gradient map itself)
        ctx.set_grid(X[:8], Y[:8])
        with pytest.raises(Exception):
            ctx.comm_get_gathered(1)
        with pytest.raises(Exception):
            ctx.comm_allgather_map(grad=True)
        with pytest.raises(Exception):
            ctx.get_grad_rx()
        assert ctx.comm_allreduce_host([3.0], "max")[0] == 3.0
        ctx.comm_destroy()
