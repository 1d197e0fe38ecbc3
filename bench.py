#!/usr/bin/env python3
"""
bench.py -- ray-path candidates/s of the fused power-map sweep (BASELINE.json metric) on N MI355X.

Step      = one forward power map of the workload, inputs resident in HBM:
            --workload cfg2 (default): 50 random walls (NumPy seed 1234), 1 TX, 1024 x 1024 RX grid over the unit square,
            orders 0..2 (C = 2501 candidates per cell) = BASELINE.json configs[1]; hard (reference default) validity.
            --workload cfg4: 200 walls, 2048 x 2048 grid, orders 0..3 (C = 7 960 201) = configs[3].
N > 1     = one process per GPU: either launched by torch.distributed.run (RANK / WORLD_SIZE / LOCAL_RANK in the
            environment), or -- plain `python bench.py --gpus N` -- by this script itself: the parent, which never
            touches HIP, starts N child processes of itself with those variables set, relays rank 0's JSON line and exits
            with the worst child's code.  torch is never imported: rendezvous through a file in /tmp, data plane, barrier
            and max-over-ranks through RCCL.  Rows are dealt to ranks in 8-row blocks
            round-robin (differt2d_amd/parallel.py).  cfg2: WEAK scaling, the grid becomes (1024 N) x 1024 cells over the
            same unit square, 1024 x 1024 per rank.  cfg4: STRONG scaling, the 2048 x 2048 grid is split over the ranks.
            The default cfg2 line also carries, at every N, configs[3] split over the ranks (`strong_cfg4`) and, at N > 1,
            configs[1]'s own 1024 x 1024 grid split over the ranks (`strong_cfg2`) and what the gather adds to a step
            (`gather_cost`: the same steps with and without it).
            Each step ends with ONE RCCL gather of the value map to rank 0 (--gather root, ncclSend/ncclRecv; or --gather
            all, ncclAllGather), on a second stream so that it overlaps the next step's sweep.  If the communicator cannot
            be created the run FAILS (exit code 3): a number without the gather is not the configured workload.
value     = all cells of all ranks x C / wall time of the K timed steps (max over ranks).
roofline  = the kernel is FP32-VALU bound (~1e6 FLOP per HBM byte, no MFMA-shaped work).  Three
            labelled fractions of the 157.3 TFLOP/s FP32 vector peak, all over the sweep kernel's own duration (HIP
            events around it, on its stream):
              executed_with_culling   every unit the kernel executed (counters of its instrumented build, include/d2d.h)
                                      priced with SURVEY.md's per-unit FLOP figures, the conservative culling itself at
                                      120 FLOP per lane and level (this repository's own pricing, not a SURVEY figure)
              reference_work_only     the same without the culling: only work the reference's algorithm also does
              unpruned_equivalent     SURVEY.md's unpruned count / time: a statement of pruning, not of utilisation
            `frac` / `achieved` are executed_with_culling (as in round 1).  `traffic`: HBM bytes per launch from the newest
            committed PMC summary (profiles/rNN_aM_pmc.json; rocprofv3 PMC passes, scripts/profile_gpu.sh), with its source in
            `traffic_source` -- not measured inside this run.
extras    = (N = 1, outside the timed region, skipped by --no-extras) the same map in hard_sigmoid and sigmoid validity, a
            moving-TX sequence (a different transmitter every step: the schedule's work history is then always one step
            stale), the first launch after set_grid, value+grad (default = culling + NaN scan, the exhaustive cross-check,
            and without the scan), `more_modes` (value+grad in hard_sigmoid / sigmoid validity, TX grids forward and
            value+grad), configs[3] in hard_sigmoid validity, `cfg5` (configs[4]: MinPath / FermatPath forward and
            value+grad+VJP), `api` (the reference's entry point, PCIe included), and `parity`: mismatch counts
            of the timed configuration's map against the oracle (committed full-map row CRCs + the cpu_baseline's rows).
cpu_baseline = the oracle's C/OpenMP restatement on a bounded row sample of the same grid (rank 0, N = 1).
Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VECTOR_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBPS = 8000.0
WORKLOADS = {
    # name: (walls, grid, max_order, BASELINE.json config, scaling at N > 1)
    "cfg2": (50, 1024, 2, "configs[1]", "weak"),
    "cfg4": (200, 2048, 3, "configs[3]", "strong"),
}


def workload(n_walls=50, grid=1024, seed=1234, rows=None):
    """layout of Scene.random_uniform_scene with a NumPy PRNG; grid = linspace(0, 1)."""
    pts = np.random.default_rng(seed).random((1 + 2 * n_walls + 1, 2), dtype=np.float32)
    tx = pts[0].copy()
    walls = pts[1 : 1 + 2 * n_walls].reshape(n_walls, 2, 2).copy()
    x = np.linspace(0.0, 1.0, grid).astype(np.float32)
    y = np.linspace(0.0, 1.0, rows or grid).astype(np.float32)
    X, Y = np.meshgrid(x, y)
    return tx, walls, X, Y


def num_candidates(n_walls, min_order, max_order):
    return sum(1 if k == 0 else n_walls * (n_walls - 1) ** (k - 1) for k in range(min_order, max_order + 1))


def unpruned_flop_per_rx(n_walls, min_order, max_order, approx):
    """FLOP(k, N) = (k+1) N F_seg + 74 k + 23 per candidate, nothing skipped."""
    f_seg = 41 if approx else 17
    total = 0
    for k in range(min_order, max_order + 1):
        c_k = 1 if k == 0 else n_walls * (n_walls - 1) ** (k - 1)
        total += c_k * ((k + 1) * n_walls * f_seg + 74 * k + 23)
    return total


def executed_flop(stats, approx):
    """Prices the executed-work counters (d2d_power_map_stats; one count = one 64-lane wave) with SURVEY.md
    section 8(d)'s per-unit figures: solver 16k + on_objects 20k per evaluated candidate, loss 29k,
    F_seg per evaluated segment/wall test, length/power/validity 9k + 23.  Returns (reference work, culling work):
    the culling -- per level 4 vertex evaluations of ~30 FLOP in each of the 64 lanes -- is work the reference does not
    have; its price is this repository's own."""
    f_seg = 41 if approx else 17
    s = [int(v) for v in stats]
    per_wave = s[6] * 36 + s[7] * 29 + s[4] * f_seg + (s[8] - s[3]) * 9 + s[3] * 23
    return per_wave * 64, s[9] * 120 * 64


def usable_cores():
    """Cores this process may really use: affinity mask, capped by a cgroup v2 CPU quota if any."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(tx, walls, X, Y, max_order, approx, budget_rows=64):
    """Oracle (C restatement, OpenMP, all usable host cores) timed on a bounded row sample of the same grid.
    Returns (the cpu_baseline object, the sampled row indices, the oracle's map on those rows).  `value` = every candidate of
    every cell fully evaluated, as the reference does (prune 0); `value_pruned` = the same oracle with its exact per-cell
    shortcuts (prune 2: leaves a candidate at the first wall whose on_objects is exactly 0, stops the occlusion tests at the
    first exact hit) -- the like-for-like figure beside a GPU sweep that culls."""
    from oracle import c_oracle as CO

    CO.build()
    cores = min(CO.max_threads(), usable_cores())
    rows = np.linspace(0, X.shape[0] - 1, budget_rows).astype(int)
    Xs, Ys = np.ascontiguousarray(X[rows]), np.ascontiguousarray(Y[rows])
    t0 = time.perf_counter()
    ref = CO.power_map(walls, tx, Xs, Ys, min_order=0, max_order=max_order, approx=approx, prune=False, nthreads=cores)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref2 = CO.power_map(walls, tx, Xs, Ys, min_order=0, max_order=max_order, approx=approx, prune=2, nthreads=cores)
    dt2 = time.perf_counter() - t0
    cands = Xs.size * num_candidates(walls.shape[0], 0, max_order)
    return {
        "value": cands / dt,
        "unit": "candidates/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{budget_rows} of {X.shape[0]} grid rows x {X.shape[1]} columns ({Xs.size} RX cells, "
                  f"{cands:.3g} candidates, every candidate fully evaluated), C/OpenMP restatement of DiffeRT2d v0.4.0 "
                  f"(the JAX reference cannot be installed here), {dt:.1f} s",
        "value_pruned": cands / dt2,
        "pruned_what": f"the same rows with the oracle's exact per-cell shortcuts (prune=2), {dt2:.2f} s; maps identical: "
                       f"{bool(np.array_equal(ref, ref2, equal_nan=True))}",
    }, rows, ref


def committed_traffic(approx):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the newest committed PMC summary under profiles/
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, scripts/profile_gpu.sh) -- not measured inside this run."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_a{int(approx)}_pmc.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        val = lambda k: float(d[k]["mean_per_dispatch"] if isinstance(d[k], dict) else d[k])  # noqa: E731
        fetch, write = val("FETCH_SIZE"), val("WRITE_SIZE")
    except (OSError, ValueError, KeyError, TypeError):
        return None, None
    return (fetch + write) * 1024.0, (f"profiles/{os.path.basename(files[-1])}: FETCH_SIZE {fetch:.0f} KiB + WRITE_SIZE {write:.0f} KiB per dispatch "
                                      f"as counted (4-byte-per-lane loads; the guide's x2 FETCH correction, calibrated on 16-B streaming reads, "
                                      f"would make it {(2 * fetch + write) * 1024 / 1e6:.1f} MB), kernel {d.get('_meta', {}).get('kernel', '?')}")


def row_crcs(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return np.array([zlib.crc32(a[i].tobytes()) for i in range(a.shape[0])], dtype=np.uint32)


def moving_transmitters(tx, n, step=0.01, seed=7):
    """A transmitter that moves a little every step (an optimisation loop's caller, examples/plot_power_optimize.py:78-93):
    random walk inside [0.05, 0.95]^2 starting at the workload's transmitter."""
    rng = np.random.default_rng(seed)
    out, p = [], np.asarray(tx, np.float64).copy()
    for _ in range(n):
        p = np.clip(p + rng.normal(0.0, step, 2), 0.05, 0.95)
        out.append(p.astype(np.float32))
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N children of this script, one per GPU, with the environment
    torch.distributed.run would give them, relay rank 0's stdout (the JSON line) and return the worst exit code.  The
    parent never initialises HIP (children are fresh processes: subprocess.Popen, never os.exec*)."""
    import shutil
    import socket
    import subprocess
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rdzv = tempfile.mkdtemp(prefix="d2d_rdzv_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(n), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), D2D_RDZV_DIR=rdzv, D2D_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0 prints the JSON line: into a file, so that nobody has to keep a pipe drained
        out = open(os.path.join(rdzv, "rank0.out"), "wb") if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, stdout=out))
        if r == 0:
            out.close()
    worst, failed_at = 0, None
    out0 = b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0:
                    worst = rc if worst == 0 or abs(rc) > abs(worst) else worst
                    failed_at = failed_at or time.time()
            # a rank that failed leaves the others waiting at the rendezvous or in a collective: give them a moment, then end them
            if failed_at and pending and time.time() - failed_at > 20.0:
                for r in pending:
                    procs[r].kill()  # (the exact processes started above)
            if pending:
                time.sleep(0.05)
        with open(os.path.join(rdzv, "rank0.out"), "rb") as f:
            out0 = f.read()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        shutil.rmtree(rdzv, ignore_errors=True)
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    return worst if worst >= 0 else 128 + abs(worst)


def api_leg(tx, walls, resident_ms, sizes=(300, 1024), n_calls=12):
    """Wall time of the reference's own entry point (scene.py:1803-1826) on the Python mirror, PCIe included: every call
    hands X, Y and the objects over again and returns the map as a host array.  `first` = the first call on a context that
    has never seen this grid (upload, allocation of the launch's buffers, no work history); `steady` = the median of the
    following calls.  Arrays as scene.grid() returns them (immutable: recognised by identity) and plain writable arrays
    (recognised byte for byte against the context's host copy of their 2 x 4 m n bytes)."""
    from differt2d_amd.geometry import Point
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power

    scene = Scene.from_walls_array(walls).with_transmitters(tx=Point(xy=tx))
    ctx = scene._ctx()
    out = {"what": "Scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True, max_order=2), hard "
                   "validity, wall time per call incl. PCIe both ways (host arrays in, host array out); steady = median of "
                   f"{n_calls - 2} calls after the first two", "resident_ms_per_step": resident_ms}
    for g in sizes:
        x = np.linspace(0.0, 1.0, g).astype(np.float32)
        entry = {}
        for label, freeze in (("immutable_arrays", True), ("writable_arrays", False)):
            X, Y = np.meshgrid(x, x)
            if freeze:
                X.setflags(write=False)
                Y.setflags(write=False)
            ts = []
            for _ in range(n_calls):
                t0 = time.perf_counter()
                Z = scene.accumulate_on_receivers_grid_over_paths(X, Y, fun=received_power, reduce_all=True, max_order=2)
                ts.append((time.perf_counter() - t0) * 1e3)
            assert Z.shape == X.shape
            entry[label] = {"first_ms": ts[0], "steady_ms": float(np.median(ts[2:])), "min_ms": float(np.min(ts[2:]))}
        # the parts of a steady call: the sweep alone (launch -> synchronize) and the download alone
        p = None
        from differt2d_amd.engine import make_params as mk

        p = mk(min_order=0, max_order=2)
        ds, ls = [], []
        for _ in range(n_calls):
            t0 = time.perf_counter()
            ctx.launch(p, tx)
            ctx.synchronize()
            ls.append((time.perf_counter() - t0) * 1e3)
            t0 = time.perf_counter()
            ctx.get_map()
            ds.append((time.perf_counter() - t0) * 1e3)
        entry["launch_and_sync_ms"] = float(np.median(ls[2:]))
        entry["download_ms"] = float(np.median(ds[2:]))
        entry["grid_reuses"] = ctx.grid_reuses()
        out[f"{g}x{g}"] = entry
    return out


def reference_harness_leg(n_calls=30):
    """The reference's OWN benchmark workload, the only one anybody holding a DiffeRT2d install (CodSpeed, GitHub runners) can put
    a number beside (reference tests/benchmarks/test_scene.py:9-29, fixtures tests/benchmarks/conftest.py:8-9, tests/conftest.py:14-20):
    `Scene.basic_scene()` (7 walls, one transmitter, one receiver), `X, Y = scene.grid(n)` for n in {5, 25, 50}, and
    `scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=approx, key=key)` with
    the defaults min_order = 0, max_order = 1 (8 candidates per cell) for approx in {False, True} -- through the Python mirror of
    that entry point, wall time per call, host arrays in and host array out (what pytest-benchmark's lambda times; the reference
    adds .block_until_ready()).  `first` = the first call of this grid and mode on the scene's context.  Beside it the CPU
    restatement (oracle/d2d_oracle.c, all usable cores) on the same call, and the parity of the two maps (bit for bit)."""
    from differt2d_amd.random import PRNGKey
    from differt2d_amd.scene import Scene
    from differt2d_amd.utils import received_power
    from oracle import c_oracle as CO

    scene = Scene.basic_scene()
    key = PRNGKey(1234)
    walls = np.stack([np.asarray(o.xys, np.float32) for o in scene.objects])
    rx = np.asarray(next(iter(scene.receivers.values())).xy, np.float32)
    out = {"what": "reference tests/benchmarks/test_scene.py: Scene.basic_scene() (7 walls), X, Y = scene.grid(n), "
                   "accumulate_on_transmitters_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=approx, key=key), "
                   "orders 0..1; wall ms per call through the Python mirror incl. PCIe both ways; cpu_ms = the C restatement of the same "
                   f"call, the faster of 1 thread and {usable_cores()} OpenMP threads (not JAX)", "calls": n_calls}
    # (all the GPU timings first: an OpenMP team of the oracle keeps spinning on the host's cores for a while after its parallel
    # region, and a call timed right behind it waited 95 ms for a core in the first version of this leg)
    maps = {}
    for n in (5, 25, 50):
        X, Y = scene.grid(n)
        for approx in (False, True):
            ts = []
            for _ in range(n_calls):
                t0 = time.perf_counter()
                Z = scene.accumulate_on_transmitters_grid_over_paths(X, Y, fun=received_power, reduce_all=True, approx=approx, key=key)
                ts.append((time.perf_counter() - t0) * 1e3)
            maps[n, approx] = Z
            out[f"n={n},approx={approx}"] = {"first_ms": ts[0], "steady_ms": float(np.median(ts[2:])), "min_ms": float(np.min(ts[2:]))}
    for n in (5, 25, 50):
        X, Y = scene.grid(n)
        for approx in (False, True):
            kw = dict(min_order=0, max_order=1, approx=approx, grid_role="tx")
            want = CO.power_map(walls, rx, X, Y, **kw)
            tc = {}
            for nt in (1, usable_cores()):  # one thread, then all usable cores (a 5 x 5 grid is less work than waking an OpenMP team up)
                tt = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    CO.power_map(walls, rx, X, Y, nthreads=nt, **kw)
                    tt.append((time.perf_counter() - t0) * 1e3)
                tc[nt] = float(np.median(tt))
            Z = maps[n, approx]
            out[f"n={n},approx={approx}"].update({"cpu_ms": min(tc.values()), "cpu_threads": min(tc, key=tc.get),
                                                  "cells_differing_from_oracle": int((~((Z == want) | (np.isnan(Z) & np.isnan(want)))).sum())})
    return out


def strong_leg(ctx, world, rank, distributed, do_gather, timed, which="cfg4", steps=5, approx=False):
    """A BASELINE.json configuration with its rows split over the ranks -- configs[3] (200 walls, 2048 x 2048, orders 0..3) or
    configs[1] (the timed workload's own 1024 x 1024 grid) -- the STRONG-scaling companion of the timed (weak) workload, in the
    same process, for the N = 1, 2, 4, 8 sequence.  Reference semantics: ONE assembled map (scene.py:1927-1953)."""
    from differt2d_amd.engine import make_params
    from differt2d_amd.parallel import RowShards

    n_walls, grid, max_order, cfg = WORKLOADS[which][:4]
    tx, walls, X, Y = workload(n_walls, grid)
    shards = RowShards(X.shape[0], world)
    ctx.set_scene(walls)
    ctx.set_grid(shards.take(X, rank), shards.take(Y, rank))
    p = make_params(min_order=0, max_order=max_order, approx=approx)

    def step():
        ctx.launch(p, tx)
        if world > 1:
            do_gather()

    # (three launches, each waited for: the list pools of the three rotating preparation sets grow to what this workload needs
    # -- hard_sigmoid lists of configs[3] are 577 MB -- before anything is timed)
    for _ in range(3):
        step()
        ctx.synchronize()
    wall, _ = timed(step, steps, 2)
    out = {}
    if world > 1:  # what the gather adds to a step
        wall0, _ = timed(lambda: ctx.launch(p, tx), steps, 2)
        out["ms_per_step_without_gather"] = wall0 * 1e3 / steps
    C = num_candidates(n_walls, 0, max_order)
    out.update({"workload": f"{n_walls} walls, {grid}x{grid} RX grid split over {world} rank(s), orders 0..{max_order} (C={C}), {'hard_sigmoid' if approx else 'hard'} validity; "
                            f"BASELINE.json {cfg}", "scaling": "strong", "steps": steps, "ms_per_step": wall * 1e3 / steps,
                "candidates_per_s": X.size * C / (wall / steps)})
    return out


def cfg5_leg(ctx, timed, steps=1000, n=5):
    """BASELINE.json configs[4]: Scene.square_scene() + RIS([[0.5, 0.3], [0.5, 0.7]], phi = pi / 4) + its two end points as
    diffraction vertices, scene.grid(n=300), order 1, MinPath with 1000 Adam steps, hard_sigmoid validity: the forward sweep
    and value + per-cell gradient + scene VJP (incl. the RIS's vertices and phi) through the Adam loop.  theta0 explicit (NumPy
    seed 1234, as tests/golden/cfg5_samples.npz).  Replaces the context's scene and grid."""
    from differt2d_amd.engine import make_params
    from differt2d_amd.geometry import RIS, objects_to_tables
    from differt2d_amd.scene import Scene

    scene = Scene.square_scene()
    ris = RIS(xys=[[0.5, 0.3], [0.5, 0.7]], phi=np.pi / 4)
    scene = scene.add_objects(ris, *ris.get_vertices())
    X, Y = scene.grid(300, 300)  # (the reference's scene.grid(n=300) is 50 x 300 cells: BASELINE.json's configs[4] says 300 x 300)
    tx = scene.transmitters["tx"].xy
    cands = scene.all_path_candidates(order=1)
    rng = np.random.default_rng(1234)
    theta0 = [rng.random(sum(o.parameters_count() for o in scene.get_interacting_objects(c)), dtype=np.float32) for c in cands]
    ctx.set_scene(*objects_to_tables(scene.objects))
    ctx.set_grid(X, Y)
    ctx.set_theta0(theta0)
    out = {"workload": f"square scene + RIS + its 2 vertices (7 objects), 300x300 RX grid, order 1 ({len(cands)} candidates), "
                       f"{steps} Adam steps per (cell, candidate), hard_sigmoid validity; BASELINE.json configs[4] on 1 GPU"}
    for solver in ("min", "fermat"):
        p = make_params(min_order=1, max_order=1, approx=True, solver=solver, steps=steps)
        w, _ = timed(lambda p=p: ctx.launch(p, tx), n, 1)
        out[f"{solver}path_forward"] = {"ms_per_step": w * 1e3 / n, "steps": n,
                                        "solver_iterations_per_s": X.size * len(cands) * steps / (w / n)}
        w, _ = timed(lambda p=p: ctx.launch_vg(p, tx, scene_vjp=True), n, 1)
        out[f"{solver}path_value_and_grad_and_vjp"] = {"ms_per_step": w * 1e3 / n, "steps": n}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 200 for cfg2, 5 for cfg4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 5 for cfg2, 1 for cfg4)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2")
    ap.add_argument("--grid", type=int, default=None, help="override the workload's grid size (cells per side and per GPU)")
    ap.add_argument("--walls", type=int, default=None)
    ap.add_argument("--max-order", type=int, default=None)
    ap.add_argument("--approx", type=int, default=0, help="validity of the TIMED map: 0 hard (reference default), 1 hard_sigmoid")
    ap.add_argument("--gather", choices=["root", "all"], default="root", help="N > 1: gather the map to rank 0, or all-gather it")
    ap.add_argument("--comm-prio", type=int, default=0, choices=[-1, 0, 1],
                    help="N > 1: priority of the stream the RCCL gather runs on beside the next sweep (lowest / default / highest): A/B")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other modes / moving-TX / value+grad / parity legs")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))  # before anything touches HIP

    wl_walls, wl_grid, wl_order, wl_cfg, wl_scaling = WORKLOADS[args.workload]
    n_walls = args.walls or wl_walls
    grid = args.grid or wl_grid
    max_order = wl_order if args.max_order is None else args.max_order
    steps = args.steps if args.steps is not None else (200 if args.workload == "cfg2" else 5)
    warmup = args.warmup if args.warmup is not None else (5 if args.workload == "cfg2" else 1)
    default_shape = (n_walls, grid, max_order) == (wl_walls, wl_grid, wl_order)

    from differt2d_amd import _lib as L
    from differt2d_amd.engine import Context, make_params
    from differt2d_amd.parallel import RowShards

    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if distributed else 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if distributed and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    n_dev = max(1, L.device_count())
    try:
        ctx = Context(local_rank % n_dev)  # one GPU per rank; ranks only share a device on a box with fewer GPUs than ranks
    except L.D2DError as e:
        # (no CPU fallback: without a GPU there is nothing to time)
        print(f"[bench rank {rank}] {e}", file=sys.stderr, flush=True)
        sys.exit(4)
    if distributed:
        ctx.set_option("comm_prio", args.comm_prio)
        # torch.distributed.run is only the launcher: rendezvous through /tmp, everything else through RCCL
        from differt2d_amd.parallel import file_rendezvous, file_rendezvous_cleanup

        try:
            ctx.comm_init(file_rendezvous(rank, world, Context.comm_unique_id), rank, world)
            ctx.comm_barrier()
        except Exception as e:  # noqa: BLE001
            print(f"[bench rank {rank}] RCCL communicator unavailable: {e}\n[bench rank {rank}] the configured workload ends every "
                  f"step with an RCCL gather; refusing to print a number without it", file=sys.stderr, flush=True)
            sys.exit(3)
        file_rendezvous_cleanup(rank)

    weak = wl_scaling == "weak"
    tx, walls, X, Y = workload(n_walls, grid, rows=grid * world if weak else grid)
    shards = RowShards(X.shape[0], world)
    Xl, Yl = shards.take(X, rank), shards.take(Y, rank)
    C = num_candidates(n_walls, 0, max_order)
    mode_kw = {"hard": dict(approx=False), "hard_sigmoid": dict(approx=True, function="hard_sigmoid"),
               "sigmoid": dict(approx=True, function="sigmoid")}
    timed_mode = "hard_sigmoid" if args.approx else "hard"
    params = make_params(min_order=0, max_order=max_order, **mode_kw[timed_mode])
    ctx.set_scene(walls)
    ctx.set_grid(Xl, Yl)
    gather = world > 1

    def barrier():
        ctx.synchronize()
        if distributed:
            ctx.comm_barrier()

    def allreduce_max(v):
        return float(ctx.comm_allreduce_host([v], "max")[0])

    def do_gather(grad=False):
        if args.gather == "root":
            ctx.comm_gather_map(root=0, grad=grad)  # on its own stream, overlapped with the next step's sweep
        else:
            ctx.comm_allgather_map(grad=grad)

    def step():
        ctx.launch(params, tx)
        if gather:
            do_gather()

    def timed(fn, n_steps, n_warmup):
        for _ in range(n_warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        ctx.timer_begin()
        for _ in range(n_steps):
            fn()
        stream_ms = ctx.timer_end()  # HIP events on the stream the kernels are launched on
        barrier()
        wall = time.perf_counter() - t0
        if distributed:
            wall = allreduce_max(wall)
        return wall, stream_ms / n_steps

    def kernel_ms_of(p, the_tx, n):
        """The sweep kernel on its own (HIP events around it, on the stream it runs on), averaged over n launches."""
        ctx.set_option("time_kernel", 1)
        ks = []
        for _ in range(n):
            ctx.launch(p, the_tx)
            ks.append(ctx.last_kernel_ms())
        ctx.set_option("time_kernel", 0)
        return float(np.mean(ks))

    # First launch after set_grid: allocates the library's device buffers, builds the scene-only wall-to-wall masks and
    # schedules the patches by the geometric proxy (no work history yet).  Setup, like set_scene / set_grid; reported.
    barrier()
    t0 = time.perf_counter()
    step()
    ctx.synchronize()
    first_launch_ms = (time.perf_counter() - t0) * 1e3
    barrier()

    wall, sequence_ms = timed(step, steps, warmup)
    ms_per_step = wall * 1e3 / steps
    gather_cost = None
    if gather:  # what the RCCL gather adds to a step: the same steps without it (never the headline: the workload includes it)
        wall0, _ = timed(lambda: ctx.launch(params, tx), steps, warmup)
        gather_cost = {"ms_per_step_with_gather": ms_per_step, "ms_per_step_without_gather": wall0 * 1e3 / steps,
                       "what": f"--gather {args.gather}: the map's gather runs on a second stream behind the sweep and overlaps the next "
                               "step's sweep; the difference is what it adds to a step"}
    kernel_ms = kernel_ms_of(params, tx, min(steps, 50))
    rccl_ranks, kernel_ms_per_rank = 1, [kernel_ms]
    if distributed:
        rccl_ranks = ctx.comm_count()  # ncclCommCount: what RCCL itself says
        mine = np.zeros(world)
        mine[rank] = kernel_ms
        kernel_ms_per_rank = [float(v) for v in ctx.comm_allreduce_host(mine, "sum")]
    cells_total = X.size
    cells_local = Xl.size
    small = (Xl.shape[0] + 7) // 8 * ((Xl.shape[1] + 7) // 8) <= 8192

    extras = {}
    if not args.no_extras and world == 1:
        n_x = max(10, min(steps, 100))
        # ---- the same map in every validity mode (the timed one included, for a like-for-like column)
        modes = {}
        for name, kw in mode_kw.items():
            p = make_params(min_order=0, max_order=max_order, **kw)
            n = n_x if name != "sigmoid" else max(3, n_x // 10)
            w, _ = timed(lambda p=p: ctx.launch(p, tx), n, 2)
            modes[name] = {"ms_per_step": w * 1e3 / n, "kernel_ms": kernel_ms_of(p, tx, min(n, 10)), "steps": n,
                           "candidates_per_s": cells_total * C / (w / n)}
        extras["modes"] = modes
        # ---- a different transmitter every step: the work history behind the patch schedule is a few steps stale
        txs = moving_transmitters(tx, n_x + 5)
        it = iter(txs)
        w, _ = timed(lambda: ctx.launch(params, next(it)), n_x, 5)
        # the same positions, each swept five times in a row and the fifth launch timed (warm history for THAT position):
        # what the sequence would cost if the schedule always knew the current position's work
        ctx.set_option("time_kernel", 1)
        warm, stale = [], []
        for k, t_ in enumerate(txs[5:]):
            ctx.launch(params, txs[5 + k - 1] if k else tx)
            ctx.launch(params, t_)
            stale.append(ctx.last_kernel_ms())  # history from the previous position
            for _ in range(4):  # (the pipelined preparation sorts by the history of three launches back)
                ctx.launch(params, t_)
            warm.append(ctx.last_kernel_ms())
        ctx.set_option("time_kernel", 0)
        extras["moving_tx"] = {"ms_per_step": w * 1e3 / n_x, "steps": n_x, "mode": timed_mode,
                               "kernel_ms_stale_history": float(np.mean(stale)), "kernel_ms_warm_history": float(np.mean(warm)),
                               "kernel_ms_warm_min_max": [float(np.min(warm)), float(np.max(warm))],
                               "what": "random walk of the transmitter (sigma 0.01 per step), a different position every step; "
                                       "shadow masks and schedule rebuilt per launch as always, work history from three positions back (the "
                                       "preparation of a launch is sorted while the previous sweeps run). "
                                       "kernel_ms_warm_history: the same positions with each position's own history (the cost of the "
                                       "positions themselves: how much of the scene a transmitter sees varies a lot along the walk); "
                                       "kernel_ms_stale_history (history from the previous position) - kernel_ms_warm_history = what a "
                                       "stale history costs"}
        # ---- a launch right after set_grid with everything allocated: geometric-proxy schedule, no cut-in-four
        cold = []
        for _ in range(5):
            ctx.set_grid(Xl, Yl)
            ctx.synchronize()
            t0 = time.perf_counter()
            ctx.launch(params, tx)
            ctx.synchronize()
            cold.append((time.perf_counter() - t0) * 1e3)
        extras["first_launch"] = {"ms_incl_allocation_and_scene_masks": first_launch_ms, "ms_cold_schedule": float(np.median(cold)),
                                  "what": "wall time launch -> synchronize of one map; the first figure is the very first launch of "
                                          "the process's context, the second a launch after set_grid with buffers and masks in place "
                                          "(no work history: geometric-proxy schedule)"}
        # ---- value + gradient (BASELINE.json configs[2]) in the timed validity mode: the default sweep (tile culling + the NaN
        # scan: NaN positions identical to the reference's), the exhaustive cross-check, and the sweep without the scan (A/B)
        vg = {}
        for label, strict, scan in (("default", False, 1), ("strict_nan", True, 1), ("without_nan_scan", False, 0)):
            p = make_params(min_order=0, max_order=max_order, strict_nan=strict, **mode_kw[timed_mode])
            n = max(3, n_x // 2) if not strict else 5
            ctx.set_option("nan_scan", scan)
            w, _ = timed(lambda p=p: ctx.launch_vg(p, tx, scene_vjp=True), n, 2)
            ctx.set_option("time_kernel", 1)
            ks = []
            for _ in range(min(n, 10)):
                ctx.launch_vg(p, tx, scene_vjp=True)
                ks.append(ctx.last_kernel_ms())
            ctx.set_option("time_kernel", 0)
            vg[label] = {"ms_per_step": w * 1e3 / n, "kernels_ms": float(np.mean(ks)), "steps": n, "candidates_per_s": cells_total * C / (w / n)}
        ctx.set_option("nan_scan", 1)
        vg["what"] = ("value + per-cell d/d rx + VJP w.r.t. TX position and wall end points (hand-derived reverse mode). `default`: "
                      "candidates that tile culling proves invalid are not evaluated; the reference's autodiff NaN positions -- an exact "
                      "zero in the backward scan of ANY candidate, a zero-length segment in the loss, a segment of exactly (-eps, -eps) "
                      "in path_length -- come from a scan of their own and coincide with the exhaustive kernel's on the whole map "
                      "(tests/test_gpu_grad.py::test_cfg3_full_map_nan_positions_equal_the_exhaustive_kernels). `strict_nan`: every "
                      "candidate of every cell evaluated (the cross-check). `without_nan_scan`: round 3's default (A/B; only the "
                      "evaluated candidates' NaN). kernels_ms: sweep + scan on their stream (HIP events)")
        extras["value_and_grad"] = vg
        # ---- the other value+grad figures README / DESIGN quote: hard_sigmoid and sigmoid validity, TX grids (forward and
        # value+grad: accumulate_on_transmitters_grid_over_paths, scene.py:1489-1648)
        more = {}
        for name in ("hard_sigmoid", "sigmoid"):
            p = make_params(min_order=0, max_order=max_order, **mode_kw[name])
            n = max(3, n_x // 4) if name != "sigmoid" else 3
            w, _ = timed(lambda p=p: ctx.launch_vg(p, tx, scene_vjp=True), n, 2)
            more[f"value_and_grad_{name}"] = {"ms_per_step": w * 1e3 / n, "steps": n}
        for name in ("hard", "hard_sigmoid"):
            p = make_params(min_order=0, max_order=max_order, grid_role=L.GRID_TX, **mode_kw[name])
            n = max(3, n_x // 4)
            w, _ = timed(lambda p=p: ctx.launch(p, tx), n, 2)
            more[f"tx_grid_forward_{name}"] = {"ms_per_step": w * 1e3 / n, "steps": n}
            w, _ = timed(lambda p=p: ctx.launch_vg(p, tx, scene_vjp=True), n, 2)
            more[f"tx_grid_value_and_grad_{name}"] = {"ms_per_step": w * 1e3 / n, "steps": n}
        more["tx_grid_exhaustive_fallbacks"] = ctx.txg_fallbacks()
        more["what"] = ("the timed workload's scene and grid; value_and_grad_* = value + per-cell gradient + scene VJP (default sweep: "
                        "culling + NaN scan); tx_grid_* = the grid cells are transmitters and the workload's transmitter the fixed "
                        "receiver; wall time per step, back to back, resident inputs")
        extras["more_modes"] = more
    elif not args.no_extras:
        def step_vg():
            ctx.launch_vg(params, tx, scene_vjp=True)
            if gather:
                do_gather()
                do_gather(grad=True)
                ctx.comm_allreduce_vjp()

        n = max(3, steps // 4)
        w, _ = timed(step_vg, n, 1)
        extras["value_and_grad"] = {"culled": {"ms_per_step": w * 1e3 / n, "steps": n, "candidates_per_s": cells_total * C / (w / n)},
                                    "what": "value + gradient maps gathered, scene VJP all-reduced, every step"}

    final_map = None
    if rank == 0:
        ctx.launch(params, tx)
        final_map = ctx.get_map()  # rank 0's shard of the timed configuration (N = 1: the whole map)
        stats = ctx.launch_stats(params, tx)  # instrumented build, outside the timed region (deterministic counts)
        list_stats = ctx.debug_region_stats()  # the region candidate lists of that launch (none for orders < 2)
    if gather_cost is not None:
        extras["gather_cost"] = gather_cost
    if not args.no_extras and args.workload == "cfg2" and default_shape:
        # (collective at N > 1: every rank takes part; replaces the context's scene and grid, hence after everything else)
        if world > 1:  # (at N = 1 the timed workload IS cfg2 on one rank)
            extras["strong_cfg2"] = strong_leg(ctx, world, rank, distributed, do_gather, timed, "cfg2", steps=min(steps, 50))
        extras["strong_cfg4"] = strong_leg(ctx, world, rank, distributed, do_gather, timed, "cfg4")
    if not args.no_extras and world == 1 and args.workload == "cfg2" and default_shape:
        extras["strong_cfg4_hard_sigmoid"] = strong_leg(ctx, world, rank, distributed, do_gather, timed, "cfg4", steps=3, approx=True)
        extras["cfg5"] = cfg5_leg(ctx, timed)
        extras["api"] = api_leg(tx, walls, ms_per_step)
        extras["reference_harness"] = reference_harness_leg()

    if rank == 0:
        approx = timed_mode != "hard"
        flop_ref, flop_cull = executed_flop(stats, approx)
        flop_unpruned = unpruned_flop_per_rx(n_walls, 0, max_order, approx) * cells_local
        per_s = 1.0 / (kernel_ms * 1e-3) / 1e12
        achieved = (flop_ref + flop_cull) * per_s
        traffic_bytes, traffic_src = committed_traffic(approx) if (args.workload == "cfg2" and default_shape and world == 1) else (None, None)
        parity = None
        line = {
            "metric": "ray-path candidates/s",
            "value": cells_total * C / (ms_per_step * 1e-3),
            "unit": "candidates/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "kernel_ms_per_rank": kernel_ms_per_rank,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": wl_scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n_walls} random walls (NumPy seed 1234), 1 TX, {X.shape[0]}x{X.shape[1]} RX grid over the "
                            f"unit square ({Xl.shape[0]}x{Xl.shape[1]} per GPU), orders 0..{max_order} (C={C} candidates per "
                            f"cell), {timed_mode} validity, received_power; BASELINE.json {wl_cfg}",
                "setup": "scene and grid resident in HBM; 1 untimed launch (buffer allocation, scene-only masks) before the warmup "
                         "steps, in which what depends on the scene and the grid only is built once (wall-to-wall masks, regions' "
                         "boxes, last-segment masks of the leaf regions: DESIGN.md Phase 0d) -- a transmitter that moves keeps them; "
                         "every timed step sweeps the same transmitter (see moving_tx for a different one every step) and "
                         "rebuilds everything that depends on it: shadow masks, region candidate lists, patch schedule (on a side "
                         "stream beside the previous step's sweep, as in any back-to-back sequence of launches: DESIGN.md section 4)",
                "sharding": f"{world} rank(s), 8-row blocks round-robin" + (f", comm stream priority {args.comm_prio}" if world > 1 else "")
                            + (f"; 1 RCCL {'gather to rank 0 (ncclSend/ncclRecv)' if args.gather == 'root' else 'all-gather'} of the "
                               f"value map per step, overlapped with the next step's sweep" if gather else ""),
            },
            "roofline": {
                "bound": "valu_fp32",
                "achieved": achieved,
                "peak": PEAK_FP32_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                "fracs": {
                    "executed_with_culling": achieved / PEAK_FP32_VECTOR_TFLOPS,
                    "reference_work_only": flop_ref * per_s / PEAK_FP32_VECTOR_TFLOPS,
                    "unpruned_equivalent": flop_unpruned * per_s / PEAK_FP32_VECTOR_TFLOPS,
                    "pmc_lane_ops": None,
                    "note": "all over kernel_ms; frac = executed_with_culling; pmc_lane_ops (64 x SQ_INSTS_VALU / time / peak) needs "
                            "rocprofv3 PMC passes and is not measured inside this run: see profiles/*_summary.md",
                },
                "traffic": traffic_bytes,
                "traffic_source": traffic_src,
                "kernel": "d2d::power_fwd_split_kernel" if small else "d2d::power_fwd_kernel",
                "kernel_ms": kernel_ms,
                # + shadow masks, region candidate lists (region_list_kernel, region_refine_kernel), patch schedule, the
                # (normally empty) queue of patches left to the enumerating kernel: 7 small kernels, 2 memsets
                "launch_sequence_ms": sequence_ms,
                "region_lists": list_stats,
                "algorithmic_flop_per_launch": flop_ref + flop_cull,
                "reference_work_flop_per_launch": flop_ref,
                "culling_flop_per_launch": flop_cull,
                "unpruned_flop_per_launch": flop_unpruned,
                "executed_lane_units": {
                    "cull_levels": int(stats[9]) * 64, "candidates_exact": int(stats[0]) * 64, "reached_loss": int(stats[1]) * 64,
                    "reached_occlusion": int(stats[2]) * 64, "reached_fun": int(stats[3]) * 64,
                    "segment_tests": int(stats[4]) * 64, "exact_divide_tests": int(stats[5]) * 64,
                },
                "hbm_algorithmic_bytes": cells_local * 12,
                "hbm_algorithmic_GBps": cells_local * 12 / (kernel_ms * 1e-3) / 1e9,
                "hbm_peak_GBps": PEAK_HBM_GBPS,
            },
        }
        line.update(extras)
        if world == 1 and not args.no_extras:
            parity = {"what": "cells / rows of the timed configuration's map that differ from the oracle's (expected 0)"}
            gold_path = os.path.join(ROOT, "tests", "golden", "cfg2_fullmap_crc.npz")
            if args.workload == "cfg2" and default_shape and os.path.exists(gold_path):
                gold = np.load(gold_path)
                key = f"rx_{'hsig' if approx else 'hard'}_power_crc"
                parity["full_map_rows_compared"] = int(final_map.shape[0])
                parity["full_map_rows_mismatching"] = int((row_crcs(final_map) != gold[key]).sum())
                parity["full_map_reference"] = "tests/golden/cfg2_fullmap_crc.npz (C oracle, every cell, one CRC-32 per row)"
        if not args.no_cpu_baseline and world == 1:
            base, rows, ref = cpu_baseline(tx, walls, X, Y, max_order, approx)
            line["cpu_baseline"] = base
            if parity is not None:
                got = final_map[rows]
                parity["sampled_cells_compared"] = int(ref.size)
                parity["sampled_cells_mismatching"] = int((~((got == ref) | (np.isnan(got) & np.isnan(ref)))).sum())
                parity["sampled_reference"] = "the cpu_baseline's own output (C oracle, every candidate fully evaluated)"
        if parity is not None:
            line["parity"] = parity
        print(json.dumps(line), flush=True)

    if distributed:
        barrier()
        ctx.comm_destroy()
    ctx.close()


if __name__ == "__main__":
    main()
