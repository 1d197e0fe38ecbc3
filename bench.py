#!/usr/bin/env python3
"""
bench.py -- ray-path candidates/s of the fused power-map sweep (BASELINE.json metric) on N MI355X.

Step      = one forward power map of the workload, inputs resident in HBM:
            50 random walls (NumPy seed 1234), 1 TX, 1024 x 1024 RX grid over the unit square, orders 0..2
            (C = 2501 candidates per cell) = BASELINE.json configs[1]; hard (reference default) validity.
N > 1     = launched by torch.distributed.run, one process per GPU.  Weak scaling: the grid becomes
            (1024 N) x 1024 cells over the same unit square, rows dealt to ranks in 8-row blocks round-robin
            (differt2d_amd/parallel.py), so every rank sweeps 1024 x 1024 cells; each step ends with ONE RCCL
            all-gather of the value map, on a second stream so that it overlaps the next step's sweep.  Control plane (rendezvous, barrier, max over
            ranks): RCCL too (d2d_comm_allreduce_host); rendezvous through a file in /tmp; torch is never imported.
value     = all cells of all ranks x C / wall time of the K timed steps (max over ranks).
roofline  = the kernel is FP32-VALU bound (~1e6 FLOP per HBM byte, no MFMA-shaped work).
            `achieved` prices the work the kernel actually executed (counters of its instrumented build, see
            include/d2d.h) with SURVEY.md's per-unit FLOP figures; `peak` is the 157.3 TFLOP/s FP32 vector peak.
cpu_baseline = the oracle's C/OpenMP restatement on a bounded row sample of the same grid (rank 0, N = 1).
Prints ONE JSON line on rank 0.
"""

import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VECTOR_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBPS = 8000.0


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
    F_seg per evaluated segment/wall test, length/power/validity 9k + 23."""
    f_seg = 41 if approx else 17
    s = [int(v) for v in stats]
    per_wave = s[6] * 36 + s[7] * 29 + s[4] * f_seg + (s[8] - s[3]) * 9 + s[3] * 23
    per_wave += s[9] * 120  # tile culling: per level 4 vertex evaluations of ~30 FLOP in each of the 64 lanes
    return per_wave * 64


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
    """Oracle (C restatement, OpenMP, all usable host cores) timed on a bounded row sample of the same grid."""
    from oracle import c_oracle as CO

    CO.build()
    cores = min(CO.max_threads(), usable_cores())
    rows = np.linspace(0, X.shape[0] - 1, budget_rows).astype(int)
    Xs, Ys = np.ascontiguousarray(X[rows]), np.ascontiguousarray(Y[rows])
    t0 = time.perf_counter()
    CO.power_map(walls, tx, Xs, Ys, min_order=0, max_order=max_order, approx=approx, prune=False, nthreads=cores)
    dt = time.perf_counter() - t0
    cands = Xs.size * num_candidates(walls.shape[0], 0, max_order)
    return {
        "value": cands / dt,
        "unit": "candidates/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{budget_rows} of {X.shape[0]} grid rows x {X.shape[1]} columns ({Xs.size} RX cells, "
                  f"{cands:.3g} candidates, every candidate fully evaluated), C/OpenMP restatement of DiffeRT2d v0.4.0 "
                  f"(the JAX reference cannot be installed here), {dt:.1f} s",
    }


def measured_traffic_bytes(approx):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc.json: FETCH_SIZE + WRITE_SIZE,
    KiB; 4-byte-per-lane accesses, for which the guide's x2 wide-read correction does not apply)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_a{int(approx)}_pmc.json")))
    if not files:
        return None
    try:
        pmc = json.load(open(files[-1]))
        return (pmc["FETCH_SIZE"]["mean_per_dispatch"] + pmc["WRITE_SIZE"]["mean_per_dispatch"]) * 1024.0
    except (KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--walls", type=int, default=50)
    ap.add_argument("--max-order", type=int, default=2)
    ap.add_argument("--approx", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grad", action="store_true", help="skip the value+grad (BASELINE.json configs[2]) timing")
    args = ap.parse_args()

    from differt2d_amd import _lib as L
    from differt2d_amd.engine import Context, make_params
    from differt2d_amd.parallel import RowShards

    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if distributed else 1
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if distributed and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not distributed and args.gpus != 1:
        raise SystemExit("N > 1 must be launched with torch.distributed.run (one process per GPU)")

    n_dev = max(1, L.device_count())
    ctx = Context(local_rank % n_dev)  # one GPU per rank; ranks only share a device on a box with fewer GPUs than ranks
    rccl_note = None
    host_comm = None  # control plane: RCCL (d2d_comm_allreduce_host) unless the communicator cannot be created
    if distributed:
        # torch.distributed.run is only the launcher: rendezvous through /tmp, everything else through RCCL
        from differt2d_amd.parallel import FileHostComm, file_rendezvous, file_rendezvous_cleanup

        try:
            ctx.comm_init(file_rendezvous(rank, world, Context.comm_unique_id), rank, world)
            ctx.comm_barrier()
        except Exception as e:  # noqa: BLE001 -- keep the sharded sweep measurable without the gather
            rccl_note = f"RCCL communicator unavailable ({str(e)[:200]}): shards timed without the all-gather, file barrier"
            print(f"[bench rank {rank}] {rccl_note}", file=sys.stderr, flush=True)
            host_comm = FileHostComm(rank, world)
            host_comm.barrier()
        file_rendezvous_cleanup(rank)

    tx, walls, X, Y = workload(args.walls, args.grid, rows=args.grid * world)
    shards = RowShards(X.shape[0], world)
    Xl, Yl = shards.take(X, rank), shards.take(Y, rank)
    C = num_candidates(args.walls, 0, args.max_order)
    params = make_params(min_order=0, max_order=args.max_order, approx=bool(args.approx))
    ctx.set_scene(walls)
    ctx.set_grid(Xl, Yl)

    gather = world > 1 and host_comm is None

    def barrier():
        ctx.synchronize()
        if host_comm is not None:
            host_comm.barrier()
        elif distributed:
            ctx.comm_barrier()

    def allreduce_max(v):
        return float((host_comm.allreduce([v], "max") if host_comm is not None else ctx.comm_allreduce_host([v], "max"))[0])

    def step():
        ctx.launch(params, tx)
        if gather:
            ctx.comm_allgather_map()  # on its own stream, overlapped with the next step's sweep

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        ctx.timer_begin()
        for _ in range(steps):
            fn()
        stream_ms = ctx.timer_end()  # HIP events on the stream the kernels are launched on
        barrier()
        wall = time.perf_counter() - t0
        if distributed:
            wall = allreduce_max(wall)
        return wall, stream_ms / steps

    # Setup, like set_scene / set_grid: one launch allocates the library's device buffers and builds the scene-only
    # wall-to-wall masks (the W warmup steps that follow are the contract's; with --warmup 0 the timed steps would
    # otherwise include hipMalloc calls).
    step()
    barrier()
    wall, sequence_ms = timed(step, args.steps, args.warmup)
    ms_per_step = wall * 1e3 / args.steps
    # the dominant kernel on its own (HIP events around it, on the stream it runs on), outside the timed region
    ctx.set_option("time_kernel", 1)
    ks = []
    for _ in range(args.steps):
        ctx.launch(params, tx)
        ks.append(ctx.last_kernel_ms())
    ctx.set_option("time_kernel", 0)
    kernel_ms = float(np.mean(ks))
    cells_total = X.size
    cells_local = Xl.size

    grad_info = None
    if not args.no_grad:
        def step_vg():
            ctx.launch_vg(params, tx, scene_vjp=True)
            if gather:
                ctx.comm_allgather_map()
                ctx.comm_allreduce_vjp()

        gwall, gkernel_ms = timed(step_vg, max(3, args.steps // 2), 1)
        gsteps = max(3, args.steps // 2)
        grad_info = {
            "what": "value + per-cell d/d rx + VJP w.r.t. TX position and wall end points (reverse-mode kernel), "
                    "BASELINE.json configs[2]",
            "ms_per_step": gwall * 1e3 / gsteps,
            "stream_ms_per_step": gkernel_ms,
            "candidates_per_s": cells_total * C / (gwall / gsteps),
        }

    if rank == 0:
        stats = ctx.launch_stats(params, tx)  # instrumented build, outside the timed region (deterministic counts)
        flop_exec = executed_flop(stats, bool(args.approx))
        flop_unpruned = unpruned_flop_per_rx(args.walls, 0, args.max_order, bool(args.approx)) * cells_local
        achieved = flop_exec / (kernel_ms * 1e-3) / 1e12
        line = {
            "metric": "ray-path candidates/s",
            "value": cells_total * C / (ms_per_step * 1e-3),
            "unit": "candidates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.walls} random walls (NumPy seed 1234), 1 TX, {X.shape[0]}x{X.shape[1]} RX grid over the "
                            f"unit square ({args.grid}x{args.grid} per GPU), orders 0..{args.max_order} (C={C} candidates per "
                            f"cell), {'approx hard_sigmoid alpha=100' if args.approx else 'hard'} validity, received_power; "
                            f"BASELINE.json configs[1]",
                "setup": "scene and grid resident in HBM; 1 untimed launch (buffer allocation, scene-only masks) before the warmup steps",
                "sharding": f"{world} rank(s), 8-row blocks round-robin"
                            + ("; 1 RCCL all-gather of the value map per step, overlapped with the next step's sweep" if gather else "")
                            + (f"; {rccl_note}" if rccl_note else ""),
            },
            "roofline": {
                "bound": "valu_fp32",
                "achieved": achieved,
                "peak": PEAK_FP32_VECTOR_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                "traffic": measured_traffic_bytes(args.approx),
                "kernel": "d2d::power_fwd_split_kernel" if (Xl.shape[0] + 7) // 8 * ((Xl.shape[1] + 7) // 8) <= 8192
                          else "d2d::power_fwd_kernel",
                "kernel_ms": kernel_ms,
                "launch_sequence_ms": sequence_ms,  # + shadow masks, patch schedule (4 small kernels, 2 memsets)
                "algorithmic_flop_per_launch": flop_exec,
                "unpruned_flop_per_launch": flop_unpruned,
                "unpruned_equiv_TFLOPs": flop_unpruned / (kernel_ms * 1e-3) / 1e12,
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
        if grad_info:
            line["value_and_grad"] = grad_info
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(tx, walls, X, Y, args.max_order, bool(args.approx))
        print(json.dumps(line), flush=True)

    if distributed:
        barrier()
        if host_comm is None:
            ctx.comm_destroy()
    ctx.close()


if __name__ == "__main__":
    main()
