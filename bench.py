#!/usr/bin/env python3
"""
bench.py -- ray-path candidates/s of the fused power-map sweep (BASELINE.json metric).

A "step" = one forward power map of the workload: 50 random walls, 1 TX, 1024 x 1024 RX grid,
orders 0..2 (C = 2501 candidates per RX cell, BASELINE.json configs[1]), inputs resident in HBM.
Prints ONE JSON line (rank 0).  N > 1: launched by torch.distributed.run, RX rows sharded over
ranks (weak scaling = every rank sweeps a full 1024^2 grid shard of a N-times taller grid).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBPS = 8000.0


def workload(n_walls=50, grid=1024, seed=1234):
    """layout of Scene.random_uniform_scene with a NumPy PRNG."""
    pts = np.random.default_rng(seed).random((1 + 2 * n_walls + 1, 2), dtype=np.float32)
    tx = pts[0].copy()
    walls = pts[1 : 1 + 2 * n_walls].reshape(n_walls, 2, 2).copy()
    x = np.linspace(0.0, 1.0, grid).astype(np.float32)
    X, Y = np.meshgrid(x, x)
    return tx, walls, X, Y


def algorithmic_flop_per_rx(n_walls, min_order, max_order, approx):
    """FLOP(k, N) = (k+1) N F_seg + 74 k + 23, unpruned."""
    f_seg = 41 if approx else 17
    total = 0
    for k in range(min_order, max_order + 1):
        c_k = 1 if k == 0 else n_walls * (n_walls - 1) ** (k - 1)
        total += c_k * ((k + 1) * n_walls * f_seg + 74 * k + 23)
    return total


def executed_flop(stats, approx):
    """Prices the kernel's executed-work counters (include/d2d.h, d2d_power_map_stats; one count =
    one 64-lane wave) with SURVEY.md section 8(d)'s per-unit figures: solver 16k + on_objects 20k per
    evaluated candidate, loss 29k, F_seg per evaluated segment/wall test, length/power/validity 9k + 23."""
    f_seg = 41 if approx else 17
    s = [int(v) for v in stats]
    per_wave = s[6] * 36 + s[7] * 29 + s[4] * f_seg + (s[8] - s[3]) * 9 + s[3] * 23
    return per_wave * 64


def num_candidates(n_walls, min_order, max_order):
    return sum(1 if k == 0 else n_walls * (n_walls - 1) ** (k - 1) for k in range(min_order, max_order + 1))


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


def cpu_baseline(tx, walls, X, Y, max_order, approx, budget_rows=24):
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
                  f"{cands:.3g} candidates), C/OpenMP restatement of DiffeRT2d v0.4.0 (not JAX), {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--walls", type=int, default=50)
    ap.add_argument("--max-order", type=int, default=2)
    ap.add_argument("--approx", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from differt2d_amd.engine import Context, make_params

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    tx, walls, X, Y = workload(args.walls, args.grid)
    C = num_candidates(args.walls, 0, args.max_order)
    params = make_params(min_order=0, max_order=args.max_order, approx=bool(args.approx))

    ctx = Context(local_rank)
    ctx.set_scene(walls)
    ctx.set_grid(X, Y)
    for _ in range(args.warmup):
        ctx.launch(params, tx)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.timer_begin()
    for _ in range(args.steps):
        ctx.launch(params, tx)
    kernel_ms = ctx.timer_end() / args.steps
    ctx.synchronize()
    wall = time.perf_counter() - t0
    ms_per_step = wall * 1e3 / args.steps

    cells = X.size
    stats = ctx.launch_stats(params, tx)  # instrumented build, outside the timed region (deterministic counts)
    flop_unpruned = algorithmic_flop_per_rx(args.walls, 0, args.max_order, bool(args.approx)) * cells
    flop = executed_flop(stats, bool(args.approx))
    achieved_tflops = flop / (kernel_ms * 1e-3) / 1e12
    line = {
        "metric": "ray-path candidates/s",
        "value": cells * C / (ms_per_step * 1e-3),
        "unit": "candidates/s",
        "n_gpus": args.gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.walls} random walls (NumPy seed 1234), 1 TX, {args.grid}x{args.grid} RX grid, "
                        f"orders 0..{args.max_order} (C={C} candidates/cell), "
                        f"{'approx hard_sigmoid alpha=100' if args.approx else 'hard'} validity, received_power",
        },
        "roofline": {
            "bound": "valu_fp32",
            "achieved": achieved_tflops,
            "peak": PEAK_FP32_VECTOR_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved_tflops / PEAK_FP32_VECTOR_TFLOPS,
            "traffic": None,
            "kernel_ms": kernel_ms,
            "algorithmic_flop_per_launch": flop,
            "unpruned_flop_per_launch": flop_unpruned,
            "unpruned_equiv_TFLOPs": flop_unpruned / (kernel_ms * 1e-3) / 1e12,
            "executed": {
                "candidates": int(stats[0]) * 64, "reached_loss": int(stats[1]) * 64,
                "reached_occlusion": int(stats[2]) * 64, "reached_fun": int(stats[3]) * 64,
                "segment_tests": int(stats[4]) * 64, "exact_divide_tests": int(stats[5]) * 64,
            },
            "hbm_algorithmic_bytes": cells * 12,
            "hbm_algorithmic_GBps": cells * 12 / (kernel_ms * 1e-3) / 1e9,
            "hbm_peak_GBps": PEAK_HBM_GBPS,
        },
    }
    if not args.no_cpu_baseline and rank == 0 and args.gpus == 1:
        line["cpu_baseline"] = cpu_baseline(tx, walls, X, Y, args.max_order, bool(args.approx))
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
