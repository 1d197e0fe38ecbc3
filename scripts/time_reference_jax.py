#!/usr/bin/env python3
"""Times the REFERENCE itself -- DiffeRT2d v0.4.0 on JAX -- on the workloads bench.py measures, for anyone who has a JAX
install (SURVEY.md section 8d asks for this recipe).  It cannot run in this repository's build container or on its GPU
box: neither has jax / jaxlib / equinox / optax / differt-core, and there is no network to install them.  No number in
BASELINE.md, DESIGN.md or bench.py's output comes from this script; it exists so that the `cpu_baseline` figure (the C
restatement, kind "port") can be replaced by a measured JAX figure wherever the reference can be installed.

    pip install differt2d==0.4.0            # pulls jax, equinox, optax, differt-core
    JAX_PLATFORMS=cpu python scripts/time_reference_jax.py [--grid 64] [--walls 50] [--max-order 2] [--approx 0]

It mirrors the reference's own benchmark (tests/benchmarks/test_scene.py: accumulate_on_receivers_grid_over_paths with
fun = received_power under pytest-benchmark) and this repository's synthetic inputs (bench.py `workload`: the layout of
Scene.random_uniform_scene, scene.py:718-733, filled from NumPy's default_rng(1234) so that both sides see the same walls).
The reference evaluates every (cell, candidate) pair densely and loops over candidates in Python, so the full 1024 x 1024
order-2 workload (2.6e9 pairs, each 150 segment tests) is hours of CPU: run it on a sub-grid and scale by cells -- the
work per cell is constant -- and say so next to the number.
"""

import argparse
import json
import os
import time

import numpy as np


def workload(n_walls, grid, seed=1234):
    pts = np.random.default_rng(seed).random((1 + 2 * n_walls + 1, 2), dtype=np.float32)
    tx = pts[0].copy()
    walls = pts[1 : 1 + 2 * n_walls].reshape(n_walls, 2, 2).copy()
    x = np.linspace(0.0, 1.0, grid).astype(np.float32)
    X, Y = np.meshgrid(x, x)
    return tx, walls, X, Y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=64, help="cells per side of the timed sub-grid (bench.py: 1024)")
    ap.add_argument("--walls", type=int, default=50)
    ap.add_argument("--max-order", type=int, default=2)
    ap.add_argument("--approx", type=int, default=0)
    ap.add_argument("--grad", action="store_true", help="value_and_grad=True (BASELINE.json configs[2])")
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()

    try:
        import jax
        import jax.numpy as jnp
        from differt2d.geometry import Point
        from differt2d.scene import Scene
        from differt2d.utils import received_power
    except ImportError as e:  # the situation in this repository's containers
        raise SystemExit(f"the reference is not installed here ({e}); see this script's docstring")

    tx, walls, X, Y = workload(args.walls, args.grid)
    scene = Scene.from_walls_array(jnp.asarray(walls)).with_transmitters(tx=Point(xy=jnp.asarray(tx)))
    X, Y = jnp.asarray(X), jnp.asarray(Y)
    C = sum(1 if k == 0 else args.walls * (args.walls - 1) ** (k - 1) for k in range(args.max_order + 1))
    kw = dict(fun=received_power, reduce_all=True, min_order=0, max_order=args.max_order, approx=bool(args.approx),
              value_and_grad=args.grad)

    def run():
        out = scene.accumulate_on_receivers_grid_over_paths(X, Y, **kw)
        jax.block_until_ready(out)
        return out

    t0 = time.perf_counter()
    run()  # tracing + compilation of every per-candidate function
    first = time.perf_counter() - t0
    times = []
    for _ in range(args.repeat):
        t0 = time.perf_counter()
        run()
        times.append(time.perf_counter() - t0)
    best = min(times)
    print(json.dumps({
        "what": "DiffeRT2d v0.4.0 (JAX) accumulate_on_receivers_grid_over_paths, received_power, reduce_all",
        "jax": jax.__version__, "backend": jax.default_backend(), "devices": [str(d) for d in jax.devices()],
        "host_cores": len(os.sched_getaffinity(0)),
        "grid": args.grid, "walls": args.walls, "max_order": args.max_order, "approx": bool(args.approx), "grad": args.grad,
        "candidates_per_cell": C, "first_call_s": first, "best_s": best,
        "candidates_per_s": args.grid * args.grid * C / best,
        "full_1024_grid_estimate_s": best * (1024 / args.grid) ** 2,
    }))


if __name__ == "__main__":
    main()
