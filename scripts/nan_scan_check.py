#!/usr/bin/env python3
"""The default value+grad sweep (tile culling + NaN scan) against the exhaustive kernel (strict_nan) on BASELINE configs[2] at
full size -- NaN positions of the per-cell gradient, of tx_bar and of walls_bar, and the finite values -- in both grid roles and
every validity mode, with the scan's counters and the kernels' times; then N random lattice-snapped scenes (fuzz_parity's
generator), where exact zeros are common.

    python scripts/nan_scan_check.py [n_fuzz_cases] [seed] [--no-full] [--crowded]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from conftest import random_scene  # noqa: E402
from fuzz_parity import crowded_case, random_case  # noqa: E402

from differt2d_amd import _lib as L  # noqa: E402
from differt2d_amd.engine import Context, make_params  # noqa: E402

F = np.float32


def compare(a, b, tag):
    """a: default, b: strict.  Returns the number of differences (0 = identical NaN pattern, equal finite values)."""
    bad = 0
    if not np.array_equal(a["value"], b["value"], equal_nan=True):
        print(f"  {tag}: VALUE maps differ")
        bad += 1
    for k in ("grad_rx", "tx_bar", "walls_bar"):
        na, nb = np.isnan(a[k]), np.isnan(b[k])
        if not np.array_equal(na, nb):
            print(f"  {tag}: {k}: NaN positions differ: default {int(na.sum())}, strict {int(nb.sum())}, default&~strict "
                  f"{int((na & ~nb).sum())}, strict&~default {int((nb & ~na).sum())}; first: {np.argwhere(na != nb)[:5].tolist()}")
            bad += 1
        fin = ~(na | nb)
        if fin.any():
            scale = max(1e-30, float(np.abs(b[k][fin]).max()))
            err = float(np.abs(a[k][fin].astype(np.float64) - b[k][fin]).max())
            # (infinities: equal or both non-finite)
            if not (err <= 1e-5 * scale or not np.isfinite(scale)):
                print(f"  {tag}: {k}: finite entries differ by {err:.3e} at scale {scale:.3e}")
                bad += 1
    return bad


def full_size():
    tx, walls = random_scene(50, seed=1234)
    x = np.linspace(0.0, 1.0, 1024).astype(F)
    X, Y = np.meshgrid(x, x)
    X.setflags(write=False)
    Y.setflags(write=False)
    total = 0
    for role, rname in ((L.GRID_RX, "rx"), (L.GRID_TX, "tx")):
        for mname, kw in (("hard", dict(approx=False)), ("hsig", dict(approx=True, function="hard_sigmoid")),
                          ("sig", dict(approx=True, function="sigmoid"))):
            with Context(0) as c:
                c.set_option("time_kernel", 1)
                c.set_option("nan_scan_stats", 1)
                c.set_scene(walls)
                kws = dict(min_order=0, max_order=2, grid_role=role, **kw)
                for _ in range(3):
                    a = c.value_and_grads(tx, X, Y, strict_nan=False, **kws)
                st = c.debug_nan_scan()
                # timing: launch -> synchronise, resident grid
                p = make_params(strict_nan=False, **kws)
                c.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    c.launch_vg(p, tx, scene_vjp=True)
                c.synchronize()
                ms = (time.perf_counter() - t0) / 20 * 1e3
                kms = c.last_kernel_ms()
                c.set_option("nan_scan", 0)
                c.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    c.launch_vg(p, tx, scene_vjp=True)
                c.synchronize()
                ms0 = (time.perf_counter() - t0) / 20 * 1e3
                kms0 = c.last_kernel_ms()
                old = c.value_and_grads(tx, X, Y, strict_nan=False, **kws)
                c.set_option("nan_scan", 1)
                t0 = time.perf_counter()
                b = c.value_and_grads(tx, X, Y, strict_nan=True, **kws)
                ms_strict = (time.perf_counter() - t0) * 1e3
            n_a, n_b, n_old = (int(np.isnan(g["grad_rx"]).any(-1).sum()) for g in (a, b, old))
            print(f"cfg3 {rname} {mname}: NaN cells default {n_a}, strict {n_b}, without the scan {n_old} | walls_bar NaN "
                  f"{int(np.isnan(a['walls_bar']).sum())}/{int(np.isnan(b['walls_bar']).sum())} | scan: {st} | step {ms:.3f} ms "
                  f"(kernels {kms:.3f}), without the scan {ms0:.3f} ({kms0:.3f}), strict call {ms_strict:.1f} ms", flush=True)
            total += compare(a, b, f"cfg3 {rname} {mname}")
    return total


def fuzz(n_cases, seed, crowded_only=False):
    rng = np.random.default_rng(seed)
    bad = 0
    done = 0
    nan_cases = 0
    t0 = time.time()
    with Context(0) as ctx:
        while done < n_cases:
            # every fourth case (--crowded: every case) is coarse and crowded -- 50 .. 200 walls under 16^2 .. 128^2 cells: the region
            # scan's list takes many rounds and its probe queue overflows; every eighth runs with tiny buffers on top
            crowded = crowded_only or done % 4 == 3
            walls, tx, X, Y, kw, allowed = crowded_case(rng) if crowded else random_case(rng)
            if len(walls) == 0:
                continue
            role = L.GRID_TX if done % 3 == 2 else L.GRID_RX
            ctx.set_option("nan_scan_wqcap", 64 if done % 8 == 7 else 0)
            ctx.set_option("nan_scan_rb", 2 if done % 8 == 7 else 0)
            ctx.set_option("region_lists", 0 if done % 7 == 6 else 1)
            ctx.set_option("nan_scan", 2 if done % 5 == 4 else 1)  # (one wave per patch / two levels: the same flags)
            ctx.set_option("sched_min_tiles", 1 if done % 4 < 2 else 1 << 40)
            ctx.set_scene(walls)
            ctx.set_candidate_mask(allowed)
            a = ctx.value_and_grads(tx, X, Y, strict_nan=False, grid_role=role, **kw)
            b = ctx.value_and_grads(tx, X, Y, strict_nan=True, grid_role=role, **kw)
            nb = compare(a, b, f"case {done} (role {role}, {len(walls)} walls, grid {X.shape}, {kw})")
            bad += 1 if nb else 0
            nan_cases += bool(np.isnan(b["grad_rx"]).any())
            done += 1
            if done % 100 == 0:
                print(f"fuzz: {done} cases, {bad} bad, {nan_cases} with NaN cells, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz: {done} cases (seed {seed}), {nan_cases} with NaN cells, {bad} mismatching")
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 200
    seed = int(args[1]) if len(args) > 1 else 0
    bad = 0
    if "--no-full" not in sys.argv:
        bad += full_size()
    if n > 0:
        bad += fuzz(n, seed, crowded_only="--crowded" in sys.argv)
    print("nan_scan_check:", "OK" if bad == 0 else f"{bad} FAILURES")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
