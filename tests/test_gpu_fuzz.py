"""Randomised differential tests (short versions of scripts/fuzz_parity.py): forward sweep vs the C oracle, and the
culled value+grad kernel vs the exhaustive one, on random scenes with lattice-snapped walls, corners, scales, offsets."""

import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forward_fuzz_against_oracle():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "120", "2024"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "0 mismatches" in out.stdout


def test_gradient_fuzz_against_the_c_gradient_oracle():
    """scripts/fuzz_parity.py --grad: value + per-cell gradient of the default (culled + NaN scan) sweep -- every fourth case
    the exhaustive kernel -- against oracle/d2d_oracle_grad.c (forward-mode duals; shares nothing with the kernels' adjoint)
    on random / lattice-snapped / scaled / offset scenes, all modes, all path functions, both grid roles, orders 0..3."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "--grad", "70", "2025"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert " 0 mismatches" in out.stdout


def test_solver_fuzz_against_the_c_oracle():
    """scripts/fuzz_opt.py: MinPath / FermatPath sweeps (values, and per-cell gradients through the Adam loop every other case)
    on random Wall / RIS / Vertex scenes against oracle/d2d_oracle_opt.c, on the cells the oracle calls well conditioned."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_opt.py"), "30", "2026"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert " 0 mismatches" in out.stdout


def test_culled_and_exhaustive_gradient_kernels_agree_on_random_scenes():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from fuzz_parity import random_case

    from differt2d_amd import _lib as L
    from differt2d_amd.engine import Context

    rng = np.random.default_rng(99)
    with Context(0) as ctx:
        done = 0
        while done < 48:  # (the oracle-backed gradient fuzz above is the main check since round 5; this one holds the two kernels together)
            walls, tx, X, Y, kw, allowed = random_case(rng)
            if len(walls) == 0:
                continue
            ctx.set_scene(walls)
            ctx.set_candidate_mask(allowed)
            role = L.GRID_TX if done % 3 == 2 else L.GRID_RX
            ctx.set_option("nan_scan", 2 if done % 5 == 4 else 1)  # (the scan's two shapes: the same flags)
            a = ctx.value_and_grads(tx, X, Y, strict_nan=False, grid_role=role, **kw)
            b = ctx.value_and_grads(tx, X, Y, strict_nan=True, grid_role=role, **kw)
            assert np.array_equal(a["value"], b["value"], equal_nan=True)
            # NaN positions coincide (the default sweep's NaN scan finds the reference's autodiff artefacts inside the
            # candidates its culling never evaluates), and everything else agrees
            for k in ("grad_rx", "tx_bar", "walls_bar"):
                assert np.array_equal(np.isnan(a[k]), np.isnan(b[k])), (k, done, kw)
            fin = np.isfinite(b["grad_rx"])
            scale = max(1e-30, float(np.abs(b["grad_rx"][fin]).max())) if fin.any() else 1.0
            assert np.abs(a["grad_rx"][fin] - b["grad_rx"][fin]).max(initial=0.0) <= 1e-5 * scale
            done += 1


def test_cut_in_four_handover_is_stable_under_repetition():
    """scripts/stress_heavy.py: hundreds of launches of the bench workload through the path where four workgroups hand a
    patch over through global memory -- every map must equal the uncut sweep's bit for bit."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_heavy.py"), "150"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "stress: 0 mismatching maps" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]
