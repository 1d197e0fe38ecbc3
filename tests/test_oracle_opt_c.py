"""oracle/d2d_oracle_opt.c -- the C restatement of the MinPath / FermatPath sweeps (Wall / RIS / Vertex objects, optax.adam,
forward-mode duals for the objective's theta-gradient, second-order jets for the per-cell gradient) -- pinned to oracle/ref.py
(NumPy fp32 / fp64 for values, torch for the autodiff parts):

* everything that does not involve the objective's gradient is ref.py's BIT FOR BIT: the objective's values (MinPath's sum of
  evaluate_cartesian over Wall / RIS / Vertex, FermatPath's path_length), the Adam update given g (ref.adam_minimize's order),
  and -- with one solver step, whose recorded loss and validity involve no gradient yet -- whole maps' validity chain;
* g itself: forward-mode duals against torch's reverse mode, a few ulp (a backward pass rounds in its own order: "bit for bit"
  is not defined between any two gradient implementations);
* whole sweeps: in fp64 the two agree to 1e-12 (same chain, the gradient's rounding no longer matters); in fp32 on the cells
  the oracle calls well conditioned;
* the per-cell gradient: against torch double-backward autodiff of ref.py (reverse mode through the Adam loop), fp64.

CPU only.  This is what makes the C oracle usable as the checker of configs[4] at full size (tests/test_gpu_opt.py)."""

import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import ref as R

F = np.float32


def ris_scene():
    """square_scene + RIS([[0.5, 0.3], [0.5, 0.7]], phi = pi / 4) + its two end points as vertices (configs[4])."""
    sq = R.square_scene_walls()
    ris = np.array([[0.5, 0.3], [0.5, 0.7]], F)
    xys = np.concatenate([sq, ris[None], np.stack([ris[0], ris[0]])[None], np.stack([ris[1], ris[1]])[None]]).astype(F)
    kinds = [0, 0, 0, 0, 1, 2, 2]
    phis = [0.0, 0.0, 0.0, 0.0, np.pi / 4, 0.0, 0.0]
    return kinds, xys, phis


def ref_objs(kinds, xys, phis, xp=R.NUMPY):
    out = []
    for k, w, ph in zip(kinds, xys, phis):
        out.append(R.Obj(R.VERTEX, xp.asarray(w[0])) if k == 2 else R.Obj(k, xp.asarray(w), float(ph)))
    return out


def random_mixed_scene(rng, n):
    kinds = rng.integers(0, 3, n).tolist()
    xys = rng.random((n, 2, 2)).astype(F)
    for j, k in enumerate(kinds):
        if k == 2:
            xys[j, 1] = xys[j, 0]
    phis = (rng.random(n) * 3.0 - 1.5).tolist()
    return kinds, xys, phis


@pytest.mark.parametrize("solver", ["min", "fermat"])
def test_objective_values_bit_for_bit_and_gradients_within_ulps(solver):
    import torch

    rng = np.random.default_rng(11)
    worst = 0.0
    for case in range(60):
        n = int(rng.integers(2, 7))
        kinds, xys, phis = random_mixed_scene(rng, n)
        k = int(rng.integers(1, 4))
        cand = [int(rng.integers(0, n))]
        while len(cand) < k:
            c = int(rng.integers(0, n))
            if c != cand[-1]:
                cand.append(c)
        tx, rx = rng.random(2).astype(F), rng.random(2).astype(F)
        nu = sum(kinds[c] != 2 for c in cand)
        theta = rng.random(nu).astype(F)
        val, g = CO.opt_objective(kinds, xys, phis, tx, rx, cand, theta, solver=solver)
        # ref.py, NumPy fp32
        objs = [ref_objs(kinds, xys, phis)[c] for c in cand]
        pts = R.parametric_to_cartesian(objs, [F(t) for t in theta], tx, rx)
        want = R.path_length(pts) if solver == "fermat" else R.path_loss(objs, pts)
        assert np.float32(val) == np.float32(want), (case, solver, val, want)
        # gradient: torch reverse mode of the same chain, fp32 and fp64
        for dt, tol in (("float32", 64.0), ("float64", 1e-3)):
            tb = R.TorchBackend(dt)
            tobjs = [ref_objs(kinds, xys, phis, tb)[c] for c in cand]
            ths = [tb.asarray(np.asarray(t)).clone().requires_grad_(True) for t in theta]
            tp = R.parametric_to_cartesian(tobjs, ths, tb.asarray(tx), tb.asarray(rx), tb)
            loss = R.path_length(tp, tb) if solver == "fermat" else R.path_loss(tobjs, tp, tb)
            if nu == 0:
                continue
            gt = np.array([float(v) for v in torch.autograd.grad(loss, ths, allow_unused=True)])
            vv, gg = CO.opt_objective(kinds, xys, phis, tx, rx, cand, theta, solver=solver, dtype=dt)
            scale = np.abs(gt).max() + 1e-30
            err = np.abs(gg - gt).max() / scale
            worst = max(worst, err) if dt == "float32" else worst
            # fp32: a few ulp of the largest partial sum (cancellation between the two segments' terms); fp64: 1e-9 and better
            assert err <= (tol * 1.2e-7 if dt == "float32" else 1e-9), (case, solver, dt, gg, gt)
    print(f"{solver}: worst fp32 gradient difference {worst:.2e} of the gradient's scale")


def test_adam_update_bit_for_bit():
    """ref.adam_minimize (oracle/ref.py:616-638: optax.scale_by_adam + scale(-lr), bias corrections from double powers cast to
    fp32) fed with a recorded gradient sequence: x, mu, nu after every one of 300 steps, fp32 and fp64, default and other
    hyper-parameters."""
    rng = np.random.default_rng(5)
    for dtype, xp in (("float32", R.NUMPY), ("float64", R.NUMPY64)):
        for hyper in (dict(), dict(lr=0.03, b1=0.8, b2=0.99, eps=1e-6)):
            gs = (rng.standard_normal(300) * np.exp(rng.uniform(-12, 2, 300))).astype(xp.dtype)
            gs[17] = 0.0
            x0 = xp.dtype.type(rng.random())
            state = {"t": 0}

            def vg(x):
                state["t"] += 1
                return xp.c(0.0), [np.asarray(gs[state["t"] - 1])]

            x, mu, nu = float(x0), 0.0, 0.0
            for t in range(1, 301):
                x, mu, nu = CO.opt_adam_step(t, float(gs[t - 1]), x, mu, nu, dtype=dtype, **hyper)
                if t in (1, 2, 17, 18, 100, 300):
                    state["t"] = 0
                    xr, _ = R.adam_minimize(vg, [np.asarray(x0)], steps=t, xp=xp, **{**dict(lr=0.1, b1=0.9, b2=0.999, eps=1e-8), **hyper})
                    assert xp.dtype.type(x) == xr[0], (dtype, hyper, t, x, xr[0])


@pytest.mark.parametrize("solver", ["min", "fermat"])
@pytest.mark.parametrize("approx,function", [(False, "hard_sigmoid"), (True, "hard_sigmoid"), (True, "sigmoid")])
def test_sweeps_against_ref(solver, approx, function):
    kinds, xys, phis = ris_scene()
    objs = ref_objs(kinds, xys, phis)
    tx = np.array([0.2, 0.2], F)
    X, Y = np.meshgrid(np.linspace(0.021, 0.981, 12).astype(F), np.linspace(0.017, 0.977, 10).astype(F))
    cands = R.all_path_candidates(len(kinds), order=1) + R.all_path_candidates(len(kinds), order=2)[:9]
    rng = np.random.default_rng(3)
    th = [rng.random(sum(kinds[int(i)] != 2 for i in c), dtype=F) for c in cands]
    kw = dict(approx=approx, function=function)
    okw = dict(objs=objs, theta0s=th, solver=solver, approx=approx, **({"function": function} if approx else {}))

    def ref_map(steps, xp=R.NUMPY, t=tx, Xg=X, Yg=Y):
        acc = None
        grid = R.vec(xp.asarray(Xg), xp.asarray(Yg), xp)
        o = ref_objs(kinds, xys, phis, xp)
        return R.facc(xp.asarray(t), o, cands, grid, "received_power", None, solver, approx, xp, theta0s=th, steps=steps,
                      **({"function": function} if approx else {}))

    # fp64: the same chain, and the gradient's rounding order hardly matters: 1e-11 (many steps: 1e-7 -- two-reflection solves
    # still in their transient amplify even a last-bit difference)
    for steps in (1, 7, 120):
        want = ref_map(steps, R.NUMPY64)
        got = CO.opt_power_map(kinds, xys, phis, tx, X, Y, cands, th, solver=solver, steps=steps, dtype="float64", **kw)
        assert np.abs(got - want).max() <= (1e-11 if steps < 100 else 1e-7) * np.abs(want).max(), (steps, np.abs(got - want).max())
    # fp32, few steps: a gradient one ulp away moves theta by an ulp
    for steps in (1, 3):
        want = ref_map(steps)
        got = CO.opt_power_map(kinds, xys, phis, tx, X, Y, cands, th, solver=solver, steps=steps, **kw)
        tol = 3e-4 if function == "sigmoid" and approx else 2e-6
        assert np.abs(got - want).max() <= tol * np.abs(want).max(), (steps, np.abs(got - want).max() / np.abs(want).max())
        assert (got == want).mean() > 0.5
    # fp32, many steps: on the cells the ORACLE calls well conditioned (CO.opt_conditioning: every candidate's solver follows
    # the same trajectory in its fp64 run, its fp32 run and its fp32 runs from inputs one ulp away), ref.py's fp32 run sits
    # within twice the oracle's own fp32-vs-fp64 distance (+ 1e-5 of the map's scale)
    # (order-1 candidates, as configs[4]: most two-reflection solves of this scene have not settled after 120 steps, and a
    # solver in its transient is chaotic in every precision)
    steps = 120
    cands, th = cands[:7], th[:7]
    cond = CO.opt_conditioning(kinds, xys, phis, tx, X, Y, cands, th, steps, solver=solver, **kw)
    want = ref_map(steps)
    stable, c64, scale = cond["stable"], cond["value64"], cond["scale"]
    assert stable.mean() > 0.5
    bar = np.maximum(1e-5 * scale + 1e-5 * np.abs(c64), 2.0 * cond["dist"])
    assert (np.abs(want - c64) <= bar)[stable].all(), int((np.abs(want - c64) > bar)[stable].sum())


@pytest.mark.parametrize("solver", ["min", "fermat"])
@pytest.mark.parametrize("approx", [False, True])
@pytest.mark.parametrize("role", ["rx", "tx"])
def test_per_cell_gradient_against_reverse_mode_through_the_loop(solver, approx, role):
    """Second-order forward jets through the Adam loop against torch double-backward autodiff of ref.py (what the reference's
    reverse mode through lax.scan computes), in fp64 where the two modes must agree to rounding; NaN positions of the fp32
    runs equal (the rules stated in the C file's header)."""
    kinds, xys, phis = ris_scene()
    tx = np.array([0.2, 0.2], F)
    X, Y = np.meshgrid(np.linspace(0.05, 0.93, 6).astype(F), np.linspace(0.08, 0.9, 5).astype(F))
    X[0, 0], Y[0, 0] = F(0.5), F(0.1)  # on the RIS's supporting line: its objective does not depend on theta, g == 0 exactly
    cands = R.all_path_candidates(len(kinds), order=1) + R.all_path_candidates(len(kinds), order=2)[:6]
    rng = np.random.default_rng(9)
    th = [rng.random(sum(kinds[int(i)] != 2 for i in c), dtype=F) for c in cands]
    steps = 40
    for dt in ("float64", "float32"):
        want = R.opt_value_and_grads(kinds, np.asarray(xys, np.float64), phis, tx, X, Y, cands, th, solver=solver, steps=steps, dtype=dt,
                                     grid_role=role, approx=approx)
        value, grad = CO.opt_power_map(kinds, xys, phis, tx, X, Y, cands, th, solver=solver, steps=steps, dtype=dt, grad=True,
                                       approx=approx, grid_role=role)
        if dt == "float64":
            assert np.abs(value - want["value"]).max() <= 1e-11 * np.abs(want["value"]).max()
            fin = np.isfinite(want["grad_cell"]) & np.isfinite(grad)
            assert fin.mean() > 0.8
            scale = np.abs(want["grad_cell"][fin]).max()
            assert np.abs(grad - want["grad_cell"])[fin].max() <= 1e-8 * scale, np.abs(grad - want["grad_cell"])[fin].max() / scale
        else:
            # (fp32 only.  On the RIS's supporting line its objective does not depend on theta: where g comes out EXACTLY 0, Adam's
            # sqrt'(0) meets a zero cotangent -- NaN in the reference's reverse mode, a rule here; whether rounding leaves an exact
            # 0 there depends on the precision and on the gradient's evaluation order)
            assert np.array_equal(np.isnan(grad), np.isnan(want["grad_cell"])), (np.isnan(grad).sum(), np.isnan(want["grad_cell"]).sum())
            fin = np.isfinite(grad)
            scale = np.abs(want["grad_cell"][fin]).max()
            assert np.abs(grad - want["grad_cell"])[fin].max() <= 2e-3 * scale
