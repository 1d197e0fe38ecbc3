// Reverse mode through the MinPath / FermatPath solvers (BASELINE.json configs[4]: "grad w.r.t. RIS vertices").
//
// The reference differentiates through the lax.scan of Adam steps (differt2d/optimize.py:83-97) of the objective (MinPath:
// sum of interaction residuals, differt2d/geometry.py:1270-1288, with the RIS residual :698-711; FermatPath: the path
// length, :1117-1204) in REVERSE mode: jax.grad of a scan is a backward scan over the stored trajectory.  This file does
// the same by hand:
//
//   forward   the solver's own loop (opt_run_t<true>: the very code of the forward sweep) writes its trajectory
//             (theta_t, g_t, mu_{t+1}, nu_{t+1} per step, 16 bytes per unknown) to HBM, one (cell, candidate) per lane, lanes
//             of a wave on consecutive addresses;
//   reverse   the adjoint of valid * fun w.r.t. the final points (hand derived, jnp.minimum / maximum ties split as JAX's
//             JVP rules do), then the steps backwards: the adjoint of the Adam update (mu, nu, theta) and, for the
//             gradient g_t = d objective / d theta inside it, the vector-Jacobian products  v^T d g / d (theta, tx, rx,
//             objects)  =  d / d eps  grad_{all inputs} objective(theta + eps v)  (mixed partials commute): ONE forward
//             tangent (Dual<1>, seeded with v = gbar_t in theta) carried through the hand-derived reverse-mode gradient
//             of the objective w.r.t. all its inputs -- no second derivatives written out by hand, and the value part of
//             the same evaluation is the first-order gradient that the recorded loss (MinPath: evaluated at theta_{T-1},
//             geometry.py:1284-1288) needs.
//
// Cost: one reverse step ~ 2.5 forward steps, whatever the number of scene objects -- the forward-tangent kernel this
// replaces (d2d_optgrad.hpp, kept as the independent cross-check: option "opt_grad_mode" = 1) carries 4 + 5 k tangents
// through the loop and 5 more per scene object through the validity chain.
//
// NaN semantics are reverse mode's: a product of local partial derivatives is NaN only where a local partial is
// infinite (sqrt'(0): Adam's sqrt(nu_hat) when the objective does not depend on theta at all -- receivers on the RIS's
// supporting line --, normalize() of a zero-length segment) -- never because an accumulated tangent overflowed on the
// way (a forward-mode artefact: cells next to a wall whose invalid candidate's solver diverges).
#pragma once
#include "d2d_optgrad.hpp"

namespace d2d {

using D1 = Dual<1>;

__device__ __forceinline__ D1 operator*(const D1& a, float s) {
    D1 r;
    r.v = a.v * s;
    r.d[0] = a.d[0] * s;
    return r;
}
__device__ __forceinline__ D1 d1(float v, float d = 0.0f) {
    D1 r;
    r.v = v;
    r.d[0] = d;
    return r;
}

struct OptRevArgs {
    OptGradArgs g;                      // the forward sweep's arguments, cotangent, per-candidate outputs, VJP partial rows
    float* __restrict__ traj;           // trajectory store of this launch
    const long long* __restrict__ traj_off;  // [C] start of candidate c's trajectories, in units of chunk_cells floats
    long cell0;                         // this launch sweeps the cells [cell0, cell0 + chunk_cells); cell0 is a multiple of 64
    long chunk_cells;
    long stride;                        // chunk_cells rounded up to whole waves: every lane of the launch owns a trajectory slot
    long total_blocks;                  // workgroups per candidate over the WHOLE grid (rows of the VJP partial sums)
};

// ---- the solver's loop, optionally recording its trajectory (the forward sweep's opt_run is opt_run_t<false>) ---------
// tr: this lane's slot of the candidate's trajectory; entry (t, q, which) at tr[((t * nu + q) * 4 + which) * stride],
// which = 0 theta_t, 1 g_t, 2 mu_{t+1}, 3 nu_{t+1}
template <bool STORE>
__device__ __forceinline__ float opt_run_t(const ObjTables& T, const AdamCfg& A, int k, const int (&cd)[D2D_MAX_ORDER],
                                           const float* __restrict__ theta0, float txx, float txy, float rxx, float rxy,
                                           float (&th)[D2D_MAX_ORDER], float* __restrict__ tr, long stride) {
    float px[NP], py[NP];
    float mu[D2D_MAX_ORDER], nu[D2D_MAX_ORDER], g[D2D_MAX_ORDER];
    int nu_ = 0;
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) {
        th[i] = theta0[i];
        mu[i] = nu[i] = 0.0f;
        if (i < k && T.kind[cd[i]] != D2D_VERTEX) ++nu_;
    }
    float last = 0.0f;
    for (int t = 0; t < A.steps; ++t) {
        theta_to_points(T, k, cd, th, txx, txy, rxx, rxy, px, py);
        last = objective_grad(T, A.solver, k, cd, px, py, g);
        const float c1 = A.bc1[t], c2 = A.bc2[t];
#pragma unroll
        for (int q = 0; q < D2D_MAX_ORDER; ++q) {
            if (q < nu_) {
                if (STORE) {
                    tr[((long)(t * nu_ + q) * 4 + 0) * stride] = th[q];
                    tr[((long)(t * nu_ + q) * 4 + 1) * stride] = g[q];
                }
                mu[q] = A.b1 * mu[q] + A.omb1 * g[q];
                nu[q] = A.b2 * nu[q] + A.omb2 * (g[q] * g[q]);
                if (STORE) {
                    tr[((long)(t * nu_ + q) * 4 + 2) * stride] = mu[q];
                    tr[((long)(t * nu_ + q) * 4 + 3) * stride] = nu[q];
                }
                float mh = mu[q] / c1, nh = nu[q] / c2;
                th[q] = th[q] + (-A.lr) * (mh / (sqrtf(nh) + A.eps));
            }
        }
    }
    return last;
}

// ---- the objective's gradient w.r.t. ALL its inputs, on Dual<1> --------------------------------------------------------
enum { OBJ_INTERACTION = 0, OBJ_LENGTH = 1 };  // sum of evaluate_cartesian (geometry.py:641-650, 698-711); path_length (:176-203)

template <int K>
struct ObjAdjoint {  // value: d objective / d (.), tangent: its derivative along the seeded theta direction
    static constexpr int KK = K > 0 ? K : 1;
    D1 th[KK];            // unknowns (in order; entries beyond the candidate's number of unknowns stay 0)
    D1 ox[KK], oy[KK];    // object origin, through its interaction point p = o + theta t
    D1 tx[KK], ty[KK];    // t = dest - origin, through the point
    D1 nx[KK], ny[KK];    // the normal
    D1 sphi[KK], cphi[KK];
    D1 ax, ay, bx, by;    // the path's first / last point (transmitter / receiver)
};

// Objects as constants (the float tables the forward solver reads), theta as Dual<1>.
template <int K>
__device__ __forceinline__ void objective_full_grad(const ObjTables& T, int which, const int (&cd)[D2D_MAX_ORDER],
                                                    const D1 (&th)[K > 0 ? K : 1], float txx, float txy, float rxx, float rxy,
                                                    ObjAdjoint<K>& out) {
    constexpr int KK = K > 0 ? K : 1;
    D1 px[K + 2], py[K + 2], pbx[K + 2], pby[K + 2];
    int kind[KK];
    float4 r0[KK], r1[KK];
    px[0] = d1(txx);
    py[0] = d1(txy);
    px[K + 1] = d1(rxx);
    py[K + 1] = d1(rxy);
    int j = 0;
    int slot[KK];  // unknown index of object i (-1: a Vertex)
#pragma unroll
    for (int i = 0; i < K; ++i) {
        kind[i] = T.kind[cd[i]];
        r0[i] = T.refl[2 * cd[i]];
        r1[i] = T.refl[2 * cd[i] + 1];
        slot[i] = -1;
        if (kind[i] == D2D_VERTEX) {
            px[i + 1] = d1(r0[i].x);
            py[i + 1] = d1(r0[i].y);
        } else {
            D1 t = th[0];
#pragma unroll
            for (int q = 1; q < KK; ++q)
                if (q == j) t = th[q];
            px[i + 1] = r0[i].x + t * r1[i].x;  // parametric_to_cartesian, geometry.py:581-587
            py[i + 1] = r0[i].y + t * r1[i].y;
            slot[i] = j;
            ++j;
        }
    }
#pragma unroll
    for (int i = 0; i < K + 2; ++i) pbx[i] = pby[i] = d1(0.0f);
#pragma unroll
    for (int i = 0; i < KK; ++i) {
        out.th[i] = out.ox[i] = out.oy[i] = out.tx[i] = out.ty[i] = d1(0.0f);
        out.nx[i] = out.ny[i] = out.sphi[i] = out.cphi[i] = d1(0.0f);
    }
    if (which == OBJ_LENGTH) {
#pragma unroll
        for (int i = 0; i <= K; ++i) {
            const D1 wx = (px[i + 1] - px[i]) + D2D_EPS, wy = (py[i + 1] - py[i]) + D2D_EPS;
            const D1 len = dsqrt(wx * wx + wy * wy);
            const D1 gx = wx / len, gy = wy / len;
            pbx[i + 1] = pbx[i + 1] + gx;
            pby[i + 1] = pby[i + 1] + gy;
            pbx[i] = pbx[i] - gx;
            pby[i] = pby[i] - gy;
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float nx = r0[i].z, ny = r0[i].w;
            if (kind[i] == D2D_WALL) {
                const D1 v1x = px[i + 1] - px[i], v1y = py[i + 1] - py[i];
                const D1 v2x = px[i + 2] - px[i + 1], v2y = py[i + 2] - py[i + 1];
                D1 ix_, iy_, rx_, ry_;
                dnormalize2(v1x, v1y, ix_, iy_);
                dnormalize2(v2x, v2y, rx_, ry_);
                const D1 din = ix_ * nx + iy_ * ny;
                const D1 s2 = 2.0f * din;
                const D1 ex = rx_ - (ix_ - s2 * nx);
                const D1 ey = ry_ - (iy_ - s2 * ny);
                const D1 ebx = 2.0f * ex, eby = 2.0f * ey;
                const D1 dinb = 2.0f * (ebx * nx + eby * ny);
                const D1 ibx = -ebx + dinb * nx, iby = -eby + dinb * ny;
                // e = r - i + s2 n:  nbar = s2 ebar (+ dinb i through din = i . n)
                out.nx[i] = s2 * ebx + dinb * ix_;
                out.ny[i] = s2 * eby + dinb * iy_;
                D1 a1x, a1y, a2x, a2y;
                dnormalize2_bwd(v1x, v1y, ibx, iby, a1x, a1y);
                dnormalize2_bwd(v2x, v2y, ebx, eby, a2x, a2y);
                pbx[i + 1] = pbx[i + 1] + (a1x - a2x);
                pby[i + 1] = pby[i + 1] + (a1y - a2y);
                pbx[i] = pbx[i] - a1x;
                pby[i] = pby[i] - a1y;
                pbx[i + 2] = pbx[i + 2] + a2x;
                pby[i + 2] = pby[i + 2] + a2y;
            } else if (kind[i] == D2D_RIS) {
                const D1 v2x = px[i + 2] - px[i + 1], v2y = py[i + 2] - py[i + 1];
                D1 rx_, ry_;
                dnormalize2(v2x, v2y, rx_, ry_);
                const D1 mx = -rx_, my = -ry_;
                const D1 sin_a = mx * ny - my * nx;
                const D1 cos_a = mx * nx + my * ny;
                const float2 sc = T.sincos[cd[i]];
                const D1 ds = sin_a - sc.x, dc = cos_a - sc.y;
                const D1 sb = 2.0f * ds, cb = 2.0f * dc;
                out.sphi[i] = -sb;
                out.cphi[i] = -cb;
                out.nx[i] = cb * mx - sb * my;
                out.ny[i] = sb * mx + cb * my;
                const D1 mbx = sb * ny + cb * nx, mby = -(sb * nx) + cb * ny;
                D1 a2x, a2y;
                dnormalize2_bwd(v2x, v2y, -mbx, -mby, a2x, a2y);
                pbx[i + 2] = pbx[i + 2] + a2x;
                pby[i + 2] = pby[i + 2] + a2y;
                pbx[i + 1] = pbx[i + 1] - a2x;
                pby[i + 1] = pby[i + 1] - a2y;
            }
        }
    }
    // points -> end points, objects, unknowns
    out.ax = pbx[0];
    out.ay = pby[0];
    out.bx = pbx[K + 1];
    out.by = pby[K + 1];
#pragma unroll
    for (int i = 0; i < K; ++i) {
        out.ox[i] = pbx[i + 1];
        out.oy[i] = pby[i + 1];
        if (slot[i] >= 0) {
            D1 t = th[0];
#pragma unroll
            for (int q = 1; q < KK; ++q)
                if (q == slot[i]) t = th[q];
            out.tx[i] = t * pbx[i + 1];
            out.ty[i] = t * pby[i + 1];
            const D1 gval = pbx[i + 1] * r1[i].x + pby[i + 1] * r1[i].y;
#pragma unroll
            for (int q = 0; q < KK; ++q)
                if (q == slot[i]) out.th[q] = gval;
        }
    }
}

// ---- jnp.minimum / jnp.maximum as reverse mode sees them: the cotangent goes to the selected argument, ties split evenly ----
__device__ __forceinline__ void tie_min(float a, float b, float& wa, float& wb) {
    wa = (a < b) ? 1.0f : ((a == b) ? 0.5f : 0.0f);
    wb = (b < a) ? 1.0f : ((a == b) ? 0.5f : 0.0f);
}
// weight of the q-th (1-based, in fold order) of r arguments that tie for the extreme of a left fold min(min(min(x0, x1), x2) ...)
__device__ __forceinline__ float fold_weight(int q, int r) {
    const int e = (q == 1) ? (r - 1) : (r - q + 1);
    return __builtin_ldexpf(1.0f, -e);
}

// activation value and derivative w.r.t. x (logic.py:218-267; relu6 = minimum(maximum(y, 0), 6) with its ties)
__device__ __forceinline__ void act_vd(int mode, float alpha, float x, float& v, float& d) {
    const float z = alpha * x;
    if (mode == MODE_HSIG) {
        const float y = z + 3.0f;
        v = minp(maxp(y, 0.0f), 6.0f) / 6.0f;
        const float w0 = (y > 0.0f) ? 1.0f : ((y == 0.0f) ? 0.5f : 0.0f);
        const float w6 = (y < 6.0f) ? 1.0f : ((y == 6.0f) ? 0.5f : 0.0f);
        d = (y != y) ? y : (alpha * w0 * w6) / 6.0f;
    } else {
        v = 1.0f / (1.0f + expf_libm(-z));
        d = alpha * (v * (1.0f - v));
    }
}

// One segment / object test of intersects_with_objects (geometry.py:153-173): its truth value, and -- WANT -- its adjoint
template <bool WANT>
__device__ __forceinline__ float seg_test(int mode, float alpha, float lo, float hi, const float4& w, float qx, float qy, float q1x, float q1y,
                                          float hb, float& g3x, float& g3y, float& g4x, float& g4y, float& p1bx, float& p1by, float& abx,
                                          float& aby) {
    const float Bx = qx - q1x, By = qy - q1y;
    const float Cx = w.x - qx, Cy = w.y - qy;
    const float fa = By * Cx - Bx * Cy, fb = w.z * Cy - w.w * Cx, fd = w.w * Bx - w.z * By;
    const bool dz = (fd == 0.0f);
    const float dd = dz ? 1.0f : fd;
    const float ta = dz ? __builtin_inff() : fa / dd, tb = dz ? __builtin_inff() : fb / dd;
    float v0, v1, v2, v3, d0, d1_, d2, d3;
    act_vd(mode, alpha, ta - lo, v0, d0);
    act_vd(mode, alpha, hi - ta, v1, d1_);
    act_vd(mode, alpha, tb - lo, v2, d2);
    act_vd(mode, alpha, hi - tb, v3, d3);
    const float ma = minp(v0, v1), mb = minp(v2, v3);
    const float h = minp(ma, mb);
    if (WANT) {
        float wa, wb, w0, w1, w2, w3;
        tie_min(ma, mb, wa, wb);
        tie_min(v0, v1, w0, w1);
        tie_min(v2, v3, w2, w3);
        // (where(fd == 0, inf, .) passes no cotangent to the untaken quotient)
        const float tab = dz ? 0.0f : hb * wa * (w0 * d0 - w1 * d1_);
        const float tbb = dz ? 0.0f : hb * wb * (w2 * d2 - w3 * d3);
        const float fab = tab / dd, fbb = tbb / dd;
        const float fdb = dz ? 0.0f : -(tab * ta + tbb * tb) / dd;
        // fa = By Cx - Bx Cy ; fb = Ax Cy - Ay Cx ; fd = Ay Bx - Ax By
        const float Bbx = -fab * Cy + fdb * w.w, Bby = fab * Cx - fdb * w.z;
        const float Cbx = fab * By - fbb * w.w, Cby = -fab * Bx + fbb * w.z;
        abx = fbb * Cy - fdb * By;
        aby = -fbb * Cx + fdb * Bx;
        p1bx = Cbx;
        p1by = Cby;
        g3x = Bbx - Cbx;  // d / d P3
        g3y = Bby - Cby;
        g4x = -Bbx;       // d / d P4
        g4y = -Bby;
    }
    return h;
}

// Everything for one (cell, candidate of order K), reverse mode.
//   grx, gry      d (valid * fun) / d cell
//   row[5 N + 2]  += cot * d / d (object end points [4 N], fixed end point [2], phi [N])   (LDS)
// Returns the contribution itself (the forward sweep's float code on the same final theta: the value map is bit-identical).
// CUST: the instance that serves a host-evaluated path function (D2D_FUN_CUSTOM; its loads and selects cost the fused
// functions' instance 3 % at cfg5 when they share one)
template <int K, bool CUST>
__device__ __forceinline__ float opt_rev_candidate(const OptRevArgs& ra, int c, const int (&cd)[D2D_MAX_ORDER], float cellx, float celly,
                                                   float cot, bool active, long lane_cell, long idx, float& grx, float& gry, float* row) {
    constexpr int KK = K > 0 ? K : 1;
    const OptGradArgs& a = ra.g;
    const OptSweepArgs& s = a.s;
    const ObjTables& T = s.T;
    const int N = T.N;
    const float txx = s.grid_is_tx ? cellx : s.txx, txy = s.grid_is_tx ? celly : s.txy;
    const float rxx = s.grid_is_tx ? s.txx : cellx, rxy = s.grid_is_tx ? s.txy : celly;
    const float* th0 = s.theta0 + (long)c * s.A.many * D2D_MAX_ORDER;
    int kind[KK], slot[KK], nu_ = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        kind[i] = T.kind[cd[i]];
        slot[i] = (kind[i] != D2D_VERTEX) ? nu_++ : -1;
    }
    float* tr = ra.traj + (long)ra.traj_off[c] * ra.stride + lane_cell;
    const long stride = ra.stride;
    // ---- forward: the solver (optimize.py:136-182: the start with the smallest recorded loss wins), trajectory recorded
    float th[D2D_MAX_ORDER] = {0.0f, 0.0f, 0.0f, 0.0f};
    float loss = 0.0f;
    if (K > 0) {
        int best_m = 0;
        if (s.A.many > 1) {
            float tmp[D2D_MAX_ORDER];
            float best_loss = opt_run_t<false>(T, s.A, K, cd, th0, txx, txy, rxx, rxy, tmp, nullptr, 0);
            for (int m = 1; m < s.A.many; ++m) {
                const float l = opt_run_t<false>(T, s.A, K, cd, th0 + m * D2D_MAX_ORDER, txx, txy, rxx, rxy, tmp, nullptr, 0);
                const bool better = (l < best_loss) || (l != l && best_loss == best_loss);
                best_loss = better ? l : best_loss;
                best_m = better ? m : best_m;
            }
        }
        loss = opt_run_t<true>(T, s.A, K, cd, th0 + best_m * D2D_MAX_ORDER, txx, txy, rxx, rxy, th, tr, stride);
    }
    float px[NP], py[NP];
    if (K == 0) image_solve(T, 0, cd, txx, txy, rxx, rxy, px, py);
    else theta_to_points(T, K, cd, th, txx, txy, rxx, rxy, px, py);
    if (K > 0 && s.A.solver == D2D_SOLVER_FERMAT) loss = interaction_loss(T, K, cd, px, py);  // geometry.py:1204
    const Truth L{s.mode, s.alpha};
    float on, hit, valid;
    literal_validity(T, L, K, cd, px, py, loss, s.tol, s.seg_lo, s.seg_hi, on, hit, valid);
    const float r = literal_length(K, px, py);
    float num = s.fnum[0];
#pragma unroll
    for (int q = 1; q <= D2D_MAX_ORDER; ++q)
        if (q == K) num = s.fnum[q];
    float f;
    if (s.fun_id == D2D_FUN_RECEIVED_POWER) f = num / (s.h2 + r * r);
    else if (s.fun_id == D2D_FUN_LENGTH_SQUARED) f = r * r;
    else if (s.fun_id == D2D_FUN_LENGTH) f = r;
    else if (CUST && s.fun_id == D2D_FUN_CUSTOM) f = s.cust_f[(long)c * s.cells + idx];  // d2d_set_path_fun_values: the host's fun on the traced path
    else f = 1.0f;
    const float contribution = valid * f;

    // ---- reverse of valid * f w.r.t. the final points, the recorded loss and the objects -------------------------------
    float pbx[K + 2], pby[K + 2];
#pragma unroll
    for (int i = 0; i < K + 2; ++i) pbx[i] = pby[i] = 0.0f;
    // object adjoints (float: first + second order already combined), per interacting object
    float obx[KK], oby[KK], tbx[KK], tby[KK], nbx[KK], nby[KK], spb[KK], cpb[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) obx[i] = oby[i] = tbx[i] = tby[i] = nbx[i] = nby[i] = spb[i] = cpb[i] = 0.0f;
    const float fbar = valid;
    float rbar;
    if (s.fun_id == D2D_FUN_RECEIVED_POWER) {
        const float Dn = s.h2 + r * r;
        rbar = -(fbar * (f / Dn)) * (2.0f * r);
    } else if (s.fun_id == D2D_FUN_LENGTH_SQUARED) rbar = fbar * (2.0f * r);
    else if (s.fun_id == D2D_FUN_LENGTH) rbar = fbar;
    else rbar = 0.0f;
    if (CUST && s.fun_id == D2D_FUN_CUSTOM) {
        // the host's d fun / d xys (its derivative w.r.t. the end points as arguments of `fun` folded into rows 0 and K + 1)
        const float* pb = s.cust_pb + ((long)c * s.cells + idx) * (2 * (D2D_MAX_ORDER + 2));
#pragma unroll
        for (int i = 0; i < K + 2; ++i) {
            pbx[i] = fbar * pb[2 * i];
            pby[i] = fbar * pb[2 * i + 1];
        }
    }
    // (fun = 1 never evaluates a length: 0 * (w / |w|) would be NaN, not 0, for a segment vector of exactly (-eps, -eps))
    if (s.fun_id == D2D_FUN_RECEIVED_POWER || s.fun_id == D2D_FUN_LENGTH_SQUARED || s.fun_id == D2D_FUN_LENGTH) {
#pragma unroll
        for (int i = 0; i <= K; ++i) {  // path_length, geometry.py:176-203
            const float wx = (px[i + 1] - px[i]) + D2D_EPS, wy = (py[i + 1] - py[i]) + D2D_EPS;
            const float len = sqrtf(wx * wx + wy * wy);
            const float gx = rbar * (wx / len), gy = rbar * (wy / len);
            pbx[i + 1] += gx;
            pby[i + 1] += gy;
            pbx[i] -= gx;
            pby[i] -= gy;
        }
    }
    float lossbar = 0.0f;
    if (s.mode != MODE_HARD) {
        // valid = nan_to_num(min(min(on, 1 - hit), ok)): recompute the three terms with their internals
        float okv, okd;
        act_vd(s.mode, s.alpha, s.tol - loss, okv, okd);
        const float nh = 1.0f - hit;
        const float m1 = minp(on, nh);
        const float vraw = minp(m1, okv);
        const float vbar = (vraw != vraw) ? 0.0f : f;  // nan_to_num: no cotangent where the value was NaN
        float w_m1, w_ok, w_on, w_nh;
        tie_min(m1, okv, w_m1, w_ok);
        tie_min(on, nh, w_on, w_nh);
        const float onbar = vbar * w_m1 * w_on;
        const float hitbar = -(vbar * w_m1 * w_nh);
        lossbar = -(vbar * w_ok * okd);
        // ---- on_objects (geometry.py:821-854): left fold of min from true_value = 1
        if (K > 0 && onbar != 0.0f) {
            float cv[KK], sd[KK];
            int r_on = (on == 1.0f) ? 1 : 0;  // the fold's initial constant ties too
#pragma unroll
            for (int i = 0; i < K; ++i) {
                cv[i] = 1.0f;
                sd[i] = 0.0f;
                if (kind[i] != D2D_VERTEX) {
                    const float4 r0 = T.refl[2 * cd[i]], r1 = T.refl[2 * cd[i] + 1];
                    const float dx = px[i + 1] - r0.x, dy = py[i + 1] - r0.y;
                    const float sp = (r1.x * dx + r1.y * dy) / r1.z;
                    float v0, v1, d0, d1_, w0, w1;
                    act_vd(s.mode, s.alpha, sp - 0.0f, v0, d0);
                    act_vd(s.mode, s.alpha, 1.0f - sp, v1, d1_);
                    tie_min(v0, v1, w0, w1);
                    cv[i] = minp(v0, v1);
                    sd[i] = w0 * d0 - w1 * d1_;  // d cval / d sp
                }
                r_on += (cv[i] == on) ? 1 : 0;
            }
            int q = (on == 1.0f) ? 1 : 0;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (cv[i] == on) {
                    ++q;
                    if (kind[i] != D2D_VERTEX) {
                        const float4 r0 = T.refl[2 * cd[i]], r1 = T.refl[2 * cd[i] + 1];
                        const float dx = px[i + 1] - r0.x, dy = py[i + 1] - r0.y;
                        const float sp = (r1.x * dx + r1.y * dy) / r1.z;
                        const float sb = onbar * fold_weight(q, r_on) * sd[i];
                        const float qq = sb / r1.z;
                        pbx[i + 1] += qq * r1.x;
                        pby[i + 1] += qq * r1.y;
                        obx[i] -= qq * r1.x;
                        oby[i] -= qq * r1.y;
                        // t enters the numerator and sq = t . t (a constant 1 when the object is degenerate)
                        const bool degenerate = (r1.x * r1.x + r1.y * r1.y == 0.0f);
                        const float sqb = degenerate ? 0.0f : -(qq * sp);
                        tbx[i] += qq * dx + 2.0f * sqb * r1.x;
                        tby[i] += qq * dy + 2.0f * sqb * r1.y;
                    }
                }
            }
        }
        // ---- intersects_with_objects (geometry.py:856-906): left fold of max from false_value = 0 over (segment, object)
        if (hitbar != 0.0f && N > 0) {
            float dummy;
            int r_hit = (hit == 0.0f) ? 1 : 0;
#pragma unroll
            for (int i = 0; i <= K; ++i) {
                const int ig0 = (i == 0) ? -1 : cd[i - 1];
                const int ig1 = (i == K) ? -1 : cd[i < D2D_MAX_ORDER ? i : 0];
                for (int j = 0; j < N; ++j) {
                    if (j == ig0 || j == ig1 || T.kind[j] == D2D_VERTEX) continue;
                    const float h = seg_test<false>(s.mode, s.alpha, s.seg_lo, s.seg_hi, T.occl[j], px[i], py[i], px[i + 1], py[i + 1], 0.0f, dummy,
                                                    dummy, dummy, dummy, dummy, dummy, dummy, dummy);
                    r_hit += (h == hit) ? 1 : 0;
                }
            }
            int q = (hit == 0.0f) ? 1 : 0;
#pragma unroll
            for (int i = 0; i <= K; ++i) {
                const int ig0 = (i == 0) ? -1 : cd[i - 1];
                const int ig1 = (i == K) ? -1 : cd[i < D2D_MAX_ORDER ? i : 0];
                for (int j = 0; j < N; ++j) {
                    if (j == ig0 || j == ig1 || T.kind[j] == D2D_VERTEX) continue;
                    const float4 w = T.occl[j];
                    float g3x, g3y, g4x, g4y, p1bx, p1by, abx, aby;
                    const float h = seg_test<false>(s.mode, s.alpha, s.seg_lo, s.seg_hi, w, px[i], py[i], px[i + 1], py[i + 1], 0.0f, g3x, g3y, g4x,
                                                    g4y, p1bx, p1by, abx, aby);
                    if (!(h == hit)) continue;
                    ++q;
                    const float hb = hitbar * fold_weight(q, r_hit);
                    seg_test<true>(s.mode, s.alpha, s.seg_lo, s.seg_hi, w, px[i], py[i], px[i + 1], py[i + 1], hb, g3x, g3y, g4x, g4y, p1bx, p1by,
                                   abx, aby);
                    pbx[i] += g3x;
                    pby[i] += g3y;
                    pbx[i + 1] += g4x;
                    pby[i + 1] += g4y;
                    if (row && active) {
                        // P1 = (1 + patch) o - patch d ; P2 = (1 + patch) d - patch o ; A = P2 - P1   (geometry.py:632-636)
                        const float P2bx = abx, P2by = aby;
                        const float P1bx = p1bx - abx, P1by = p1by - aby;
                        const float pa = a.patch;
                        float* w4 = row + 4 * j;
                        atomicAdd(&w4[0], cot * ((1.0f + pa) * P1bx - pa * P2bx));
                        atomicAdd(&w4[1], cot * ((1.0f + pa) * P1by - pa * P2by));
                        atomicAdd(&w4[2], cot * ((1.0f + pa) * P2bx - pa * P1bx));
                        atomicAdd(&w4[3], cot * ((1.0f + pa) * P2by - pa * P1by));
                    }
                }
            }
        }
    }

    // ---- end points, objects and unknowns: what the contribution says directly, then the solver backwards ---------------
    float thb[KK], mub[KK], nub[KK];
#pragma unroll
    for (int i = 0; i < KK; ++i) thb[i] = mub[i] = nub[i] = 0.0f;
    float axb = pbx[0], ayb = pby[0], bxb = pbx[K + 1], byb = pby[K + 1];
#pragma unroll
    for (int i = 0; i < K; ++i) {  // p = o + theta t (a Vertex: p = o)
        obx[i] += pbx[i + 1];
        oby[i] += pby[i + 1];
        if (slot[i] >= 0) {
            const float4 r1 = T.refl[2 * cd[i] + 1];
            float t = th[0];
#pragma unroll
            for (int q = 1; q < KK; ++q)
                if (q == slot[i]) t = th[q];
            tbx[i] += t * pbx[i + 1];
            tby[i] += t * pby[i + 1];
            const float gval = r1.x * pbx[i + 1] + r1.y * pby[i + 1];
#pragma unroll
            for (int q = 0; q < KK; ++q)
                if (q == slot[i]) thb[q] = gval;
        }
    }
    auto accumulate = [&](const ObjAdjoint<K>& G, float lam, bool second) {
        // lam * (first-order gradient)  +  (second ? the derivative along the seeded direction : 0)
        auto take = [&](const D1& x) -> float { return second ? (lam != 0.0f ? lam * x.v + x.d[0] : x.d[0]) : lam * x.v; };
        axb += take(G.ax);
        ayb += take(G.ay);
        bxb += take(G.bx);
        byb += take(G.by);
#pragma unroll
        for (int i = 0; i < K; ++i) {
            obx[i] += take(G.ox[i]);
            oby[i] += take(G.oy[i]);
            tbx[i] += take(G.tx[i]);
            tby[i] += take(G.ty[i]);
            nbx[i] += take(G.nx[i]);
            nby[i] += take(G.ny[i]);
            spb[i] += take(G.sphi[i]);
            cpb[i] += take(G.cphi[i]);
        }
#pragma unroll
        for (int q = 0; q < KK; ++q) thb[q] += take(G.th[q]);
    };
    if (K > 0 && nu_ > 0) {
        ObjAdjoint<K> G;
        D1 thd[KK];
        if (s.A.solver == D2D_SOLVER_FERMAT) {
            // the loss attached to a FermatPath is the interaction loss of its final points (geometry.py:1204)
            if (lossbar != 0.0f) {
#pragma unroll
                for (int q = 0; q < KK; ++q) thd[q] = d1(th[q]);
                objective_full_grad<K>(T, OBJ_INTERACTION, cd, thd, txx, txy, rxx, rxy, G);
                accumulate(G, lossbar, false);
            }
        }
        const int which = (s.A.solver == D2D_SOLVER_FERMAT) ? OBJ_LENGTH : OBJ_INTERACTION;
        for (int t = s.A.steps - 1; t >= 0; --t) {
            const float c1 = s.A.bc1[t], c2 = s.A.bc2[t];
            float gbar[KK], tht[KK];
#pragma unroll
            for (int q = 0; q < KK; ++q) {
                gbar[q] = tht[q] = 0.0f;
                if (q < nu_) {
                    const float* e = tr + ((long)(t * nu_ + q) * 4) * stride;
                    tht[q] = e[0];
                    const float gt = e[stride], mu1 = e[2 * stride], nu1 = e[3 * stride];
                    const float mh = mu1 / c1, nh = nu1 / c2;
                    const float sq = sqrtf(nh);
                    const float den = sq + s.A.eps;
                    // theta' = theta + (-lr) * (mh / den)
                    const float ub = (-s.A.lr) * thb[q];
                    const float mhb = ub / den;
                    const float denb = -(ub * mh) / (den * den);
                    const float nhb = denb * (0.5f / sq);  // sqrt'(0) = inf: 0 * inf = NaN, as jnp.sqrt's rule
                    const float m1b = mub[q] + mhb / c1;
                    const float n1b = nub[q] + nhb / c2;
                    gbar[q] = s.A.omb1 * m1b + s.A.omb2 * ((2.0f * gt) * n1b);
                    mub[q] = s.A.b1 * m1b;
                    nub[q] = s.A.b2 * n1b;
                }
            }
            D1 seed[KK];
#pragma unroll
            for (int q = 0; q < KK; ++q) seed[q] = d1(tht[q], gbar[q]);
            objective_full_grad<K>(T, which, cd, seed, txx, txy, rxx, rxy, G);
            // MinPath records the objective at theta_{T-1} as the path's loss (geometry.py:1284-1288)
            const float lam = (t == s.A.steps - 1 && s.A.solver != D2D_SOLVER_FERMAT) ? lossbar : 0.0f;
            accumulate(G, lam, true);
        }
    }

    // ---- per-cell gradient and the scene VJP ---------------------------------------------------------------------------
    const float cbx = s.grid_is_tx ? axb : bxb, cby = s.grid_is_tx ? ayb : byb;  // the lane's own end point
    const float fbx = s.grid_is_tx ? bxb : axb, fby = s.grid_is_tx ? byb : ayb;  // the launch's fixed end point
    grx = active ? cbx : 0.0f;
    gry = active ? cby : 0.0f;
    if (CUST && s.fun_id == D2D_FUN_CUSTOM && active && !(fabsf(f) < 3.0e38f)) {
        // a host function that is not finite here: f * d valid holds a 0 * inf (or a NaN) whether the candidate is valid or not
        grx = gry = __builtin_nanf("");
    }
    if (row) {
        const float wgt = active ? cot : 0.0f;
        const float sfx = wave_sum(wgt * fbx), sfy = wave_sum(wgt * fby);
        if ((threadIdx.x & 63) == 0) {
            row[4 * N] += sfx;
            row[4 * N + 1] += sfy;
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float4 r0 = T.refl[2 * cd[i]], r1 = T.refl[2 * cd[i] + 1];
            float ob_x = obx[i], ob_y = oby[i], db_x = 0.0f, db_y = 0.0f, phib = 0.0f;
            if (kind[i] != D2D_VERTEX) {
                // n = m / |m|, m = (t_y, -t_x) (geometry.py:561-573); t = dest - origin
                const float len = r1.w;  // |t| guarded to 1
                const bool z = (r1.x * r1.x + r1.y * r1.y == 0.0f);
                const float dd = z ? 0.0f : (nbx[i] * r0.z + nby[i] * r0.w);
                const float mbx = (nbx[i] - dd * r0.z) / len, mby = (nby[i] - dd * r0.w) / len;
                const float ttx = tbx[i] - mby, tty = tby[i] + mbx;  // m_x = t_y, m_y = -t_x
                db_x = ttx;
                db_y = tty;
                ob_x -= ttx;
                ob_y -= tty;
                const float2 sc = T.sincos[cd[i]];
                phib = spb[i] * sc.y - cpb[i] * sc.x;  // d sin = cos, d cos = -sin (geometry.py:709-710)
            }
            const float s0 = wave_sum(wgt * ob_x), s1 = wave_sum(wgt * ob_y), s2 = wave_sum(wgt * db_x), s3 = wave_sum(wgt * db_y);
            const float s4 = wave_sum(wgt * phib);
            if ((threadIdx.x & 63) == 0) {
                float* w4 = row + 4 * cd[i];
                atomicAdd(&w4[0], s0);
                atomicAdd(&w4[1], s1);
                atomicAdd(&w4[2], s2);
                atomicAdd(&w4[3], s3);
                atomicAdd(&row[4 * N + 2 + cd[i]], s4);
            }
        }
    }
    return contribution;
}

#ifdef D2D_OPTREV_KERNELS  // defined once, in d2d_optrev.hip
// One (cell, candidate) per lane; candidate = c_first + blockIdx.y (wave-uniform, all of order K: the enumeration is by ascending
// order, so every order is a contiguous range and gets a launch -- and a register allocation -- of its own), cells
// [cell0, cell0 + chunk_cells).
template <int K, bool CUST>
__global__ void __launch_bounds__(64) power_opt_rev_kernel(OptRevArgs ra, int c_first) {
    extern __shared__ float row[];  // [5 N + 2]
    const OptGradArgs& a = ra.g;
    const OptSweepArgs& s = a.s;
    const int lane = threadIdx.x & 63;
    const int n_elem = 5 * s.T.N + 2;
    for (int i = lane; i < n_elem; i += 64) row[i] = 0.0f;
    __syncthreads();
    const long lc0 = (long)blockIdx.x * 64 + lane;  // cell within the chunk
    const long idx0 = ra.cell0 + lc0;
    const bool active = lc0 < ra.chunk_cells && idx0 < s.cells;
    const long idx = active ? idx0 : (s.cells - 1);
    const long lane_cell = lc0;  // (< stride: inactive lanes of the last wave have a slot of their own)
    const int c = c_first + blockIdx.y;
    const float cellx = s.X[idx], celly = s.Y[idx];
    const float cot = active ? (a.cot ? a.cot[idx] : 1.0f) : 0.0f;
    int cd[D2D_MAX_ORDER];
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) cd[i] = s.cand[c * D2D_MAX_ORDER + i];
    float grx = 0.0f, gry = 0.0f;
    float* r = a.partial ? row : nullptr;
    const float v = opt_rev_candidate<K, CUST>(ra, c, cd, cellx, celly, cot, active, lane_cell, idx, grx, gry, r);
    if (active) {
        a.contrib[(long)c * s.cells + idx] = v;
        a.gcontrib[((long)c * s.cells + idx) * 2] = grx;
        a.gcontrib[((long)c * s.cells + idx) * 2 + 1] = gry;
    }
    if (a.partial) {
        __syncthreads();
        float* dst = a.partial + ((long)c * ra.total_blocks + (ra.cell0 / 64 + blockIdx.x)) * n_elem;
        for (int i = lane; i < n_elem; i += 64) dst[i] = row[i];
    }
}
#endif  // D2D_OPTREV_KERNELS

// candidates [c_first, c_first + grid.y) must all be of order K
hipError_t launch_opt_rev(int K, const OptRevArgs& a, int c_first, dim3 grid, size_t lds, hipStream_t stream);

}  // namespace d2d
