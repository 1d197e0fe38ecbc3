// Gradients through the MinPath / FermatPath solvers (BASELINE.json configs[4]: "grad w.r.t. RIS vertices").
//
// The reference differentiates through the lax.scan of Adam steps (differt2d/optimize.py:83-97) of the objective
// (MinPath: sum of interaction residuals, differt2d/geometry.py:1270-1288, with the RIS residual :698-711; FermatPath:
// the path length, :1117-1204), then through parametric_to_cartesian (:988-1010), Path.is_valid (:908-963) and `fun`.
// Here the same derivative is carried FORWARD: every quantity of the solver -- theta, Adam's two moments, the
// hand-derived gradient of the objective (objective_grad, the same expression the forward solver uses) -- carries the
// tangents of B input directions next to its value (struct Dual), so one pass of the `steps` iterations yields
// d theta_T / d(direction) exactly as reverse mode through the scan would (same function, same derivative; no
// trajectory to store, no second pass), and the validity / `fun` chain evaluated on those Duals yields the directional
// derivatives of the contribution valid * fun.  Directions: the cell (2), the fixed end point (2), and per object its
// two end points (4) and its phi (1); a candidate's solver only depends on its own objects, so the Adam loop runs with
// the 4 + 5 k directions that move theta (B at a time), the remaining objects only enter through occlusion.
//
// Semantics mirrored from JAX: where(c, a, b) passes the tangent of the taken branch; minimum / maximum pass the
// tangent of the selected argument and split ties evenly (lax.min / lax.max JVP); lax.logistic's rule s (1 - s);
// sqrt'(0) = inf (so normalize() of a zero-length segment poisons the gradient with NaN, as in the reference);
// nan_to_num passes the tangent where the value is finite.
//
// Values: the contribution itself is computed by the forward solver's own float code (opt_contribution), so the value
// map of a value+grad sweep is bit-identical to the forward sweep's.
#pragma once
#include "d2d_kernels.hpp"

namespace d2d {

template <int B>
struct Dual {
    float v;
    float d[B];
};

#define D2D_DUAL_FN template <int B> __device__ __forceinline__

D2D_DUAL_FN Dual<B> dconst(float v) {
    Dual<B> r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = 0.0f;
    return r;
}
// value v, tangent 1 in the slots of this batch that stand for global direction g
D2D_DUAL_FN Dual<B> dseed(float v, int g, const int (&gdir)[B]) {
    Dual<B> r;
    r.v = v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = (gdir[i] == g) ? 1.0f : 0.0f;
    return r;
}
D2D_DUAL_FN Dual<B> operator+(const Dual<B>& a, const Dual<B>& b) {
    Dual<B> r;
    r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = a.d[i] + b.d[i];
    return r;
}
D2D_DUAL_FN Dual<B> operator-(const Dual<B>& a, const Dual<B>& b) {
    Dual<B> r;
    r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = a.d[i] - b.d[i];
    return r;
}
D2D_DUAL_FN Dual<B> operator-(const Dual<B>& a) {
    Dual<B> r;
    r.v = -a.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = -a.d[i];
    return r;
}
D2D_DUAL_FN Dual<B> operator*(const Dual<B>& a, const Dual<B>& b) {
    Dual<B> r;
    r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
    return r;
}
D2D_DUAL_FN Dual<B> operator/(const Dual<B>& a, const Dual<B>& b) {
    Dual<B> r;
    r.v = a.v / b.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v;
    return r;
}
D2D_DUAL_FN Dual<B> operator+(const Dual<B>& a, float s) {
    Dual<B> r = a;
    r.v = a.v + s;
    return r;
}
D2D_DUAL_FN Dual<B> operator+(float s, const Dual<B>& a) {
    Dual<B> r = a;
    r.v = s + a.v;
    return r;
}
D2D_DUAL_FN Dual<B> operator-(const Dual<B>& a, float s) {
    Dual<B> r = a;
    r.v = a.v - s;
    return r;
}
D2D_DUAL_FN Dual<B> operator-(float s, const Dual<B>& a) {
    Dual<B> r;
    r.v = s - a.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = -a.d[i];
    return r;
}
D2D_DUAL_FN Dual<B> operator*(float s, const Dual<B>& a) {
    Dual<B> r;
    r.v = s * a.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = s * a.d[i];
    return r;
}
D2D_DUAL_FN Dual<B> operator/(const Dual<B>& a, float s) {
    Dual<B> r;
    r.v = a.v / s;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = a.d[i] / s;
    return r;
}
D2D_DUAL_FN Dual<B> operator/(float s, const Dual<B>& b) {
    Dual<B> r;
    r.v = s / b.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = -(r.v * b.d[i]) / b.v;
    return r;
}
D2D_DUAL_FN Dual<B> dsqrt(const Dual<B>& a) {
    Dual<B> r;
    r.v = sqrtf(a.v);
    const float h = 0.5f / r.v;  // inf at 0: 0 * inf = NaN, as jnp.sqrt's JVP
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = a.d[i] * h;
    return r;
}
// jnp.where(c, a, b)
D2D_DUAL_FN Dual<B> dwhere(bool c, const Dual<B>& a, const Dual<B>& b) { return c ? a : b; }
// jnp.minimum / jnp.maximum: NaN-propagating values; tangent of the selected argument, ties split evenly
D2D_DUAL_FN Dual<B> dmin(const Dual<B>& a, const Dual<B>& b) {
    if (a.v != a.v || b.v != b.v) {
        Dual<B> r = a + b;
        r.v = __builtin_nanf("");
        return r;
    }
    if (a.v < b.v) return a;
    if (b.v < a.v) return b;
    Dual<B> r;
    r.v = a.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = 0.5f * (a.d[i] + b.d[i]);
    return r;
}
D2D_DUAL_FN Dual<B> dmax(const Dual<B>& a, const Dual<B>& b) {
    if (a.v != a.v || b.v != b.v) {
        Dual<B> r = a + b;
        r.v = __builtin_nanf("");
        return r;
    }
    if (a.v > b.v) return a;
    if (b.v > a.v) return b;
    Dual<B> r;
    r.v = a.v;
#pragma unroll
    for (int i = 0; i < B; ++i) r.d[i] = 0.5f * (a.d[i] + b.d[i]);
    return r;
}

// logic.py:218-267 on Duals (approx modes only; hard mode has no gradient through validity)
template <int B>
struct DTruth {
    int mode;
    float alpha;
    __device__ __forceinline__ Dual<B> act(const Dual<B>& x) const {
        Dual<B> z = alpha * x;
        if (mode == MODE_HSIG) return dmin(dmax(z + 3.0f, dconst<B>(0.0f)), dconst<B>(6.0f)) / 6.0f;
        Dual<B> r;  // lax.logistic: value 1 / (1 + exp(-z)), JVP s (1 - s)
        r.v = 1.0f / (1.0f + expf_libm(-z.v));
        const float s1 = r.v * (1.0f - r.v);
#pragma unroll
        for (int i = 0; i < B; ++i) r.d[i] = s1 * z.d[i];
        return r;
    }
    __device__ __forceinline__ Dual<B> ge(const Dual<B>& x, float y) const { return act(x - y); }
    __device__ __forceinline__ Dual<B> le(const Dual<B>& x, float y) const { return act(y - x); }
};

template <int B>
struct DObj {  // an object with its end points (and phi) as differentiable inputs
    int kind;
    Dual<B> ox, oy;          // origin (a Vertex: its point)
    Dual<B> tx, ty, sq;      // t = dest - origin, t.t guarded (geometry.py:479-487, 596-597)
    Dual<B> nx, ny;          // normal (geometry.py:561-573)
    Dual<B> p1x, p1y, ax, ay;  // patched origin and P2 - P1 (geometry.py:632-636)
    Dual<B> sphi, cphi;      // RIS (geometry.py:709-710)
};

// Global direction ids: 0, 1 the grid cell; 2, 3 the fixed end point; 4 + 5 j + {0, 1, 2, 3} object j's origin.xy, dest.xy;
// 4 + 5 j + 4 its phi.
__host__ __device__ constexpr int dir_obj(int j, int c) { return 4 + 5 * j + c; }

// Same operations, same order as the host builds the float tables (d2d.hip: upload_refl, upload_occl)
template <int B>
__device__ __forceinline__ DObj<B> make_dobj(const ObjTables& T, int j, float patch, const int (&gdir)[B]) {
    DObj<B> o;
    o.kind = T.kind[j];
    const float4 e = T.xys[j];
    o.ox = dseed<B>(e.x, dir_obj(j, 0), gdir);
    o.oy = dseed<B>(e.y, dir_obj(j, 1), gdir);
    const Dual<B> dx = dseed<B>(e.z, dir_obj(j, 2), gdir), dy = dseed<B>(e.w, dir_obj(j, 3), gdir);
    o.tx = dx - o.ox;
    o.ty = dy - o.oy;
    const Dual<B> vx = o.ty, vy = -o.tx;
    Dual<B> len = dsqrt(vx * vx + vy * vy);
    len = dwhere(len.v == 0.0f, dconst<B>(1.0f), len);
    o.nx = vx / len;
    o.ny = vy / len;
    Dual<B> sq = o.tx * o.tx + o.ty * o.ty;
    o.sq = dwhere(sq.v == 0.0f, dconst<B>(1.0f), sq);
    const Dual<B> ptx = patch * o.tx, pty = patch * o.ty;
    o.p1x = o.ox - ptx;
    o.p1y = o.oy - pty;
    const Dual<B> p2x = dx + ptx, p2y = dy + pty;
    o.ax = p2x - o.p1x;
    o.ay = p2y - o.p1y;
    const float2 sc = T.sincos[j];
    const Dual<B> ph = dseed<B>(0.0f, dir_obj(j, 4), gdir);  // only its tangent is used
    o.sphi = dconst<B>(sc.x);
    o.cphi = dconst<B>(sc.y);
#pragma unroll
    for (int i = 0; i < B; ++i) {
        o.sphi.d[i] = sc.y * ph.d[i];
        o.cphi.d[i] = -sc.x * ph.d[i];
    }
    return o;
}

template <int B>
__device__ __forceinline__ void dnormalize2(const Dual<B>& vx, const Dual<B>& vy, Dual<B>& ox, Dual<B>& oy) {
    Dual<B> len = dsqrt(vx * vx + vy * vy);
    len = dwhere(len.v == 0.0f, dconst<B>(1.0f), len);
    ox = vx / len;
    oy = vy / len;
}
// the Dual of normalize2_bwd (d2d_kernels.hpp): vbar = (obar - (obar . o) o) / len
template <int B>
__device__ __forceinline__ void dnormalize2_bwd(const Dual<B>& vx, const Dual<B>& vy, const Dual<B>& obx, const Dual<B>& oby,
                                                 Dual<B>& vbx, Dual<B>& vby) {
    Dual<B> len = dsqrt(vx * vx + vy * vy);
    const bool z = (len.v == 0.0f);
    len = dwhere(z, dconst<B>(1.0f), len);
    const Dual<B> ox = vx / len, oy = vy / len;
    const Dual<B> d = dwhere(z, dconst<B>(0.0f), obx * ox + oby * oy);
    vbx = (obx - d * ox) / len;
    vby = (oby - d * oy) / len;
}

// parametric_to_cartesian (geometry.py:988-1010) for K interacting objects; th[] holds the unknowns in order
template <int K, int B>
__device__ __forceinline__ void dpoints(const DObj<B> (&ob)[K > 0 ? K : 1], const Dual<B> (&th)[K > 0 ? K : 1], const Dual<B>& ax_,
                                        const Dual<B>& ay_, const Dual<B>& bx_, const Dual<B>& by_, Dual<B> (&px)[K + 2],
                                        Dual<B> (&py)[K + 2]) {
    px[0] = ax_;
    py[0] = ay_;
    int j = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (ob[i].kind == D2D_VERTEX) {
            px[i + 1] = ob[i].ox;
            py[i + 1] = ob[i].oy;
        } else {
            Dual<B> t = th[0];
#pragma unroll
            for (int q = 1; q < K; ++q)
                if (q == j) t = th[q];
            px[i + 1] = ob[i].ox + t * ob[i].tx;
            py[i + 1] = ob[i].oy + t * ob[i].ty;
            ++j;
        }
    }
    px[K + 1] = bx_;
    py[K + 1] = by_;
}

// sum_k obj_k.evaluate_cartesian(P[k:k+3]) (Wall geometry.py:641-650, RIS :698-711, Vertex :416-419)
template <int K, int B>
__device__ __forceinline__ Dual<B> dinteraction_loss(const DObj<B> (&ob)[K > 0 ? K : 1], const Dual<B> (&px)[K + 2],
                                                     const Dual<B> (&py)[K + 2]) {
    Dual<B> loss = dconst<B>(0.0f);
#pragma unroll
    for (int i = 0; i < K; ++i) {
        Dual<B> ev = dconst<B>(0.0f);
        if (ob[i].kind == D2D_WALL) {
            Dual<B> ix_, iy_, rx_, ry_;
            dnormalize2(px[i + 1] - px[i], py[i + 1] - py[i], ix_, iy_);
            dnormalize2(px[i + 2] - px[i + 1], py[i + 2] - py[i + 1], rx_, ry_);
            const Dual<B> din = ix_ * ob[i].nx + iy_ * ob[i].ny;
            const Dual<B> s2 = 2.0f * din;
            const Dual<B> ex = rx_ - (ix_ - s2 * ob[i].nx);
            const Dual<B> ey = ry_ - (iy_ - s2 * ob[i].ny);
            ev = ex * ex + ey * ey;
        } else if (ob[i].kind == D2D_RIS) {
            Dual<B> rx_, ry_;
            dnormalize2(px[i + 2] - px[i + 1], py[i + 2] - py[i + 1], rx_, ry_);
            const Dual<B> mx = -rx_, my = -ry_;
            const Dual<B> sin_a = mx * ob[i].ny - my * ob[i].nx;
            const Dual<B> cos_a = mx * ob[i].nx + my * ob[i].ny;
            const Dual<B> ds = sin_a - ob[i].sphi, dc = cos_a - ob[i].cphi;
            ev = ds * ds + dc * dc;
        }
        loss = loss + ev;
    }
    return loss;
}

// path_length (geometry.py:176-203)
template <int K, int B>
__device__ __forceinline__ Dual<B> dlength(const Dual<B> (&px)[K + 2], const Dual<B> (&py)[K + 2]) {
    Dual<B> r = dconst<B>(0.0f);
#pragma unroll
    for (int i = 0; i <= K; ++i) {
        const Dual<B> vx = (px[i + 1] - px[i]) + D2D_EPS;
        const Dual<B> vy = (py[i + 1] - py[i]) + D2D_EPS;
        r = r + dsqrt(vx * vx + vy * vy);
    }
    return r;
}

// The Dual of objective_grad (d2d_kernels.hpp): the objective and its hand-derived gradient w.r.t. theta, both with tangents.
template <int K, int B>
__device__ __forceinline__ Dual<B> dobjective_grad(int solver, const DObj<B> (&ob)[K > 0 ? K : 1], const Dual<B> (&px)[K + 2],
                                                   const Dual<B> (&py)[K + 2], Dual<B> (&gth)[K > 0 ? K : 1]) {
    Dual<B> pbx[K + 2], pby[K + 2];
#pragma unroll
    for (int i = 0; i < K + 2; ++i) pbx[i] = pby[i] = dconst<B>(0.0f);
    Dual<B> loss = dconst<B>(0.0f);
    if (solver == D2D_SOLVER_FERMAT) {
#pragma unroll
        for (int i = 0; i <= K; ++i) {
            const Dual<B> wx = (px[i + 1] - px[i]) + D2D_EPS, wy = (py[i + 1] - py[i]) + D2D_EPS;
            const Dual<B> len = dsqrt(wx * wx + wy * wy);
            loss = loss + len;
            const Dual<B> gx = wx / len, gy = wy / len;
            pbx[i + 1] = pbx[i + 1] + gx;
            pby[i + 1] = pby[i + 1] + gy;
            pbx[i] = pbx[i] - gx;
            pby[i] = pby[i] - gy;
        }
    } else {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            Dual<B> ev = dconst<B>(0.0f);
            if (ob[i].kind == D2D_WALL) {
                const Dual<B> v1x = px[i + 1] - px[i], v1y = py[i + 1] - py[i];
                const Dual<B> v2x = px[i + 2] - px[i + 1], v2y = py[i + 2] - py[i + 1];
                Dual<B> ix_, iy_, rx_, ry_;
                dnormalize2(v1x, v1y, ix_, iy_);
                dnormalize2(v2x, v2y, rx_, ry_);
                const Dual<B> din = ix_ * ob[i].nx + iy_ * ob[i].ny;
                const Dual<B> s2 = 2.0f * din;
                const Dual<B> ex = rx_ - (ix_ - s2 * ob[i].nx);
                const Dual<B> ey = ry_ - (iy_ - s2 * ob[i].ny);
                ev = ex * ex + ey * ey;
                const Dual<B> ebx = 2.0f * ex, eby = 2.0f * ey;
                const Dual<B> dinb = 2.0f * (ebx * ob[i].nx + eby * ob[i].ny);
                const Dual<B> ibx = -ebx + dinb * ob[i].nx, iby = -eby + dinb * ob[i].ny;
                Dual<B> a1x, a1y, a2x, a2y;
                dnormalize2_bwd(v1x, v1y, ibx, iby, a1x, a1y);
                dnormalize2_bwd(v2x, v2y, ebx, eby, a2x, a2y);
                pbx[i + 1] = pbx[i + 1] + (a1x - a2x);
                pby[i + 1] = pby[i + 1] + (a1y - a2y);
                pbx[i] = pbx[i] - a1x;
                pby[i] = pby[i] - a1y;
                pbx[i + 2] = pbx[i + 2] + a2x;
                pby[i + 2] = pby[i + 2] + a2y;
            } else if (ob[i].kind == D2D_RIS) {
                const Dual<B> v2x = px[i + 2] - px[i + 1], v2y = py[i + 2] - py[i + 1];
                Dual<B> rx_, ry_;
                dnormalize2(v2x, v2y, rx_, ry_);
                const Dual<B> mx = -rx_, my = -ry_;
                const Dual<B> sin_a = mx * ob[i].ny - my * ob[i].nx;
                const Dual<B> cos_a = mx * ob[i].nx + my * ob[i].ny;
                const Dual<B> ds = sin_a - ob[i].sphi, dc = cos_a - ob[i].cphi;
                ev = ds * ds + dc * dc;
                const Dual<B> sb = 2.0f * ds, cb = 2.0f * dc;
                const Dual<B> mbx = sb * ob[i].ny + cb * ob[i].nx, mby = -(sb * ob[i].nx) + cb * ob[i].ny;
                Dual<B> a2x, a2y;
                dnormalize2_bwd(v2x, v2y, -mbx, -mby, a2x, a2y);
                pbx[i + 2] = pbx[i + 2] + a2x;
                pby[i + 2] = pby[i + 2] + a2y;
                pbx[i + 1] = pbx[i + 1] - a2x;
                pby[i + 1] = pby[i + 1] - a2y;
            }
            loss = loss + ev;
        }
    }
    // d / d theta_j = t_j . pbar_j
    int j = 0;
#pragma unroll
    for (int q = 0; q < (K > 0 ? K : 1); ++q) gth[q] = dconst<B>(0.0f);
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if (ob[i].kind != D2D_VERTEX) {
            const Dual<B> gval = ob[i].tx * pbx[i + 1] + ob[i].ty * pby[i + 1];
#pragma unroll
            for (int q = 0; q < K; ++q)
                if (q == j) gth[q] = gval;
            ++j;
        }
    }
    return loss;
}

// One Adam run (the Dual of opt_run): final theta in th[], returns the objective recorded at the last step.
template <int K, int B>
__device__ __forceinline__ Dual<B> dopt_run(const AdamCfg& A, const DObj<B> (&ob)[K > 0 ? K : 1], const float* __restrict__ theta0,
                                            const Dual<B>& ax_, const Dual<B>& ay_, const Dual<B>& bx_, const Dual<B>& by_,
                                            Dual<B> (&th)[K > 0 ? K : 1]) {
    constexpr int KK = K > 0 ? K : 1;
    Dual<B> px[K + 2], py[K + 2];
    Dual<B> mu[KK], nu[KK], g[KK];
    int nu_ = 0;
#pragma unroll
    for (int i = 0; i < KK; ++i) {
        th[i] = dconst<B>(theta0[i]);
        mu[i] = nu[i] = dconst<B>(0.0f);
        if (i < K && ob[i].kind != D2D_VERTEX) ++nu_;
    }
    Dual<B> last = dconst<B>(0.0f);
    for (int t = 0; t < A.steps; ++t) {
        dpoints<K, B>(ob, th, ax_, ay_, bx_, by_, px, py);
        last = dobjective_grad<K, B>(A.solver, ob, px, py, g);
        const float c1 = A.bc1[t], c2 = A.bc2[t];
#pragma unroll
        for (int q = 0; q < KK; ++q) {
            if (q < nu_) {
                mu[q] = A.b1 * mu[q] + A.omb1 * g[q];
                nu[q] = A.b2 * nu[q] + A.omb2 * (g[q] * g[q]);
                const Dual<B> mh = mu[q] / c1, nh = nu[q] / c2;
                th[q] = th[q] + (-A.lr) * (mh / (dsqrt(nh) + A.eps));
            }
        }
    }
    return last;
}

struct OptGradArgs {
    OptSweepArgs s;                    // the forward sweep's arguments (values come from its float code)
    float patch;
    const float* __restrict__ cot;     // [cells] cotangent of the value map, or null (= ones)
    float* __restrict__ contrib;       // [C][cells] valid * fun per candidate
    float* __restrict__ gcontrib;      // [C][cells][2] d contribution / d cell per candidate
    float* __restrict__ partial;       // [C * blocks][5 N + 2] per-workgroup partial sums of the scene VJP, or null
};

// valid * fun on Duals for K interacting objects (is_valid geometry.py:908-963 on the literal chain of literal_validity)
template <int K, int B>
__device__ __forceinline__ Dual<B> dcontribution(const OptGradArgs& a, const int (&cd)[D2D_MAX_ORDER], const DObj<B> (&ob)[K > 0 ? K : 1],
                                                 const Dual<B> (&px)[K + 2], const Dual<B> (&py)[K + 2], const Dual<B>& loss,
                                                 const int (&gdir)[B]) {
    const OptSweepArgs& s = a.s;
    // fun (utils.py:17-54, geometry.py:811-819)
    const Dual<B> r = dlength<K, B>(px, py);
    Dual<B> f;
    if (s.fun_id == D2D_FUN_RECEIVED_POWER) {
        float num = s.fnum[0];
#pragma unroll
        for (int q = 1; q <= D2D_MAX_ORDER; ++q)
            if (q == K) num = s.fnum[q];
        f = num / (s.h2 + r * r);
    } else if (s.fun_id == D2D_FUN_LENGTH_SQUARED) f = r * r;
    else if (s.fun_id == D2D_FUN_LENGTH) f = r;
    else f = dconst<B>(1.0f);
    if (s.mode == MODE_HARD) {
        // jnp.logical_* on booleans: no gradient through validity; valid from the forward float chain
        const Truth L{s.mode, s.alpha};
        float fpx[NP], fpy[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) fpx[i] = fpy[i] = __builtin_nanf("");
#pragma unroll
        for (int i = 0; i < K + 2; ++i) {
            fpx[i] = px[i].v;
            fpy[i] = py[i].v;
        }
        float on, hit, valid;
        literal_validity(s.T, L, K, cd, fpx, fpy, loss.v, s.tol, s.seg_lo, s.seg_hi, on, hit, valid);
        return valid * f;
    }
    const DTruth<B> L{s.mode, s.alpha};
    // on_objects (geometry.py:821-854)
    Dual<B> on = dconst<B>(1.0f);
#pragma unroll
    for (int i = 0; i < K; ++i) {
        Dual<B> cval = dconst<B>(1.0f);  // Vertex: true_value (geometry.py:397-403)
        if (ob[i].kind != D2D_VERTEX) {
            const Dual<B> dx = px[i + 1] - ob[i].ox, dy = py[i + 1] - ob[i].oy;
            const Dual<B> sp = (ob[i].tx * dx + ob[i].ty * dy) / ob[i].sq;
            cval = dmin(L.ge(sp, 0.0f), L.le(sp, 1.0f));
        }
        on = dmin(on, cval);
    }
    // intersects_with_objects (geometry.py:856-906)
    Dual<B> hit = dconst<B>(0.0f);
#pragma unroll
    for (int i = 0; i <= K; ++i) {
        const int ig0 = (i == 0) ? -1 : cd[i - 1];
        const int ig1 = (i == K) ? -1 : cd[i < D2D_MAX_ORDER ? i : 0];
        const Dual<B> bx = px[i] - px[i + 1], by = py[i] - py[i + 1];
        for (int j = 0; j < s.T.N; ++j) {
            if (j == ig0 || j == ig1) continue;
            if (s.T.kind[j] == D2D_VERTEX) continue;  // geometry.py:407-414
            const DObj<B> w = make_dobj<B>(s.T, j, a.patch, gdir);
            const Dual<B> Cx = w.p1x - px[i], Cy = w.p1y - py[i];
            const Dual<B> fa = by * Cx - bx * Cy;
            const Dual<B> fb = w.ax * Cy - w.ay * Cx;
            const Dual<B> fd = w.ay * bx - w.ax * by;
            const bool dz = (fd.v == 0.0f);
            const Dual<B> dd = dwhere(dz, dconst<B>(1.0f), fd);
            const Dual<B> ta = dwhere(dz, dconst<B>(__builtin_inff()), fa / dd);
            const Dual<B> tb = dwhere(dz, dconst<B>(__builtin_inff()), fb / dd);
            const Dual<B> h = dmin(dmin(L.ge(ta, s.seg_lo), L.le(ta, s.seg_hi)), dmin(L.ge(tb, s.seg_lo), L.le(tb, s.seg_hi)));
            hit = dmax(hit, h);
        }
    }
    const Dual<B> ok = L.act(s.tol - loss);  // less(loss, tol)
    Dual<B> valid = dmin(dmin(on, 1.0f - hit), ok);
    if (valid.v != valid.v) valid = dconst<B>(0.0f);  // jnp.nan_to_num: value 0, no tangent
    return valid * f;
}

// Everything for one (cell, candidate of order K): directional derivatives of valid * fun, B directions at a time.
//   grx, gry      += d / d cell
//   row[5 N + 2]  += cot * d / d (object end points [4 N], fixed end point [2], phi [N])   (LDS, wave-uniform targets)
template <int K, int B>
__device__ __forceinline__ void opt_grad_candidate(const OptGradArgs& a, int c, const int (&cd)[D2D_MAX_ORDER], float cellx, float celly,
                                                   float cot, bool active, float& grx, float& gry, float* row) {
    constexpr int KK = K > 0 ? K : 1;
    const OptSweepArgs& s = a.s;
    const int N = s.T.N;
    // ---- which start wins (optimize.py:136-182): the forward float runs, identical to the forward sweep's
    const float txx = s.grid_is_tx ? cellx : s.txx, txy = s.grid_is_tx ? celly : s.txy;
    const float rxx = s.grid_is_tx ? s.txx : cellx, rxy = s.grid_is_tx ? s.txy : celly;
    const float* th0 = s.theta0 + (long)c * s.A.many * D2D_MAX_ORDER;
    int best_m = 0;
    float best_th[D2D_MAX_ORDER] = {0.0f, 0.0f, 0.0f, 0.0f};
    float best_loss = 0.0f;  // what the winning run recorded at its last step (MinPath's path loss, geometry.py:1284-1288)
    if (K > 0) {
        float th[D2D_MAX_ORDER];
        best_loss = opt_run(s.T, s.A, K, cd, th0, txx, txy, rxx, rxy, best_th);
        for (int m = 1; m < s.A.many; ++m) {
            const float l = opt_run(s.T, s.A, K, cd, th0 + m * D2D_MAX_ORDER, txx, txy, rxx, rxy, th);
            const bool better = (l < best_loss) || (l != l && best_loss == best_loss);
            best_loss = better ? l : best_loss;
            best_m = better ? m : best_m;
#pragma unroll
            for (int q = 0; q < D2D_MAX_ORDER; ++q) best_th[q] = better ? th[q] : best_th[q];
        }
    }
    // ---- the directions, ordered: cell, fixed point, the candidate's own objects (each once), then the others
    // slot -> global direction; the first n_theta slots move theta
    int objs[KK];
    int n_own = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        bool seen = false;
#pragma unroll
        for (int q = 0; q < K; ++q)
            if (q < i && cd[q] == cd[i]) seen = true;
        if (!seen) objs[n_own++] = cd[i];
    }
    const int n_theta = 4 + 5 * n_own;
    const int n_dirs = (s.mode == MODE_HARD) ? n_theta : 4 + 5 * N;  // hard: other objects only gate a boolean
    auto slot_dir = [&](int slot) -> int {
        if (slot < 4) return slot;
        if (slot >= n_dirs) return -1;
        const int o = (slot - 4) / 5, comp = (slot - 4) % 5;
        if (o < n_own) {
            int id = objs[0];
#pragma unroll
            for (int q = 1; q < KK; ++q)
                if (q == o) id = objs[q];
            return dir_obj(id, comp);
        }
        // the (o - n_own)-th object that is not one of the candidate's
        int skip = o - n_own, id = -1;
        for (int j = 0; j < N; ++j) {
            bool own = false;
#pragma unroll
            for (int q = 0; q < KK; ++q)
                if (q < n_own && objs[q] == j) own = true;
            if (own) continue;
            if (skip == 0) {
                id = j;
                break;
            }
            --skip;
        }
        return id < 0 ? -1 : dir_obj(id, comp);
    };
    for (int b0 = 0; b0 < n_dirs; b0 += B) {
        int gdir[B];
#pragma unroll
        for (int i = 0; i < B; ++i) gdir[i] = slot_dir(b0 + i);
        // end points: exactly one is the lane's cell (direction 0, 1), the other the launch's fixed point (2, 3)
        const Dual<B> cx = dseed<B>(cellx, 0, gdir), cy = dseed<B>(celly, 1, gdir);
        const Dual<B> fx = dseed<B>(s.txx, 2, gdir), fy = dseed<B>(s.txy, 3, gdir);
        const Dual<B> ax_ = s.grid_is_tx ? cx : fx, ay_ = s.grid_is_tx ? cy : fy;  // transmitter
        const Dual<B> bx_ = s.grid_is_tx ? fx : cx, by_ = s.grid_is_tx ? fy : cy;  // receiver
        DObj<B> ob[KK];
#pragma unroll
        for (int i = 0; i < K; ++i) ob[i] = make_dobj<B>(s.T, cd[i], a.patch, gdir);
        Dual<B> th[KK];
        Dual<B> loss = dconst<B>(0.0f);
        if (K > 0) {
            if (b0 < n_theta) {
                loss = dopt_run<K, B>(s.A, ob, th0 + best_m * D2D_MAX_ORDER, ax_, ay_, bx_, by_, th);
            } else {
                // none of these directions moves theta: the solver's outputs are constants here
#pragma unroll
                for (int q = 0; q < KK; ++q) th[q] = dconst<B>(best_th[q]);
                loss = dconst<B>(best_loss);
            }
        }
        Dual<B> px[K + 2], py[K + 2];
        dpoints<K, B>(ob, th, ax_, ay_, bx_, by_, px, py);
        if (K > 0 && s.A.solver == D2D_SOLVER_FERMAT) loss = dinteraction_loss<K, B>(ob, px, py);  // geometry.py:1204
        const Dual<B> cv = dcontribution<K, B>(a, cd, ob, px, py, loss, gdir);
#pragma unroll
        for (int i = 0; i < B; ++i) {
            const int g = gdir[i];  // wave-uniform
            if (g < 0) continue;
            const float dv = active ? cv.d[i] : 0.0f;
            if (g == 0) grx += dv;
            else if (g == 1) gry += dv;
            else if (row) {
                const float sum = wave_sum(cot * dv);
                if ((threadIdx.x & 63) == 0) {
                    // layout of the VJP vector: [4 N] object end points, [2] fixed end point, [N] phi
                    const int at = (g < 4) ? 4 * N + (g - 2) : (((g - 4) % 5 == 4) ? 4 * N + 2 + (g - 4) / 5 : 4 * ((g - 4) / 5) + (g - 4) % 5);
                    row[at] += sum;
                }
            }
        }
    }
}

#ifdef D2D_OPTGRAD_KERNELS  // the two kernels are defined once, in d2d_optgrad.hip
constexpr int OPTGRAD_B = 9;  // tangents carried at a time: an order-1 candidate's 4 + 5 directions in one Adam pass

// One (cell, candidate) per lane; candidate = blockIdx.y (wave-uniform), as power_opt_cand_kernel.
__global__ void __launch_bounds__(64) power_opt_grad_kernel(OptGradArgs a) {
    extern __shared__ float row[];  // [5 N + 2]
    const OptSweepArgs& s = a.s;
    const int lane = threadIdx.x & 63;
    const int n_elem = 5 * s.T.N + 2;
    for (int i = lane; i < n_elem; i += 64) row[i] = 0.0f;
    __syncthreads();
    const long idx0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = idx0 < s.cells;
    const long idx = active ? idx0 : s.cells - 1;
    const int c = blockIdx.y;
    const float cellx = s.X[idx], celly = s.Y[idx];
    const float txx = s.grid_is_tx ? cellx : s.txx, txy = s.grid_is_tx ? celly : s.txy;
    const float rxx = s.grid_is_tx ? s.txx : cellx, rxy = s.grid_is_tx ? s.txy : celly;
    const float cot = active ? (a.cot ? a.cot[idx] : 1.0f) : 0.0f;
    const int k = s.order[c];
    int cd[D2D_MAX_ORDER];
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) cd[i] = s.cand[c * D2D_MAX_ORDER + i];
    float grx = 0.0f, gry = 0.0f;
    float* r = a.partial ? row : nullptr;
    switch (k) {  // wave-uniform
        case 0: opt_grad_candidate<0, OPTGRAD_B>(a, c, cd, cellx, celly, cot, active, grx, gry, r); break;
        case 1: opt_grad_candidate<1, OPTGRAD_B>(a, c, cd, cellx, celly, cot, active, grx, gry, r); break;
        case 2: opt_grad_candidate<2, OPTGRAD_B>(a, c, cd, cellx, celly, cot, active, grx, gry, r); break;
        case 3: opt_grad_candidate<3, OPTGRAD_B>(a, c, cd, cellx, celly, cot, active, grx, gry, r); break;
        default: opt_grad_candidate<4, OPTGRAD_B>(a, c, cd, cellx, celly, cot, active, grx, gry, r); break;
    }
    if (active) {
        a.contrib[(long)c * s.cells + idx] = opt_contribution(s, c, txx, txy, rxx, rxy);  // the forward sweep's own value
        a.gcontrib[((long)c * s.cells + idx) * 2] = grx;
        a.gcontrib[((long)c * s.cells + idx) * 2 + 1] = gry;
    }
    if (a.partial) {
        __syncthreads();
        float* dst = a.partial + ((long)c * gridDim.x + blockIdx.x) * n_elem;
        for (int i = lane; i < n_elem; i += 64) dst[i] = row[i];
    }
}

// values and per-cell gradients added up in candidate order (scene.py:1893-1916)
__global__ void __launch_bounds__(256) opt_grad_reduce_kernel(const float* __restrict__ contrib, const float* __restrict__ gcontrib, int C,
                                                              long cells, float* __restrict__ out, float* __restrict__ grad, int out_mode) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cells) return;
    float acc = 0.0f, gx = 0.0f, gy = 0.0f;
    for (int c = 0; c < C; ++c) {
        acc = acc + contrib[(long)c * cells + idx];
        gx = gx + gcontrib[((long)c * cells + idx) * 2];
        gy = gy + gcontrib[((long)c * cells + idx) * 2 + 1];
    }
    if (out_mode == D2D_OUT_ADD) {
        out[idx] = out[idx] + acc;
        grad[2 * idx] = grad[2 * idx] + gx;
        grad[2 * idx + 1] = grad[2 * idx + 1] + gy;
    } else {
        out[idx] = acc;
        grad[2 * idx] = gx;
        grad[2 * idx + 1] = gy;
    }
}

#endif  // D2D_OPTGRAD_KERNELS

// host-side launchers (d2d_optgrad.hip)
hipError_t launch_opt_grad(const OptGradArgs& a, dim3 grid, size_t lds, hipStream_t stream);
hipError_t launch_opt_grad_reduce(const float* contrib, const float* gcontrib, int C, long cells, float* out, float* grad, int out_mode,
                                  hipStream_t stream);

}  // namespace d2d
