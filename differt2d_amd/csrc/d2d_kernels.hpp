// Fused forward power-map sweep for gfx950 (MI355X): one RX cell per lane, all path
// candidates looped inside the kernel, wave-uniform wall/candidate data fetched with scalar
// loads, exact wave-level skips decided with ballots.
//
// Numerics contract (DESIGN.md "Exactness"): every value that reaches the output is computed
// with the reference's operations in the reference's order, one IEEE fp32 rounding each
// (-ffp-contract=off, correctly rounded divide/sqrt).  Work is skipped only where the skipped
// result provably cannot change the output bit pattern:
//   * a candidate whose validity is exactly 0 in every lane of the wave adds +0.0 to acc;
//   * and/or chains of hard_sigmoid activations are reduced before the (monotone) division by 6;
//   * a segment/wall test is decided without dividing when a*sign(d) vs (lo,hi)*|d| proves
//     t = fl(a/d) outside the soft window by a 1e-5 relative margin.
//
// Reference lines followed (DiffeRT2d v0.4.0): see include/d2d.h and oracle/ref.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/d2d.h"

namespace d2d {

enum Mode { MODE_HARD = 0, MODE_HSIG = 1, MODE_SIG = 2 };

struct SweepArgs {
    // scene tables (device, read-only, wave-uniform indexing -> scalar loads)
    const float4* __restrict__ occl;  // [N]  {p1x, p1y, Ax, Ay}: patched origin and P2-P1 (geometry.py:632-636)
    const float4* __restrict__ refl;  // [2N] {ox, oy, nx, ny}, {tx, ty, sq, 0}: reflection data
    const int* __restrict__ cw;       // [Nc] object indices allowed in candidates (filter_objects)
    int N, Nc;
    // grid
    const float* __restrict__ X;
    const float* __restrict__ Y;
    float* __restrict__ out;
    int m, n;
    // per-sweep scalars
    float txx, txy;
    int min_order, max_order;
    float alpha, tol;
    float seg_lo, seg_hi;      // -seg_tol, 1 + seg_tol (fp32, as geometry.py:168-169)
    float flt_lo, flt_hi;      // conservative "certainly outside the window" thresholds for the divide-free filter
    float fnum[D2D_MAX_ORDER + 1];  // r_coef ** k (lax.integer_pow), k = 0..D2D_MAX_ORDER
    float h2;                  // height * height
    int fun_id;
    int out_mode;
    unsigned long long* stats; // [9] executed-work counters (STATS build only), may be null
};

#define D2D_EPS 1.1920929e-07f  // jnp.finfo(float32).eps, geometry.py:200

// Executed-work counters of one wave (wave-uniform, live in SGPRs). Only the STATS build of the
// kernel touches them; the timed kernel is compiled without.
//   [0] candidates evaluated (points + on_objects)      [1] candidates that reached the loss stage
//   [2] candidates that reached the occlusion loop      [3] candidates that reached valid*fun
//   [4] segment/wall tests evaluated (filter)           [5] tests that took the exact-divide path
//   [6] sum over [0] of the candidate order k           [7] sum over [1] of k   [8] sum over [3] of (k+1)
struct WaveStats {
    unsigned long long c[9];
};

__device__ __forceinline__ bool wave_any(bool p) { return __any(p); }

// Reflection of a point (wave-uniform data, geometry.py:652-670).
__device__ __forceinline__ void image_of(const float4& r0, float px, float py, float& ox, float& oy) {
    float ix = px - r0.x, iy = py - r0.y;
    float dn = ix * r0.z + iy * r0.w;
    float s = 2.0f * dn;
    ox = px - s * r0.z;
    oy = py - s * r0.w;
}

// geometry.py:206-230
__device__ __forceinline__ void normalize2(float vx, float vy, float& ox, float& oy) {
    float len = sqrtf(vx * vx + vy * vy);
    len = (len == 0.0f) ? 1.0f : len;
    ox = vx / len;
    oy = vy / len;
}

// Pre-division hard_sigmoid: clamp(alpha*x + 3, 0, 6); hard_sigmoid(x) = clampact(x) / 6.
__device__ __forceinline__ float clampact(float x, float alpha) {
    float z = alpha * x;
    return fminf(fmaxf(z + 3.0f, 0.0f), 6.0f);
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

template <int K, int MODE, bool STATS>
__device__ __forceinline__ void eval_candidate(const SweepArgs& a, const int (&cand)[D2D_MAX_ORDER],
                                               const float (&imgx)[D2D_MAX_ORDER], const float (&imgy)[D2D_MAX_ORDER],
                                               float rxx, float rxy, bool lane_bad, float& acc, WaveStats& st) {
    if (STATS) {
        st.c[0] += 1;
        st.c[6] += K;
    }
    float px[K + 2], py[K + 2];
    px[0] = a.txx;
    py[0] = a.txy;
    px[K + 1] = rxx;
    py[K + 1] = rxy;

    // ---- backward scan of the image method, geometry.py:1093-1110 -------------------------
    {
        float ptx = rxx, pty = rxy;
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
            const float4 r0 = a.refl[2 * cand[i]];
            float ux = ptx - imgx[i], uy = pty - imgy[i];
            float vx = r0.x - ptx, vy = r0.y - pty;
            float un = ux * r0.z + uy * r0.w;
            float vn = vx * r0.z + vy * r0.w;
            bool z = (un == 0.0f);
            float den = z ? 1.0f : un;
            float incx = z ? 0.0f : (vn * ux) / den;
            float incy = z ? 0.0f : (vn * uy) / den;
            ptx = ptx + incx;
            pty = pty + incy;
            px[i + 1] = ptx;
            py[i + 1] = pty;
        }
    }

    // Lanes whose coordinates are not comfortably finite never take part in a skip decision
    // (0 * fun must then be evaluated for real: it may be NaN).
    bool bad = lane_bad;
#pragma unroll
    for (int i = 1; i <= K; ++i) bad = bad || !(fabsf(px[i]) < 1e18f) || !(fabsf(py[i]) < 1e18f);

    // ---- on_objects, geometry.py:821-854 / 589-621 ----------------------------------------
    bool on_b = true;     // MODE_HARD
    float on_c = 6.0f;    // MODE_HSIG: min of clamped pre-activations (true_value = 6/6)
    float on_z = 3.0e38f; // MODE_SIG: min of alpha*x (true_value = 1.0 handled at the end)
    bool nanflag = false;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const float4 r0 = a.refl[2 * cand[i]];
        const float4 r1 = a.refl[2 * cand[i] + 1];
        float dx = px[i + 1] - r0.x, dy = py[i + 1] - r0.y;
        float s = (r1.x * dx + r1.y * dy) / r1.z;
        if (MODE == MODE_HARD) {
            on_b = on_b && (s >= 0.0f) && (s <= 1.0f);
        } else if (MODE == MODE_HSIG) {
            nanflag = nanflag || (s != s);
            on_c = fminf(on_c, fminf(clampact(s - 0.0f, a.alpha), clampact(1.0f - s, a.alpha)));
        } else {
            nanflag = nanflag || (s != s);
            on_z = fminf(on_z, fminf(a.alpha * (s - 0.0f), a.alpha * (1.0f - s)));
        }
    }
    // sigmoid(z) is exactly 0 only once exp(-z) overflows: z <= -89
    bool on_zero = (MODE == MODE_HARD) ? !on_b : (MODE == MODE_HSIG) ? (on_c == 0.0f) : (on_z <= -89.0f);
    if (K > 0 && !wave_any(!on_zero || bad)) return;  // valid == 0 in every lane: acc + 0.0

    if (STATS) {
        st.c[1] += 1;
        st.c[7] += K;
    }
    // ---- path loss, geometry.py:1077-1084 / 641-650 ---------------------------------------
    float loss = 0.0f;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const float4 r0 = a.refl[2 * cand[i]];
        float ix, iy, rx_, ry_;
        normalize2(px[i + 1] - px[i], py[i + 1] - py[i], ix, iy);
        normalize2(px[i + 2] - px[i + 1], py[i + 2] - py[i + 1], rx_, ry_);
        float din = ix * r0.z + iy * r0.w;
        float s2 = 2.0f * din;
        float ex = rx_ - (ix - s2 * r0.z);
        float ey = ry_ - (iy - s2 * r0.w);
        loss = loss + (ex * ex + ey * ey);
    }
    bool ok_b = loss < a.tol;                                    // hard: jnp.less
    float ok_x = a.tol - loss;                                   // approx: activation(tol - loss)
    if (MODE != MODE_HARD) nanflag = nanflag || (loss != loss);

    // Lanes for which the occlusion result can still change the output.
    bool live;
    if (MODE == MODE_HARD) live = (on_b && ok_b) || bad;
    else if (MODE == MODE_HSIG) live = !(on_zero || clampact(ok_x, a.alpha) == 0.0f) || bad;
    else live = !(on_zero || a.alpha * ok_x <= -89.0f) || bad;
    if (!wave_any(live)) return;
    if (STATS) st.c[2] += 1;

    // ---- intersects_with_objects, geometry.py:856-906 / 623-639 / 82-173 -------------------
    float bx[K + 1], by[K + 1];
#pragma unroll
    for (int i = 0; i <= K; ++i) {
        bx[i] = px[i] - px[i + 1];  // B = P3 - P4
        by[i] = py[i] - py[i + 1];
    }
    bool hit_b = false;     // MODE_HARD
    float hit_c = 0.0f;     // MODE_HSIG (false_value = 0/6)
    float hit_z = -3.0e38f; // MODE_SIG: max over tests of min(z1..z4); "no test yet" = false_value handled below
    bool any_test = false;
    bool active = live;     // lanes still undecided
    for (int j = 0; j < a.N; ++j) {
        const float4 w = a.occl[j];
#pragma unroll
        for (int i = 0; i <= K; ++i) {
            const int ig0 = (i == 0) ? -1 : cand[i - 1];
            const int ig1 = (i == K) ? -1 : cand[i];
            if (j == ig0 || j == ig1) continue;  // wave-uniform
            float Cx = w.x - px[i], Cy = w.y - py[i];
            float fa = by[i] * Cx - bx[i] * Cy;   // geometry.py:157
            float fb = w.z * Cy - w.w * Cx;       // geometry.py:158
            float fd = w.w * bx[i] - w.z * by[i]; // geometry.py:159
            // divide-free filter: t = fl(num/fd) certainly outside [flt_lo, flt_hi]?
            float D = fabsf(fd);
            float ua = (fd < 0.0f) ? -fa : fa;
            float ub = (fd < 0.0f) ? -fb : fb;
            float lo = a.flt_lo * D, hi = a.flt_hi * D;
            bool miss = (fd == 0.0f) || ((D >= 1e-30f) && ((ua < lo) || (ua > hi) || (ub < lo) || (ub > hi)));
            if (MODE == MODE_SIG) any_test = true;
            if (STATS) st.c[4] += 1;
            const bool need = wave_any(active && (!miss || bad));
            if (STATS && need) st.c[5] += 1;
            if (need) {
                // exact path, geometry.py:163-171
                bool dz = (fd == 0.0f);
                float dd = dz ? 1.0f : fd;
                float ta = dz ? __builtin_inff() : fa / dd;
                float tb = dz ? __builtin_inff() : fb / dd;
                if (MODE == MODE_HARD) {
                    bool h = (ta >= a.seg_lo) && (ta <= a.seg_hi) && (tb >= a.seg_lo) && (tb <= a.seg_hi);
                    hit_b = hit_b || h;
                } else if (MODE == MODE_HSIG) {
                    nanflag = nanflag || (ta != ta) || (tb != tb);
                    float c = fminf(fminf(clampact(ta - a.seg_lo, a.alpha), clampact(a.seg_hi - ta, a.alpha)),
                                    fminf(clampact(tb - a.seg_lo, a.alpha), clampact(a.seg_hi - tb, a.alpha)));
                    hit_c = fmaxf(hit_c, c);
                } else {
                    nanflag = nanflag || (ta != ta) || (tb != tb);
                    float z = fminf(fminf(a.alpha * (ta - a.seg_lo), a.alpha * (a.seg_hi - ta)),
                                    fminf(a.alpha * (tb - a.seg_lo), a.alpha * (a.seg_hi - tb)));
                    hit_z = fmaxf(hit_z, z);
                }
            }
        }
        // decided lanes: occlusion already makes valid exactly 0
        if (MODE == MODE_HARD) active = active && (!hit_b || bad);
        else if (MODE == MODE_HSIG) active = active && (hit_c != 6.0f || bad);
        else active = active && (hit_z < 17.5f || bad);
        if (!wave_any(active)) break;
    }

    if (STATS) {
        st.c[3] += 1;
        st.c[8] += K + 1;
    }
    // ---- is_valid, geometry.py:947-963 -----------------------------------------------------
    float valid;
    if (MODE == MODE_HARD) {
        valid = (on_b && !hit_b && ok_b) ? 1.0f : 0.0f;
    } else if (MODE == MODE_HSIG) {
        float on_v = on_c / 6.0f;
        float hit_v = hit_c / 6.0f;
        float ok_v = clampact(ok_x, a.alpha) / 6.0f;
        valid = fminf(fminf(on_v, 1.0f - hit_v), ok_v);
        valid = nanflag ? 0.0f : valid;  // NaN-propagating min, then nan_to_num
    } else {
        float on_v = (K == 0) ? 1.0f : sigmoidf_(on_z);
        float hit_v = any_test ? fmaxf(0.0f, sigmoidf_(hit_z)) : 0.0f;
        float ok_v = sigmoidf_(a.alpha * ok_x);
        valid = fminf(fminf(on_v, 1.0f - hit_v), ok_v);
        valid = nanflag ? 0.0f : valid;
    }

    // ---- fun(path), geometry.py:176-203 and utils.py:17-54 ---------------------------------
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i <= K; ++i) {
        float vx = (px[i + 1] - px[i]) + D2D_EPS;
        float vy = (py[i + 1] - py[i]) + D2D_EPS;
        r = r + sqrtf(vx * vx + vy * vy);
    }
    float f;
    if (a.fun_id == D2D_FUN_RECEIVED_POWER) f = a.fnum[K] / (a.h2 + r * r);
    else if (a.fun_id == D2D_FUN_LENGTH_SQUARED) f = r * r;
    else if (a.fun_id == D2D_FUN_LENGTH) f = r;
    else f = 1.0f;
    acc = acc + valid * f;  // scene.py:1909
}

// All candidates of order K in lexicographic order (scene.py:122-175), images built incrementally
// (geometry.py:1086-1091, 1109).
template <int K, int MODE, bool STATS>
__device__ __forceinline__ void sweep_order(const SweepArgs& a, float rxx, float rxy, bool lane_bad, float& acc,
                                            WaveStats& st) {
    int cand[D2D_MAX_ORDER] = {-1, -1, -1, -1};
    float imgx[D2D_MAX_ORDER], imgy[D2D_MAX_ORDER];
    if (K == 0) {
        eval_candidate<0, MODE, STATS>(a, cand, imgx, imgy, rxx, rxy, lane_bad, acc, st);
        return;
    }
    for (int i0 = 0; i0 < a.Nc; ++i0) {
        cand[0] = a.cw[i0];
        image_of(a.refl[2 * cand[0]], a.txx, a.txy, imgx[0], imgy[0]);
        if (K == 1) {
            eval_candidate<K, MODE, STATS>(a, cand, imgx, imgy, rxx, rxy, lane_bad, acc, st);
            continue;
        }
        for (int i1 = 0; i1 < a.Nc; ++i1) {
            cand[1] = a.cw[i1];
            if (cand[1] == cand[0]) continue;
            image_of(a.refl[2 * cand[1]], imgx[0], imgy[0], imgx[1], imgy[1]);
            if (K == 2) {
                eval_candidate<K, MODE, STATS>(a, cand, imgx, imgy, rxx, rxy, lane_bad, acc, st);
                continue;
            }
            for (int i2 = 0; i2 < a.Nc; ++i2) {
                cand[2] = a.cw[i2];
                if (cand[2] == cand[1]) continue;
                image_of(a.refl[2 * cand[2]], imgx[1], imgy[1], imgx[2], imgy[2]);
                if (K == 3) {
                    eval_candidate<K, MODE, STATS>(a, cand, imgx, imgy, rxx, rxy, lane_bad, acc, st);
                    continue;
                }
                for (int i3 = 0; i3 < a.Nc; ++i3) {
                    cand[3] = a.cw[i3];
                    if (cand[3] == cand[2]) continue;
                    image_of(a.refl[2 * cand[3]], imgx[2], imgy[2], imgx[3], imgy[3]);
                    eval_candidate<(K >= 4 ? 4 : K), MODE, STATS>(a, cand, imgx, imgy, rxx, rxy, lane_bad, acc, st);
                }
            }
        }
    }
}

constexpr int TILE_W = 8;  // a wave covers an 8 x 8 patch of RX cells: neighbouring cells share skips
constexpr int TILE_H = 8;

template <int MODE, bool STATS>
__global__ void __launch_bounds__(64) power_fwd_kernel(SweepArgs a) {
    const int lane = threadIdx.x & 63;
    const int tiles_x = (a.n + TILE_W - 1) / TILE_W;
    const int tile = blockIdx.x;
    const int tcol = tile % tiles_x, trow = tile / tiles_x;
    const int col = tcol * TILE_W + (lane & (TILE_W - 1));
    const int row = trow * TILE_H + (lane / TILE_W);
    const bool in_range = (col < a.n) && (row < a.m);
    const int ccol = col < a.n ? col : a.n - 1;
    const int crow = row < a.m ? row : a.m - 1;
    const long idx = (long)crow * a.n + ccol;
    const float rxx = a.X[idx], rxy = a.Y[idx];
    const bool lane_bad = !(fabsf(rxx) < 1e18f) || !(fabsf(rxy) < 1e18f) || !(fabsf(a.txx) < 1e18f) ||
                          !(fabsf(a.txy) < 1e18f);
    float acc = 0.0f;  // scene.py:1893
    WaveStats st;
#pragma unroll
    for (int i = 0; i < 9; ++i) st.c[i] = 0;
    if (a.min_order <= 0 && a.max_order >= 0) sweep_order<0, MODE, STATS>(a, rxx, rxy, lane_bad, acc, st);
    if (a.min_order <= 1 && a.max_order >= 1) sweep_order<1, MODE, STATS>(a, rxx, rxy, lane_bad, acc, st);
    if (a.min_order <= 2 && a.max_order >= 2) sweep_order<2, MODE, STATS>(a, rxx, rxy, lane_bad, acc, st);
    if (a.min_order <= 3 && a.max_order >= 3) sweep_order<3, MODE, STATS>(a, rxx, rxy, lane_bad, acc, st);
    if (a.min_order <= 4 && a.max_order >= 4) sweep_order<4, MODE, STATS>(a, rxx, rxy, lane_bad, acc, st);
    if (in_range) {
        if (a.out_mode == D2D_OUT_ADD) a.out[idx] = a.out[idx] + acc;
        else a.out[idx] = acc;
    }
    if (STATS && lane == 0 && a.stats) {
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicAdd(&a.stats[i], st.c[i]);
    }
}

}  // namespace d2d

// =====================================================================================
// Path tracing kernel ("emit paths"): one thread per (tx/rx pair, candidate); writes the
// interaction points, the loss and the validity of every candidate.  Serves
// Scene.all_paths / all_valid_paths / accumulate_over_paths (scene.py:1156-1334),
// ImagePath.from_tx_objects_rx (geometry.py:1013-1114) and Path.is_valid / on_objects /
// intersects_with_objects (geometry.py:821-963) of the host mirror.  Small problem sizes:
// written literally (every activation evaluated, NaN-propagating min/max), no skipping.
// =====================================================================================
namespace d2d {

struct TraceArgs {
    const float4* __restrict__ occl;
    const float4* __restrict__ refl;
    const unsigned char* __restrict__ kind;  // [N] D2D_WALL / D2D_RIS / D2D_VERTEX
    const float* __restrict__ phi;           // [N]
    int N;
    const int* __restrict__ cand;   // [C][D2D_MAX_ORDER]
    const int* __restrict__ order;  // [C]
    int C;
    const float* __restrict__ tx;  // [P][2]
    const float* __restrict__ rx;  // [P][2]
    int P;
    const float* __restrict__ xys_in;   // [P][C][D2D_MAX_ORDER+2][2] or null: validate these paths instead of solving
    const float* __restrict__ loss_in;  // [P][C] or null
    float* __restrict__ xys;    // [P][C][D2D_MAX_ORDER+2][2]
    float* __restrict__ loss;   // [P][C]
    float* __restrict__ valid;  // [P][C]   final is_valid
    float* __restrict__ on;     // [P][C]   on_objects            (may be null)
    float* __restrict__ hit;    // [P][C]   intersects_with_objects (may be null)
    float* __restrict__ length; // [P][C]   path_length           (may be null)
    int mode;  // MODE_*
    float alpha, tol, seg_lo, seg_hi;
};

// NaN-propagating min / max (jnp.minimum / jnp.maximum)
__device__ __forceinline__ float minp(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a < b ? a : b); }
__device__ __forceinline__ float maxp(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a > b ? a : b); }

struct Truth {
    int mode;
    float alpha;
    __device__ float act(float x) const {
        float z = alpha * x;
        if (mode == MODE_HSIG) return minp(maxp(z + 3.0f, 0.0f), 6.0f) / 6.0f;
        return 1.0f / (1.0f + expf(-z));
    }
    __device__ float t_and(float a, float b) const { return mode ? minp(a, b) : ((a != 0.0f && b != 0.0f) ? 1.0f : 0.0f); }
    __device__ float t_or(float a, float b) const { return mode ? maxp(a, b) : ((a != 0.0f || b != 0.0f) ? 1.0f : 0.0f); }
    __device__ float t_not(float a) const { return mode ? (1.0f - a) : (a != 0.0f ? 0.0f : 1.0f); }
    __device__ float ge(float x, float y) const { return mode ? act(x - y) : (x >= y ? 1.0f : 0.0f); }
    __device__ float le(float x, float y) const { return mode ? act(y - x) : (x <= y ? 1.0f : 0.0f); }
    __device__ float lt(float x, float y) const { return mode ? act(y - x) : (x < y ? 1.0f : 0.0f); }
};

__global__ void __launch_bounds__(64) trace_kernel(TraceArgs a) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long)a.P * a.C) return;
    const int p = (int)(tid / a.C), c = (int)(tid % a.C);
    const int k = a.order[c];
    int cd[D2D_MAX_ORDER];
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) cd[i] = a.cand[c * D2D_MAX_ORDER + i];
    constexpr int NP = D2D_MAX_ORDER + 2;
    float px[NP], py[NP];
    const float txx = a.tx[2 * p], txy = a.tx[2 * p + 1], rxx = a.rx[2 * p], rxy = a.rx[2 * p + 1];
    const Truth T{a.mode, a.alpha};
    float loss = 0.0f;

    if (a.xys_in) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            px[i] = a.xys_in[(tid * NP + i) * 2];
            py[i] = a.xys_in[(tid * NP + i) * 2 + 1];
        }
        loss = a.loss_in ? a.loss_in[tid] : 0.0f;
    } else {
#pragma unroll
        for (int i = 0; i < NP; ++i) px[i] = py[i] = __builtin_nanf("");
        px[0] = txx;
        py[0] = txy;
        // forward images, geometry.py:1086-1091
        float imx[D2D_MAX_ORDER], imy[D2D_MAX_ORDER];
        float ix = txx, iy = txy;
#pragma unroll
        for (int i = 0; i < D2D_MAX_ORDER; ++i) {
            if (i < k) {
                float ox, oy;
                image_of(a.refl[2 * cd[i]], ix, iy, ox, oy);
                ix = ox;
                iy = oy;
                imx[i] = ix;
                imy[i] = iy;
            }
        }
        // backward scan, geometry.py:1093-1110
        float ptx = rxx, pty = rxy;
#pragma unroll
        for (int i = D2D_MAX_ORDER - 1; i >= 0; --i) {
            if (i < k) {
                const float4 r0 = a.refl[2 * cd[i]];
                float ux = ptx - imx[i], uy = pty - imy[i];
                float vx = r0.x - ptx, vy = r0.y - pty;
                float un = ux * r0.z + uy * r0.w;
                float vn = vx * r0.z + vy * r0.w;
                bool z = (un == 0.0f);
                float den = z ? 1.0f : un;
                float incx = z ? 0.0f : (vn * ux) / den;
                float incy = z ? 0.0f : (vn * uy) / den;
                ptx = ptx + incx;
                pty = pty + incy;
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    if (q == i + 1) {
                        px[q] = ptx;
                        py[q] = pty;
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (q == k + 1) {
                px[q] = rxx;
                py[q] = rxy;
            }
        // path loss, geometry.py:1077-1084
#pragma unroll
        for (int i = 0; i < D2D_MAX_ORDER; ++i) {
            if (i < k) {
                const float4 r0 = a.refl[2 * cd[i]];
                float ix_, iy_, rx_, ry_;
                normalize2(px[i + 1] - px[i], py[i + 1] - py[i], ix_, iy_);
                normalize2(px[i + 2] - px[i + 1], py[i + 2] - py[i + 1], rx_, ry_);
                float din = ix_ * r0.z + iy_ * r0.w;
                float s2 = 2.0f * din;
                float ex = rx_ - (ix_ - s2 * r0.z);
                float ey = ry_ - (iy_ - s2 * r0.w);
                loss = loss + (ex * ex + ey * ey);
            }
        }
    }

    // on_objects, geometry.py:821-854
    float on = 1.0f;
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) {
        if (i < k) {
            float cval;
            if (a.kind[cd[i]] == D2D_VERTEX) {
                cval = 1.0f;  // geometry.py:397-403
            } else {
                const float4 r0 = a.refl[2 * cd[i]];
                const float4 r1 = a.refl[2 * cd[i] + 1];
                float dx = px[i + 1] - r0.x, dy = py[i + 1] - r0.y;
                float s = (r1.x * dx + r1.y * dy) / r1.z;
                cval = T.t_and(T.ge(s, 0.0f), T.le(s, 1.0f));
            }
            on = T.t_and(on, cval);
        }
    }
    // intersects_with_objects, geometry.py:856-906
    float hit = 0.0f;
#pragma unroll
    for (int i = 0; i <= D2D_MAX_ORDER; ++i) {
        if (i <= k) {
            const int ig0 = (i == 0) ? -1 : cd[i - 1];
            const int ig1 = (i == k) ? -1 : cd[i < D2D_MAX_ORDER ? i : 0];
            const float bx = px[i] - px[i + 1], by = py[i] - py[i + 1];
            for (int j = 0; j < a.N; ++j) {
                if (j == ig0 || j == ig1) continue;
                if (a.kind[j] == D2D_VERTEX) continue;  // geometry.py:407-414: false_value, or() leaves hit unchanged
                const float4 w = a.occl[j];
                float Cx = w.x - px[i], Cy = w.y - py[i];
                float fa = by * Cx - bx * Cy;
                float fb = w.z * Cy - w.w * Cx;
                float fd = w.w * bx - w.z * by;
                bool dz = (fd == 0.0f);
                float dd = dz ? 1.0f : fd;
                float ta = dz ? __builtin_inff() : fa / dd;
                float tb = dz ? __builtin_inff() : fb / dd;
                float h = T.t_and(T.t_and(T.ge(ta, a.seg_lo), T.le(ta, a.seg_hi)), T.t_and(T.ge(tb, a.seg_lo), T.le(tb, a.seg_hi)));
                hit = T.t_or(hit, h);
            }
        }
    }
    float ok = T.lt(loss, a.tol);
    float valid = T.t_and(T.t_and(on, T.t_not(hit)), ok);
    if (valid != valid) valid = 0.0f;  // jnp.nan_to_num
    // path length, geometry.py:176-203
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i <= D2D_MAX_ORDER; ++i) {
        if (i <= k) {
            float vx = (px[i + 1] - px[i]) + D2D_EPS;
            float vy = (py[i + 1] - py[i]) + D2D_EPS;
            r = r + sqrtf(vx * vx + vy * vy);
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        a.xys[(tid * NP + i) * 2] = px[i];
        a.xys[(tid * NP + i) * 2 + 1] = py[i];
    }
    a.loss[tid] = loss;
    a.valid[tid] = valid;
    if (a.on) a.on[tid] = on;
    if (a.hit) a.hit[tid] = hit;
    if (a.length) a.length[tid] = r;
}

}  // namespace d2d
