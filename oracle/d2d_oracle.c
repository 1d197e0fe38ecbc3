/*
 * CPU ORACLE (C) -- test infrastructure only, never shipped, never on the product path.
 *
 * Plain-C restatement of DiffeRT2d v0.4.0's fused power-map sweep
 * (Scene.accumulate_on_receivers_grid_over_paths, ImagePath solver, Wall objects):
 * one rounding per fp32 operation, operations in the order the reference writes them.
 * Build with  gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp  (oracle/Makefile):
 * contraction off so that a*b+c never becomes an fma, which is what "bit-exact
 * intersection counts" is defined against.
 *
 * It mirrors oracle/ref.py function by function (that file carries the line-by-line
 * citations and is what the reference's known answers are checked against;
 * tests/test_oracle_c.py then checks this file == ref.py bit for bit).
 *
 * Reference lines followed (paths relative to the DiffeRT2d checkout):
 *   candidates            differt2d/scene.py:122-175 (lexicographic, no equal neighbours)
 *   image path            differt2d/geometry.py:1013-1114
 *   wall normal/image     differt2d/geometry.py:561-573, 652-670, 206-230
 *   specular residual     differt2d/geometry.py:641-650
 *   on_objects            differt2d/geometry.py:821-854, 589-621
 *   intersects            differt2d/geometry.py:856-906, 623-639, 82-173
 *   is_valid              differt2d/geometry.py:908-963
 *   soft logic            differt2d/logic.py:218-537
 *   path length           differt2d/geometry.py:176-203
 *   received power        differt2d/utils.py:17-54
 *   accumulation          differt2d/scene.py:1892-1918
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_ORDER 4

typedef struct orc_params {
    int32_t min_order, max_order;
    int32_t approx;   /* 0: jnp.logical_*, 1: min/max/activation */
    int32_t act;      /* 0: hard_sigmoid, 1: sigmoid */
    float alpha;      /* logic.py:262 */
    float tol;        /* geometry.py:915 (loss tolerance) */
    float patch;      /* geometry.py:916 */
    float seg_tol;    /* geometry.py:89 */
    int32_t fun_id;   /* 0 received_power, 1 length**2, 2 length, 3 one */
    float r_coef, height;
    int32_t prune;    /* 0: evaluate everything; 1: skip the occlusion tests of a candidate whose on_objects is exactly 0;
                         2: also leave the candidate at the first wall (in the backward scan's order) whose on_objects term is
                         exactly 0 -- per cell and candidate, exact fp32 arithmetic on the reference's own values: nothing
                         here knows about the GPU's patch-level culling */
    int32_t grid_is_tx; /* 0: grid cells are receivers (scene.py:1803-1953); 1: transmitters, `tx` is the fixed receiver (scene.py:1489-1648) */
} orc_params;

typedef struct {
    float ox, oy, dx, dy; /* origin, dest */
    float tx_, ty_;       /* t = dest - origin */
    float nx, ny;         /* normal */
    float p1x, p1y, p2x, p2y; /* patched end points origin - patch*t, dest + patch*t */
} wall_t;

/* NaN-propagating min / max (jnp.minimum / jnp.maximum). */
static inline float minp(float a, float b) { return (a != a || b != b) ? NAN : (a < b ? a : b); }
static inline float maxp(float a, float b) { return (a != a || b != b) ? NAN : (a > b ? a : b); }

static inline float activation(float x, const orc_params* p) {
    float z = p->alpha * x;
    if (p->act == 0) /* jax.nn.hard_sigmoid: relu6(z + 3) / 6 */
        return minp(maxp(z + 3.0f, 0.0f), 6.0f) / 6.0f;
    return 1.0f / (1.0f + expf(-z)); /* jax.nn.sigmoid */
}

/* Truthy value: approx -> float in [0,1]; hard -> 0.0f / 1.0f standing for False / True. */
static inline float t_and(float a, float b, int approx) { return approx ? minp(a, b) : ((a != 0.0f && b != 0.0f) ? 1.0f : 0.0f); }
static inline float t_or(float a, float b, int approx) { return approx ? maxp(a, b) : ((a != 0.0f || b != 0.0f) ? 1.0f : 0.0f); }
static inline float t_not(float a, int approx) { return approx ? (1.0f - a) : (a != 0.0f ? 0.0f : 1.0f); }
static inline float t_ge(float x, float y, const orc_params* p) { return p->approx ? activation(x - y, p) : (x >= y ? 1.0f : 0.0f); }
static inline float t_le(float x, float y, const orc_params* p) { return p->approx ? activation(y - x, p) : (x <= y ? 1.0f : 0.0f); }
static inline float t_lt(float x, float y, const orc_params* p) { return p->approx ? activation(y - x, p) : (x < y ? 1.0f : 0.0f); }

static void make_wall(wall_t* w, const float* xy, float patch) {
    w->ox = xy[0]; w->oy = xy[1]; w->dx = xy[2]; w->dy = xy[3];
    w->tx_ = w->dx - w->ox; w->ty_ = w->dy - w->oy;
    /* normal: normalize((t_y, -t_x)) */
    float vx = w->ty_, vy = -w->tx_;
    float len = sqrtf(vx * vx + vy * vy);
    if (len == 0.0f) len = 1.0f;
    w->nx = vx / len; w->ny = vy / len;
    w->p1x = w->ox - patch * w->tx_; w->p1y = w->oy - patch * w->ty_;
    w->p2x = w->dx + patch * w->tx_; w->p2y = w->dy + patch * w->ty_;
}

static inline void normalize2(float vx, float vy, float* ox, float* oy) {
    float len = sqrtf(vx * vx + vy * vy);
    if (len == 0.0f) len = 1.0f;
    *ox = vx / len; *oy = vy / len;
}

/* geometry.py:82-173 with P1,P2 = patched wall, P3,P4 = path segment */
static inline float seg_test(float num, float den, const orc_params* p) {
    int den_is_zero = (den == 0.0f);
    float dd = den_is_zero ? 1.0f : den;
    float t = den_is_zero ? INFINITY : num / dd;
    return t_and(t_ge(t, -p->seg_tol, p), t_le(t, 1.0f + p->seg_tol, p), p->approx);
}

static inline float wall_hits(const wall_t* w, float p3x, float p3y, float p4x, float p4y, const orc_params* p) {
    float Ax = w->p2x - w->p1x, Ay = w->p2y - w->p1y;
    float Bx = p3x - p4x, By = p3y - p4y;
    float Cx = w->p1x - p3x, Cy = w->p1y - p3y;
    float a = By * Cx - Bx * Cy;
    float b = Ax * Cy - Ay * Cx;
    float d = Ay * Bx - Ax * By;
    return t_and(seg_test(a, d, p), seg_test(b, d, p), p->approx);
}

static inline float ipow(float x, int n) { /* lax.integer_pow */
    if (n == 0) return 1.0f;
    float acc = 0.0f; int have = 0;
    while (n > 0) {
        if (n & 1) { acc = have ? acc * x : x; have = 1; }
        n >>= 1;
        if (n > 0) x = x * x;
    }
    return acc;
}

/* One (rx, candidate) evaluation: returns valid (as float) and the path function value. */
static void eval_candidate(const wall_t* W, int N, const int* cand, int k, const float* img /*[k][2]*/,
                           float txx, float txy, float rxx, float rxy, const orc_params* p,
                           float* valid_out, float* fun_out) {
    float px[ORC_MAX_ORDER + 2], py[ORC_MAX_ORDER + 2];
    px[0] = txx; py[0] = txy; px[k + 1] = rxx; py[k + 1] = rxy;
    float loss = 0.0f;
    if (k > 0) {
        /* backward scan, geometry.py:1093-1110 */
        float ptx = rxx, pty = rxy;
        for (int i = k - 1; i >= 0; --i) {
            const wall_t* w = &W[cand[i]];
            float ux = ptx - img[2 * i], uy = pty - img[2 * i + 1];
            float vx = w->ox - ptx, vy = w->oy - pty;
            float un = ux * w->nx + uy * w->ny;
            float vn = vx * w->nx + vy * w->ny;
            float incx, incy;
            if (un == 0.0f) { incx = 0.0f; incy = 0.0f; }
            else { incx = (vn * ux) / un; incy = (vn * uy) / un; }
            ptx = ptx + incx; pty = pty + incy;
            px[i + 1] = ptx; py[i + 1] = pty;
            if (p->prune >= 2) {
                /* prune level 2: this wall's term of on_objects (the very expression evaluated below, on the very same
                 * point) right away.  Exactly 0 here makes on_objects -- a min / and over the walls -- exactly 0 and the
                 * contribution valid * fun = 0 * fun, as at prune level 1; the remaining points, the loss and the occlusion
                 * tests are then never computed.  (fun is finite: for coordinates of order 1 a non-zero u.n is no smaller
                 * than an ulp of its terms, so the interaction points stay of order 1e9 at most and the path length is
                 * finite.  tests/test_oracle_c.py checks level 2 against levels 0 and 1 cell for cell.) */
                float ox_ = ptx - w->ox, oy_ = pty - w->oy;
                float sq = w->tx_ * w->tx_ + w->ty_ * w->ty_;
                if (sq == 0.0f) sq = 1.0f;
                float s = (w->tx_ * ox_ + w->ty_ * oy_) / sq;
                float c = t_and(t_ge(s, 0.0f, p), t_le(s, 1.0f, p), p->approx);
                if (c == 0.0f && isfinite(ptx) && isfinite(pty)) {
                    *valid_out = 0.0f;
                    *fun_out = 0.0f;
                    return;
                }
            }
        }
    }
    /* on_objects, geometry.py:821-854 */
    float on = 1.0f;
    for (int i = 0; i < k; ++i) {
        const wall_t* w = &W[cand[i]];
        float ox_ = px[i + 1] - w->ox, oy_ = py[i + 1] - w->oy;
        float sq = w->tx_ * w->tx_ + w->ty_ * w->ty_;
        if (sq == 0.0f) sq = 1.0f;
        float s = (w->tx_ * ox_ + w->ty_ * oy_) / sq;
        float c = t_and(t_ge(s, 0.0f, p), t_le(s, 1.0f, p), p->approx);
        on = t_and(on, c, p->approx);
    }
    int skip_rest = p->prune && (on == 0.0f); /* valid is exactly 0 whatever follows (NaNs aside, see below) */
    /* path loss, geometry.py:1077-1084 / 641-650 */
    if (k > 0) {
        for (int i = 0; i < k; ++i) {
            const wall_t* w = &W[cand[i]];
            float ix, iy, rx_, ry_;
            normalize2(px[i + 1] - px[i], py[i + 1] - py[i], &ix, &iy);
            normalize2(px[i + 2] - px[i + 1], py[i + 2] - py[i + 1], &rx_, &ry_);
            float din = ix * w->nx + iy * w->ny;
            float ex = rx_ - (ix - 2.0f * din * w->nx);
            float ey = ry_ - (iy - 2.0f * din * w->ny);
            loss = loss + (ex * ex + ey * ey);
        }
    }
    /* intersects_with_objects, geometry.py:856-906 */
    float hit = 0.0f;
    if (!skip_rest) {
        for (int i = 0; i <= k; ++i) {
            int ig0 = (i == 0) ? -1 : cand[i - 1];
            int ig1 = (i == k) ? -1 : cand[i];
            for (int j = 0; j < N; ++j) {
                if (j == ig0 || j == ig1) continue;
                hit = t_or(hit, wall_hits(&W[j], px[i], py[i], px[i + 1], py[i + 1], p), p->approx);
                /* prune level 2: an occluder whose test is exactly True / saturated to 1 settles not(intersects) = 0 and
                 * with it valid = 0: whatever the remaining tests return -- a larger value does not exist, a NaN ends in
                 * nan_to_num(NaN) = 0 as well -- so they are not evaluated */
                if (p->prune >= 2 && hit == 1.0f) { i = k; break; }
            }
        }
    }
    float ok = t_lt(loss, p->tol, p);
    float valid = t_and(t_and(on, t_not(hit, p->approx), p->approx), ok, p->approx);
    /* jnp.nan_to_num */
    if (valid != valid) valid = 0.0f;
    /* path function */
    float r = 0.0f;
    for (int i = 0; i <= k; ++i) {
        float vx = (px[i + 1] - px[i]) + 1.1920929e-07f;
        float vy = (py[i + 1] - py[i]) + 1.1920929e-07f;
        r = r + sqrtf(vx * vx + vy * vy);
    }
    float f;
    switch (p->fun_id) {
        case 0: f = ipow(p->r_coef, k) / (p->height * p->height + r * r); break;
        case 1: f = r * r; break;
        case 2: f = r; break;
        default: f = 1.0f; break;
    }
    *valid_out = valid;
    *fun_out = f;
}

typedef struct {
    int k;
    int idx[ORC_MAX_ORDER];
    float img[2 * ORC_MAX_ORDER];
} cand_t;

/* Enumerate candidates of order k lexicographically into a flat array (scene.py:122-175). */
static long enum_candidates(int N, const uint8_t* allowed, int k, const wall_t* W, float txx, float txy,
                            cand_t* out /* may be NULL: count only */) {
    long count = 0;
    int idx[ORC_MAX_ORDER];
    if (k == 0) {
        if (out) { out[0].k = 0; }
        return 1;
    }
    /* iterative odometer */
    int depth = 0;
    idx[0] = -1;
    while (depth >= 0) {
        int w = idx[depth] + 1;
        while (w < N && ((allowed && !allowed[w]) || (depth > 0 && idx[depth - 1] == w))) ++w;
        if (w >= N) { --depth; continue; }
        idx[depth] = w;
        if (depth == k - 1) {
            if (out) {
                cand_t* c = &out[count];
                c->k = k;
                float ix = txx, iy = txy;
                for (int i = 0; i < k; ++i) {
                    c->idx[i] = idx[i];
                    /* image_of, geometry.py:652-670 */
                    const wall_t* ww = &W[idx[i]];
                    float dxp = ix - ww->ox, dyp = iy - ww->oy;
                    float dn = dxp * ww->nx + dyp * ww->ny;
                    ix = ix - 2.0f * dn * ww->nx;
                    iy = iy - 2.0f * dn * ww->ny;
                    c->img[2 * i] = ix; c->img[2 * i + 1] = iy;
                }
            }
            ++count;
        } else {
            ++depth;
            idx[depth] = -1;
        }
    }
    return count;
}

long orc_num_candidates(int N, const uint8_t* allowed, int min_order, int max_order) {
    long c = 0;
    for (int k = min_order; k <= max_order; ++k) c += enum_candidates(N, allowed, k, NULL, 0, 0, NULL);
    return c;
}

/*
 * Power map for one transmitter.  walls: [N][2][2]; allowed: [N] or NULL (filter_objects);
 * X, Y: [ncell]; out: [ncell].  Returns 0, or <0 on bad arguments.
 */
int orc_power_map_ex(const float* walls, int N, const uint8_t* allowed, const orc_params* p, const float* tx,
                     const float* X, const float* Y, long ncell, float* out, float* out_count, int nthreads) {
    if (N < 0 || p->max_order > ORC_MAX_ORDER || p->min_order < 0) return -1;
    wall_t* W = (wall_t*)malloc(sizeof(wall_t) * (N > 0 ? N : 1));
    for (int j = 0; j < N; ++j) make_wall(&W[j], walls + 4 * j, p->patch);
    long total = orc_num_candidates(N, allowed, p->min_order, p->max_order);
    cand_t* C = (cand_t*)malloc(sizeof(cand_t) * (total > 0 ? total : 1));
    long off = 0;
    for (int k = p->min_order; k <= p->max_order; ++k) off += enum_candidates(N, allowed, k, W, tx[0], tx[1], C + off);
#ifdef _OPENMP
    /* (a num_threads clause, not omp_set_num_threads: the count must not stick to later calls that ask for the default) */
    const int nt_ = nthreads > 0 ? nthreads : omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt_)
    for (long c = 0; c < ncell; ++c) {
        float acc = 0.0f;
        float cnt = 0.0f; /* the same sweep with fun = 1.0: the map of (soft) valid-path counts */
        for (long ci = 0; ci < total; ++ci) {
            float valid, f;
            if (!p->grid_is_tx) {
                eval_candidate(W, N, C[ci].idx, C[ci].k, C[ci].img, tx[0], tx[1], X[c], Y[c], p, &valid, &f);
            } else {
                /* the cell is the transmitter: its images are per cell (geometry.py:1086-1091) */
                float img[2 * ORC_MAX_ORDER];
                float ix = X[c], iy = Y[c];
                for (int i = 0; i < C[ci].k; ++i) {
                    const wall_t* ww = &W[C[ci].idx[i]];
                    float dxp = ix - ww->ox, dyp = iy - ww->oy;
                    float dn = dxp * ww->nx + dyp * ww->ny;
                    ix = ix - 2.0f * dn * ww->nx;
                    iy = iy - 2.0f * dn * ww->ny;
                    img[2 * i] = ix; img[2 * i + 1] = iy;
                }
                eval_candidate(W, N, C[ci].idx, C[ci].k, img, X[c], Y[c], tx[0], tx[1], p, &valid, &f);
            }
            acc = acc + valid * f;
            cnt = cnt + valid * 1.0f;
        }
        out[c] = acc;
        if (out_count) out_count[c] = cnt;
    }
    free(C);
    free(W);
    return 0;
}

int orc_power_map(const float* walls, int N, const uint8_t* allowed, const orc_params* p, const float* tx,
                  const float* X, const float* Y, long ncell, float* out, int nthreads) {
    return orc_power_map_ex(walls, N, allowed, p, tx, X, Y, ncell, out, NULL, nthreads);
}

/* Per-candidate dump for a single RX (debug / known-answer tests): valid[C], fun[C]. */
int orc_eval_candidates(const float* walls, int N, const uint8_t* allowed, const orc_params* p, const float* tx,
                        const float* rx, float* valid, float* fun, int32_t* cand_idx /*[C][ORC_MAX_ORDER]*/) {
    wall_t* W = (wall_t*)malloc(sizeof(wall_t) * (N > 0 ? N : 1));
    for (int j = 0; j < N; ++j) make_wall(&W[j], walls + 4 * j, p->patch);
    long total = orc_num_candidates(N, allowed, p->min_order, p->max_order);
    cand_t* C = (cand_t*)malloc(sizeof(cand_t) * (total > 0 ? total : 1));
    long off = 0;
    for (int k = p->min_order; k <= p->max_order; ++k) off += enum_candidates(N, allowed, k, W, tx[0], tx[1], C + off);
    for (long ci = 0; ci < total; ++ci) {
        eval_candidate(W, N, C[ci].idx, C[ci].k, C[ci].img, tx[0], tx[1], rx[0], rx[1], p, &valid[ci], &fun[ci]);
        if (cand_idx)
            for (int i = 0; i < ORC_MAX_ORDER; ++i) cand_idx[ci * ORC_MAX_ORDER + i] = (i < C[ci].k) ? C[ci].idx[i] : -1;
    }
    free(C);
    free(W);
    return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
