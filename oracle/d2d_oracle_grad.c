/*
 * CPU ORACLE (C), gradients -- test infrastructure only, never shipped, never on the product path.
 *
 * Per-cell gradient of DiffeRT2d v0.4.0's power-map sweep, d facc / d cell (scene.py:1920-1923: jax.grad(facc, argnums=1)
 * for a receiver grid, argnums=0 for a transmitter grid, scene.py:1617-1620), by FORWARD-mode dual numbers carried through
 * the very op chain of oracle/d2d_oracle.c (and oracle/ref.py, which holds the line-by-line citations):
 *
 *   value    fp32, one rounding per operation, the reference's order -- every branch (where, min / max selection, the
 *            saturation of an activation) is decided on the reference's own fp32 numbers, bit for bit d2d_oracle.c's;
 *   tangent  two doubles (d / d cell.x, d / d cell.y), each operation's exact derivative at its fp32 operands.
 *
 * An independent checker of the GPU's hand-derived REVERSE-mode kernels: no adjoint is written here, nothing is shared
 * with differt2d_amd/csrc, and it is cheap enough for whole rows of BASELINE configs[2] (65 536 cells x 2 501 candidates).
 *
 * Reverse mode and forward mode agree wherever every primitive has a derivative -- including the tie rule of
 * jnp.minimum / maximum (each argument gets half: a linear map, the same in both modes) and lax.logistic's JVP rule.
 * They differ where the reference's reverse mode sends a ZERO cotangent into an infinite local derivative (0 * inf = NaN),
 * which a forward tangent never sees because jnp.where drops the untaken branch.  The reference has exactly two such
 * traps on this path, and both are stated here as explicit rules (oracle/ref.py reproduces them by running the same
 * chain under torch.autograd; tests/test_oracle_grad_c.py holds this file against that, NaN positions included):
 *   (1) jnp.where(un == 0, 0, vn * u / un), geometry.py:1105: un == 0 in any step of the backward scan of any candidate
 *       makes the cell's gradient NaN, valid or not -- provided the candidate's contribution is differentiated at all
 *       (hard mode with a `fun` that ignores the path, fun = 1: valid is a bool, nothing reaches the interaction points);
 *   (2) normalize() of a zero-length vector inside the loss, geometry.py:227-228 (sqrt'(0) behind a where): approx modes
 *       only (in hard mode the loss feeds a comparison, which has no derivative).
 *
 * Build: oracle/Makefile (same flags as d2d_oracle.c; -ffp-contract=off matters for the value part).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_ORDER 4

typedef struct orc_params { /* = d2d_oracle.c */
    int32_t min_order, max_order;
    int32_t approx;
    int32_t act;
    float alpha;
    float tol;
    float patch;
    float seg_tol;
    int32_t fun_id;
    float r_coef, height;
    int32_t prune;
    int32_t grid_is_tx;
} orc_params;

typedef struct {
    float v;      /* the reference's fp32 value */
    double x, y;  /* d / d cell.x, d / d cell.y */
} dual;

static inline dual dc(float v) { dual r = {v, 0.0, 0.0}; return r; }
static inline dual dadd(dual a, dual b) { dual r = {a.v + b.v, a.x + b.x, a.y + b.y}; return r; }
static inline dual dsub(dual a, dual b) { dual r = {a.v - b.v, a.x - b.x, a.y - b.y}; return r; }
static inline dual dmul(dual a, dual b) {
    dual r = {a.v * b.v, a.x * (double)b.v + (double)a.v * b.x, a.y * (double)b.v + (double)a.v * b.y};
    return r;
}
static inline dual dmulc(float c, dual a) { dual r = {c * a.v, (double)c * a.x, (double)c * a.y}; return r; }
static inline dual ddiv(dual a, dual b) {
    const double q = (double)a.v / (double)b.v;
    dual r = {a.v / b.v, (a.x - q * b.x) / (double)b.v, (a.y - q * b.y) / (double)b.v};
    return r;
}
static inline dual dsqrt(dual a) {
    const double s = sqrt((double)a.v);
    dual r = {sqrtf(a.v), a.x / (2.0 * s), a.y / (2.0 * s)};
    return r;
}
/* jnp.minimum / jnp.maximum: NaN-propagating; the JVP gives each argument half at a tie.  g_kink (per thread): a tie between
 * arguments with DIFFERENT tangents was met -- the cell sits on a kink of the map, where the derivative is a convention (JAX's:
 * the mean) and where an evaluation with any other rounding may see no tie at all. */
static _Thread_local int g_kink;
/* g_site: where the next minimum / maximum sits -- bit of g_kink it sets when it ties with different tangents: 0 inside an
 * activation (relu6), 1 on_objects' ge / le pair, 2 on_objects' fold over the walls, 3 a segment test's ge / le pair, 4 the pair of
 * a wall test's two quotients, 5 the fold of the occlusion tests, 6 valid = all(on, not hit, ok) */
static _Thread_local int g_site;
/* Diagnostic (environment ORC_TIE_FIRST = bit mask of sites): at those sites a tie sends the whole cotangent to the FIRST
 * argument instead of splitting it -- what an adjoint without a tie rule computes; used to attribute a deviation to a site. */
static int g_tie_first = 0;
/* g_amp (per thread, per candidate): the largest |u| / |u.n| met in the backward scan -- how much a step amplifies the
 * rounding of the point it starts from (a pole of the image method nearby: every fp32 evaluation order gets its own digits) */
static _Thread_local double g_amp;
static inline dual dmin(dual a, dual b) {
    if (a.v != a.v || b.v != b.v) { dual r = {NAN, NAN, NAN}; return r; }
    if (a.v < b.v) return a;
    if (b.v < a.v) return b;
    if (a.x != b.x || a.y != b.y) g_kink |= 1 << g_site;
    if ((g_tie_first >> g_site) & 1) return a;
    dual r = {a.v, 0.5 * (a.x + b.x), 0.5 * (a.y + b.y)};
    return r;
}
static inline dual dmax(dual a, dual b) {
    if (a.v != a.v || b.v != b.v) { dual r = {NAN, NAN, NAN}; return r; }
    if (a.v > b.v) return a;
    if (b.v > a.v) return b;
    if (a.x != b.x || a.y != b.y) g_kink |= 1 << g_site;
    if ((g_tie_first >> g_site) & 1) return a;
    dual r = {a.v, 0.5 * (a.x + b.x), 0.5 * (a.y + b.y)};
    return r;
}

/* logic.py:218-267 */
static inline dual activation(dual x, const orc_params* p) {
    dual z = dmulc(p->alpha, x);
    if (p->act == 0) { /* jax.nn.hard_sigmoid: relu6(z + 3) / 6, relu6 = minimum(maximum(., 0), 6) */
        const int site = g_site;
        g_site = 0;
        dual r = ddiv(dmin(dmax(dadd(z, dc(3.0f)), dc(0.0f)), dc(6.0f)), dc(6.0f));
        g_site = site;
        return r;
    }
    /* jax.nn.sigmoid = lax.logistic: value 1 / (1 + exp(-z)), JVP y (1 - y) */
    const float y = 1.0f / (1.0f + expf(-z.v));
    const double g = (double)y * (1.0 - (double)y);
    dual r = {y, g * z.x, g * z.y};
    return r;
}

/* Truthy values.  Hard mode: 0.0f / 1.0f standing for False / True, no derivative. */
static inline dual t_and(dual a, dual b, int approx) { return approx ? dmin(a, b) : dc((a.v != 0.0f && b.v != 0.0f) ? 1.0f : 0.0f); }
static inline dual t_or(dual a, dual b, int approx) { return approx ? dmax(a, b) : dc((a.v != 0.0f || b.v != 0.0f) ? 1.0f : 0.0f); }
static inline dual t_not(dual a, int approx) { return approx ? dsub(dc(1.0f), a) : dc(a.v != 0.0f ? 0.0f : 1.0f); }
static inline dual t_ge(dual x, dual y, const orc_params* p) { return p->approx ? activation(dsub(x, y), p) : dc(x.v >= y.v ? 1.0f : 0.0f); }
static inline dual t_le(dual x, dual y, const orc_params* p) { return p->approx ? activation(dsub(y, x), p) : dc(x.v <= y.v ? 1.0f : 0.0f); }
static inline dual t_lt(dual x, dual y, const orc_params* p) { return p->approx ? activation(dsub(y, x), p) : dc(x.v < y.v ? 1.0f : 0.0f); }

typedef struct {
    float ox, oy, dx, dy, tx_, ty_, nx, ny, p1x, p1y, p2x, p2y;
} wall_t;

static void make_wall(wall_t* w, const float* xy, float patch) { /* = d2d_oracle.c */
    w->ox = xy[0]; w->oy = xy[1]; w->dx = xy[2]; w->dy = xy[3];
    w->tx_ = w->dx - w->ox; w->ty_ = w->dy - w->oy;
    float vx = w->ty_, vy = -w->tx_;
    float len = sqrtf(vx * vx + vy * vy);
    if (len == 0.0f) len = 1.0f;
    w->nx = vx / len; w->ny = vy / len;
    w->p1x = w->ox - patch * w->tx_; w->p1y = w->oy - patch * w->ty_;
    w->p2x = w->dx + patch * w->tx_; w->p2y = w->dy + patch * w->ty_;
}

/* geometry.py:206-230; *zero: the vector has length exactly 0 (rule 2 of the header) */
static inline void normalize2(dual vx, dual vy, dual* ox, dual* oy, int* zero) {
    dual len = dsqrt(dadd(dmul(vx, vx), dmul(vy, vy)));
    if (len.v == 0.0f) { len = dc(1.0f); *zero = 1; } /* where(length == 0, 1, length) */
    *ox = ddiv(vx, len);
    *oy = ddiv(vy, len);
}

/* geometry.py:163-171 */
static inline dual seg_test(dual num, dual den, const orc_params* p) {
    const int den_is_zero = (den.v == 0.0f);
    dual dd = den_is_zero ? dc(1.0f) : den;
    dual t = den_is_zero ? dc(INFINITY) : ddiv(num, dd);
    dual ge = t_ge(t, dc(-p->seg_tol), p), le = t_le(t, dc(1.0f + p->seg_tol), p);
    g_site = 3;
    return t_and(ge, le, p->approx);
}

/* geometry.py:82-173 with P1, P2 = the patched wall (constants here), P3, P4 = a path segment */
static inline dual wall_hits(const wall_t* w, dual p3x, dual p3y, dual p4x, dual p4y, const orc_params* p) {
    const float Ax = w->p2x - w->p1x, Ay = w->p2y - w->p1y;
    dual Bx = dsub(p3x, p4x), By = dsub(p3y, p4y);
    dual Cx = dsub(dc(w->p1x), p3x), Cy = dsub(dc(w->p1y), p3y);
    dual a = dsub(dmul(By, Cx), dmul(Bx, Cy));
    dual b = dsub(dmulc(Ax, Cy), dmulc(Ay, Cx));
    dual d = dsub(dmulc(Ay, Bx), dmulc(Ax, By));
    dual sa = seg_test(a, d, p), sb = seg_test(b, d, p);
    g_site = 4;
    return t_and(sa, sb, p->approx);
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

/* geometry.py:652-670 */
static inline void image_of(const wall_t* w, dual px, dual py, dual* ox, dual* oy) {
    dual ix = dsub(px, dc(w->ox)), iy = dsub(py, dc(w->oy));
    dual dn = dadd(dmulc(w->nx, ix), dmulc(w->ny, iy));
    dual s = dmulc(2.0f, dn);
    *ox = dsub(px, dmulc(w->nx, s));
    *oy = dsub(py, dmulc(w->ny, s));
}

/* One (cell, candidate): contribution valid * fun as a dual number; *poison: the reference's reverse mode yields NaN. */
static dual eval_candidate(const wall_t* W, int N, const int* cand, int k, dual txx, dual txy, dual rxx, dual rxy,
                           const orc_params* p, int* poison) {
    dual px[ORC_MAX_ORDER + 2], py[ORC_MAX_ORDER + 2];
    /* hard validity is a bool: the contribution is differentiated through `fun` alone, and fun = 1 ignores the path */
    const int differentiated = p->approx || p->fun_id != 3;
    px[0] = txx; py[0] = txy; px[k + 1] = rxx; py[k + 1] = rxy;
    if (k > 0) {
        /* forward scan of images, geometry.py:1086-1091 */
        dual imx[ORC_MAX_ORDER], imy[ORC_MAX_ORDER];
        dual ix = txx, iy = txy;
        for (int i = 0; i < k; ++i) {
            image_of(&W[cand[i]], ix, iy, &ix, &iy);
            imx[i] = ix; imy[i] = iy;
        }
        /* backward scan, geometry.py:1093-1110 */
        dual ptx = rxx, pty = rxy;
        for (int i = k - 1; i >= 0; --i) {
            const wall_t* w = &W[cand[i]];
            dual ux = dsub(ptx, imx[i]), uy = dsub(pty, imy[i]);
            dual vx = dsub(dc(w->ox), ptx), vy = dsub(dc(w->oy), pty);
            dual un = dadd(dmulc(w->nx, ux), dmulc(w->ny, uy));
            dual vn = dadd(dmulc(w->nx, vx), dmulc(w->ny, vy));
            {
                const double am = (fabs((double)ux.v) + fabs((double)uy.v)) / fabs((double)un.v);
                if (!(am <= g_amp)) g_amp = am; /* (NaN and inf included) */
            }
            dual incx, incy;
            if (un.v == 0.0f) {
                incx = dc(0.0f); incy = dc(0.0f);
                if (differentiated) *poison = 1; /* rule (1) */
            } else {
                incx = ddiv(dmul(vn, ux), un);
                incy = ddiv(dmul(vn, uy), un);
            }
            ptx = dadd(ptx, incx); pty = dadd(pty, incy);
            px[i + 1] = ptx; py[i + 1] = pty;
        }
    }
    /* on_objects, geometry.py:821-854 */
    dual on = dc(1.0f);
    for (int i = 0; i < k; ++i) {
        const wall_t* w = &W[cand[i]];
        dual ox_ = dsub(px[i + 1], dc(w->ox)), oy_ = dsub(py[i + 1], dc(w->oy));
        float sq = w->tx_ * w->tx_ + w->ty_ * w->ty_;
        if (sq == 0.0f) sq = 1.0f;
        dual s = ddiv(dadd(dmulc(w->tx_, ox_), dmulc(w->ty_, oy_)), dc(sq));
        dual ge = t_ge(s, dc(0.0f), p), le = t_le(s, dc(1.0f), p);
        g_site = 1;
        dual c = t_and(ge, le, p->approx);
        g_site = 2;
        on = t_and(on, c, p->approx);
    }
    /* path loss, geometry.py:1077-1084 / 641-650 */
    dual loss = dc(0.0f);
    for (int i = 0; i < k; ++i) {
        const wall_t* w = &W[cand[i]];
        dual ix, iy, rx_, ry_;
        int zero = 0;
        normalize2(dsub(px[i + 1], px[i]), dsub(py[i + 1], py[i]), &ix, &iy, &zero);
        normalize2(dsub(px[i + 2], px[i + 1]), dsub(py[i + 2], py[i + 1]), &rx_, &ry_, &zero);
        if (zero && p->approx) *poison = 1; /* rule (2) */
        dual din = dadd(dmulc(w->nx, ix), dmulc(w->ny, iy));
        dual ex = dsub(rx_, dsub(ix, dmulc(w->nx, dmulc(2.0f, din))));
        dual ey = dsub(ry_, dsub(iy, dmulc(w->ny, dmulc(2.0f, din))));
        loss = dadd(loss, dadd(dmul(ex, ex), dmul(ey, ey)));
    }
    /* Exact shortcut (orc_params.prune): on_objects exactly 0 with a zero tangent makes valid = min(0, ..) = 0 with a zero
     * tangent whatever the occlusion tests return (they are >= 0; a NaN among them ends in nan_to_num(NaN) = 0 as well);
     * likewise an occluder saturated to exactly 1 with a zero tangent.  Only a tie at 0 between on_objects and a term whose
     * own tangent is non-zero AT exactly 0 could tell the difference: tests/test_oracle_grad_c.py compares the two levels. */
    const int skip = p->prune && on.v == 0.0f && on.x == 0.0 && on.y == 0.0;
    /* intersects_with_objects, geometry.py:856-906 */
    dual hit = dc(0.0f);
    if (!skip) {
        for (int i = 0; i <= k; ++i) {
            const int ig0 = (i == 0) ? -1 : cand[i - 1];
            const int ig1 = (i == k) ? -1 : cand[i];
            for (int j = 0; j < N; ++j) {
                if (j == ig0 || j == ig1) continue;
                dual wh = wall_hits(&W[j], px[i], py[i], px[i + 1], py[i + 1], p);
                g_site = 5;
                hit = t_or(hit, wh, p->approx);
                if (p->prune && hit.v == 1.0f && hit.x == 0.0 && hit.y == 0.0) { i = k; break; }
            }
        }
    }
    dual ok = t_lt(loss, dc(p->tol), p);
    g_site = 6;
    dual valid = skip ? dc(0.0f) : t_and(t_and(on, t_not(hit, p->approx), p->approx), ok, p->approx);
    if (valid.v != valid.v) valid = dc(0.0f); /* jnp.nan_to_num: value 0, nothing flows back */
    /* path function: geometry.py:176-203, utils.py:17-54 */
    dual r = dc(0.0f);
    for (int i = 0; i <= k; ++i) {
        dual vx = dadd(dsub(px[i + 1], px[i]), dc(1.1920929e-07f));
        dual vy = dadd(dsub(py[i + 1], py[i]), dc(1.1920929e-07f));
        r = dadd(r, dsqrt(dadd(dmul(vx, vx), dmul(vy, vy))));
    }
    dual f;
    switch (p->fun_id) {
        case 0: f = ddiv(dc(ipow(p->r_coef, k)), dadd(dc(p->height * p->height), dmul(r, r))); break;
        case 1: f = dmul(r, r); break;
        case 2: f = r; break;
        default: f = dc(1.0f); break;
    }
    return dmul(valid, f);
}

typedef struct {
    int k;
    int idx[ORC_MAX_ORDER];
} cand_t;

/* scene.py:122-175: lexicographic, no equal neighbours (= d2d_oracle.c) */
static long enum_candidates(int N, const uint8_t* allowed, int k, cand_t* out) {
    long count = 0;
    int idx[ORC_MAX_ORDER];
    if (k == 0) {
        if (out) out[0].k = 0;
        return 1;
    }
    int depth = 0;
    idx[0] = -1;
    while (depth >= 0) {
        int w = idx[depth] + 1;
        while (w < N && ((allowed && !allowed[w]) || (depth > 0 && idx[depth - 1] == w))) ++w;
        if (w >= N) { --depth; continue; }
        idx[depth] = w;
        if (depth == k - 1) {
            if (out) {
                out[count].k = k;
                for (int i = 0; i < k; ++i) out[count].idx[i] = idx[i];
            }
            ++count;
        } else {
            ++depth;
            idx[depth] = -1;
        }
    }
    return count;
}

/*
 * Value map and per-cell gradient for one fixed end point.  grad: [ncell][2] doubles = d facc / d (cell.x, cell.y), NaN
 * where the reference's reverse mode yields NaN.  value: [ncell] floats (bit for bit orc_power_map's).  gabs (may be NULL):
 * [ncell] sum over the candidates of |d contribution / d cell.x| + |d contribution / d cell.y| -- the magnitude an fp32
 * evaluation's rounding scales with (a gradient that is a small difference of large contributions cannot be held to a
 * relative bar of its own size).  kink (may be NULL): [ncell] 1 where some minimum / maximum tied between arguments with
 * different tangents (a bit mask of where: see g_site).  amp (may be NULL): [ncell] the largest |u| / |u.n| of the backward scans of the candidates whose
 * contribution has a non-zero tangent (how ill-conditioned the cell's gradient is: a pole of the image method nearby).
 */
int orc_power_map_grad(const float* walls, int N, const uint8_t* allowed, const orc_params* p, const float* tx, const float* X,
                       const float* Y, long ncell, float* value, double* grad, double* gabs, uint8_t* kink, double* amp,
                       int nthreads) {
    if (N < 0 || p->max_order > ORC_MAX_ORDER || p->min_order < 0) return -1;
    g_tie_first = getenv("ORC_TIE_FIRST") ? atoi(getenv("ORC_TIE_FIRST")) : 0;
    wall_t* W = (wall_t*)malloc(sizeof(wall_t) * (N > 0 ? N : 1));
    for (int j = 0; j < N; ++j) make_wall(&W[j], walls + 4 * j, p->patch);
    long total = 0;
    for (int k = p->min_order; k <= p->max_order; ++k) total += enum_candidates(N, allowed, k, NULL);
    cand_t* C = (cand_t*)malloc(sizeof(cand_t) * (total > 0 ? total : 1));
    long off = 0;
    for (int k = p->min_order; k <= p->max_order; ++k) off += enum_candidates(N, allowed, k, C + off);
#ifdef _OPENMP
    /* (a num_threads clause, not omp_set_num_threads: the count must not stick to later calls that ask for the default) */
    const int nt_ = nthreads > 0 ? nthreads : omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt_)
    for (long c = 0; c < ncell; ++c) {
        dual acc = dc(0.0f);
        double ga = 0.0, cell_amp = 0.0;
        int poison = 0;
        g_kink = 0;
        dual cx = {X[c], 1.0, 0.0}, cy = {Y[c], 0.0, 1.0};
        dual fx = dc(tx[0]), fy = dc(tx[1]);
        for (long ci = 0; ci < total; ++ci) {
            g_amp = 0.0;
            dual t = p->grid_is_tx ? eval_candidate(W, N, C[ci].idx, C[ci].k, cx, cy, fx, fy, p, &poison)
                                   : eval_candidate(W, N, C[ci].idx, C[ci].k, fx, fy, cx, cy, p, &poison);
            acc = dadd(acc, t); /* scene.py:1909 */
            ga += fabs(t.x) + fabs(t.y);
            if ((t.x != 0.0 || t.y != 0.0) && !(g_amp <= cell_amp)) cell_amp = g_amp;
        }
        if (amp) amp[c] = cell_amp;
        if (gabs) gabs[c] = ga;
        if (kink) kink[c] = (uint8_t)g_kink;
        value[c] = acc.v;
        grad[2 * c] = poison ? (double)NAN : acc.x;
        grad[2 * c + 1] = poison ? (double)NAN : acc.y;
    }
    free(C);
    free(W);
    return 0;
}
