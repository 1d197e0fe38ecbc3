/*
 * CPU ORACLE (C), MinPath / FermatPath sweeps -- test infrastructure only, never shipped, never on the product path.
 *
 * DiffeRT2d v0.4.0's power-map sweep with the optimiser-based path classes over a scene of Wall / RIS / Vertex objects
 * (BASELINE.json configs[4]), restated in plain C so that the GPU's solver kernels (K4 of DESIGN.md) have a checker that is
 * fast enough for whole maps and for fuzzing -- oracle/ref.py, which holds the line-by-line citations, runs the same chain under
 * NumPy / torch at a few cells per second.  Reference lines followed (paths relative to the DiffeRT2d checkout):
 *   parametric_to_cartesian     differt2d/geometry.py:988-1010; Wall :581-587; Vertex :381-385
 *   objective, MinPath          differt2d/geometry.py:1207-1288: sum of evaluate_cartesian -- Wall :641-650, RIS :698-711,
 *                               Vertex :416-419 (0); recorded loss = the objective BEFORE the last update (:1284-1288)
 *   objective, FermatPath       differt2d/geometry.py:1117-1204: path_length (:176-203); recorded loss = sum of
 *                               evaluate_cartesian at the final points (:1204)
 *   minimize                    differt2d/optimize.py:44-97: `steps` x { loss, g = value_and_grad(objective)(theta);
 *                               theta += optax.adam(0.1) update }, optax 0.2.4's scale_by_adam, in oracle/ref.py:616-638's order
 *   validity, path function     as oracle/d2d_oracle.c (geometry.py:821-963, logic.py:218-537, utils.py:17-54); a Vertex
 *                               contains every point and intersects nothing (:397-414)
 *   accumulation                differt2d/scene.py:1892-1918; the initial guesses theta0[candidate] are shared by all cells
 *                               (:1887-1890) and are an INPUT here (the reference draws them from its Threefry key)
 *
 * The derivative of the objective w.r.t. theta -- jax.value_and_grad in the reference, a hand-derived gradient in the kernels
 * -- is taken here by FORWARD-mode dual numbers: the value in the working precision with one rounding per operation in the
 * reference's order, the tangents in double; g = the exact derivative of that chain, rounded to the working precision.  No
 * adjoint code, nothing shared with differt2d_amd/csrc.  (Bit-for-bit agreement with ANY reverse-mode evaluation of g is not
 * defined -- a backward pass rounds in its own order -- so the solver's trajectory is compared within a tolerance, on cells
 * the oracle itself calls well conditioned; everything that does not involve g -- the objective's values, the Adam update
 * given g, validity, path function -- is ref.py's bit for bit, tests/test_oracle_opt_c.py.)
 *
 * Two precisions are instantiated from one body: _f32 (the reference's) and _f64 (the conditioning mask: a cell where the
 * fp32 run, the fp32 run from inputs one ulp away and the fp64 run disagree is ill conditioned for every fp32 evaluation).
 *
 * With `grad` the per-cell gradient d facc / d cell is carried along too (scene.py:1920-1923 through lax.scan's reverse mode
 * in the reference): second-order forward jets of the objective in (theta, cell) give d g / d cell = H_theta,cell +
 * H_theta,theta d theta / d cell inside the loop; first-order duals afterwards.  The reference's reverse-mode NaN conventions
 * that forward mode does not show by itself are stated as rules: normalize() of a zero-length vector inside a differentiated
 * objective / loss (geometry.py:227-228 behind a where) and sqrt'(0) of Adam's second moment when g == 0 exactly.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#ifndef ORC_OPT_INSTANCE

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* (a jet holds NV slots, J_nv of them live: the others are never read) */
#pragma GCC diagnostic ignored "-Wmaybe-uninitialized"

#define ORC_MAX_ORDER 4
#define ORC_WALL 0
#define ORC_RIS 1
#define ORC_VERTEX 2

typedef struct orc_opt_params {
    int32_t approx;   /* 0: jnp.logical_*, 1: min / max / activation */
    int32_t act;      /* 0: hard_sigmoid, 1: sigmoid */
    int32_t fun_id;   /* 0 received_power, 1 length**2, 2 length, 3 one */
    double alpha, tol, patch, seg_tol, r_coef, height; /* python floats in the reference: cast to the working precision (xp.c) */
    int32_t solver;   /* 1: MinPath, 2: FermatPath */
    int32_t steps;    /* >= 1 */
    double lr, b1, b2, eps; /* optax.adam(0.1): 0.1, 0.9, 0.999, 1e-8 */
    int32_t grid_is_tx;
    int32_t g_ulps; /* conditioning probe: every objective gradient moved by this many units in the last place (0: as computed) --
                       what another implementation of the same derivative (a reverse pass, a hand-derived formula) differs by */
} orc_opt_params;

#define ORC_OPT_INSTANCE
/* five instances of the body below: {fp32, fp64} x {lean: first-order jets in theta only (value maps), full: second-order
 * jets in (theta, cell) (value + per-cell gradient)}, and the full fp32 one again with its DERIVATIVES in fp32 as well (TREAL =
 * float: what an fp32 derivative arithmetic -- any fp32 autodiff, forward or reverse -- loses against the exact derivative of
 * the fp32 chain; the yardstick of the gradient comparisons) */
#define REAL float
#define TREAL double
#define SQRT sqrtf
#define EXP expf
#define REAL_EPS 1.1920929e-07f
#define SFX(name) name##_f32
#define JSECOND 0
#include "d2d_oracle_opt.c"
#undef SFX
#undef JSECOND
#define SFX(name) name##_f32g
#define JSECOND 1
#include "d2d_oracle_opt.c"
#undef SFX
#undef TREAL
#define TREAL float
#define SFX(name) name##_f32t
#include "d2d_oracle_opt.c"
#undef SFX
#undef JSECOND
#undef TREAL
#undef REAL
#undef SQRT
#undef EXP
#undef REAL_EPS
#define REAL double
#define TREAL double
#define SQRT sqrt
#define EXP exp
#define REAL_EPS 2.220446049250313e-16
#define SFX(name) name##_f64
#define JSECOND 0
#include "d2d_oracle_opt.c"
#undef SFX
#undef JSECOND
#define SFX(name) name##_f64g
#define JSECOND 1
#include "d2d_oracle_opt.c"

/*
 * xys [N][2][2] (a Vertex keeps its point in row 0), kind [N], sincos [N][2] = sin(phi), cos(phi) as the caller's backend
 * evaluates them; cands [C][ORC_MAX_ORDER], cand_k [C]; theta0 [C][ORC_MAX_ORDER] (one guess per unknown of the candidate, in
 * order).  f64: 0 fp32, 1 fp64, 2 fp32 with fp32 derivatives (gradient runs).  fixed [2]: the transmitter (receiver for a TX grid).  Outputs: value [ncell] (as double: exact for the fp32 run);
 * grad [ncell][2] or NULL; pts [ncell][C][n_snap][ORC_MAX_ORDER][2] or NULL (the solver's interaction points after snaps[s]
 * updates, 1 <= snaps[s] <= steps) and loss [ncell][C] or NULL (the recorded loss): what the trajectory agreement of the
 * conditioning mask is taken from.
 */
int orc_opt_power_map(int f64, const double* xys, const uint8_t* kind, const double* sincos, int N, const orc_opt_params* p,
                      const double* fixed, const double* X, const double* Y, long ncell, const int32_t* cands,
                      const int32_t* cand_k, long C, const double* theta0, double* value, double* grad, double* pts,
                      double* loss, const int32_t* snaps, int n_snap, int nthreads) {
#define ORC_OPT_ARGS xys, kind, sincos, N, p, fixed, X, Y, ncell, cands, cand_k, C, theta0, value, grad, pts, loss, snaps, n_snap, nthreads
    if (grad) return f64 == 1 ? orc_opt_power_map_f64g(ORC_OPT_ARGS) : (f64 == 2 ? orc_opt_power_map_f32t(ORC_OPT_ARGS) : orc_opt_power_map_f32g(ORC_OPT_ARGS));
    return f64 ? orc_opt_power_map_f64(ORC_OPT_ARGS) : orc_opt_power_map_f32(ORC_OPT_ARGS);
}

/* The objective and its theta-gradient at given parameters (pins the dual-number gradient against autodiff of ref.py) and one
 * Adam update given g (pins the update against ref.adam_minimize bit for bit): see tests/test_oracle_opt_c.py. */
int orc_opt_objective(int f64, const double* xys, const uint8_t* kind, const double* sincos, int N, int solver,
                      const double* tx, const double* rx, const int32_t* cand, int k, const double* theta, double* value,
                      double* g) {
    return f64 ? orc_opt_objective_f64(xys, kind, sincos, N, solver, tx, rx, cand, k, theta, value, g)
               : orc_opt_objective_f32(xys, kind, sincos, N, solver, tx, rx, cand, k, theta, value, g);
}

int orc_opt_adam_step(int f64, const orc_opt_params* p, int t, double g, double* x, double* mu, double* nu) {
    return f64 ? orc_opt_adam_step_f64(p, t, g, x, mu, nu) : orc_opt_adam_step_f32(p, t, g, x, mu, nu);
}

#else /* ------------------------------------------------------------------------------------------------ the body, per precision */

/* Second-order forward jet in up to NV = ORC_MAX_ORDER + 2 variables (theta_0 .. theta_{n-1}, cell.x, cell.y): value in the
 * working precision (one rounding per operation), first and second derivatives in double.  J_nv variables are live; the
 * Hessian only when J_second. */
#if JSECOND
#define NV (ORC_MAX_ORDER + 2)
#else
#define NV ORC_MAX_ORDER
#endif
typedef struct {
    REAL v;
    TREAL g[NV];
#if JSECOND
    TREAL h[NV][NV];
#endif
} SFX(jet);
#define JET SFX(jet)

/* (initial-exec: a plain %fs-relative load -- the default model of a shared object calls __tls_get_addr per access) */
static _Thread_local int SFX(J_nv) __attribute__((tls_model("initial-exec"))), SFX(J_second) __attribute__((tls_model("initial-exec"))),
    SFX(J_poison) __attribute__((tls_model("initial-exec")));
#define J_nv SFX(J_nv)
#define J_second SFX(J_second)
#define J_poison SFX(J_poison)
#if JSECOND
#define IF2(stmt) if (J_second) stmt
#else
#define IF2(stmt)
#endif

static inline JET SFX(jc)(REAL v) {
    JET r;
    r.v = v;
    for (int i = 0; i < J_nv; ++i) {
        r.g[i] = 0.0;
        IF2(for (int j = 0; j < J_nv; ++j) r.h[i][j] = 0.0;)
    }
    return r;
}
static inline JET SFX(jvar)(REAL v, int i) {
    JET r = SFX(jc)(v);
    r.g[i] = 1.0;
    return r;
}
static inline JET SFX(jadd)(JET a, JET b) {
    JET r;
    r.v = a.v + b.v;
    for (int i = 0; i < J_nv; ++i) {
        r.g[i] = a.g[i] + b.g[i];
        IF2(for (int j = 0; j < J_nv; ++j) r.h[i][j] = a.h[i][j] + b.h[i][j];)
    }
    return r;
}
static inline JET SFX(jsub)(JET a, JET b) {
    JET r;
    r.v = a.v - b.v;
    for (int i = 0; i < J_nv; ++i) {
        r.g[i] = a.g[i] - b.g[i];
        IF2(for (int j = 0; j < J_nv; ++j) r.h[i][j] = a.h[i][j] - b.h[i][j];)
    }
    return r;
}
static inline JET SFX(jmul)(JET a, JET b) {
    JET r;
    r.v = a.v * b.v;
    const TREAL av = (TREAL)a.v, bv = (TREAL)b.v;
    for (int i = 0; i < J_nv; ++i) {
        r.g[i] = a.g[i] * bv + av * b.g[i];
        IF2(for (int j = 0; j < J_nv; ++j) r.h[i][j] = a.h[i][j] * bv + a.g[i] * b.g[j] + a.g[j] * b.g[i] + av * b.h[i][j];)
    }
    return r;
}
static inline JET SFX(jmulc)(REAL c, JET a) {
    JET r;
    r.v = c * a.v;
    for (int i = 0; i < J_nv; ++i) {
        r.g[i] = (TREAL)c * a.g[i];
        IF2(for (int j = 0; j < J_nv; ++j) r.h[i][j] = (TREAL)c * a.h[i][j];)
    }
    return r;
}
static inline JET SFX(jdiv)(JET a, JET b) {
    JET r;
    r.v = a.v / b.v;
    const TREAL ib = (TREAL)1 / (TREAL)b.v, q = (TREAL)a.v * ib;
    for (int i = 0; i < J_nv; ++i) r.g[i] = (a.g[i] - q * b.g[i]) * ib;
    /* d2 (a / b) = (a_ij - q_i b_j - q_j b_i - q b_ij) / b */
    IF2(for (int i = 0; i < J_nv; ++i) for (int j = 0; j < J_nv; ++j)
            r.h[i][j] = (a.h[i][j] - r.g[i] * b.g[j] - r.g[j] * b.g[i] - q * b.h[i][j]) * ib;)
    return r;
}
static inline JET SFX(jsqrt)(JET a) {
    JET r;
    r.v = SQRT(a.v);
    const TREAL s = (TREAL)sqrt((double)a.v), i2s = (TREAL)1 / ((TREAL)2 * s);
    for (int i = 0; i < J_nv; ++i) r.g[i] = a.g[i] * i2s;
    /* (a_ij - 2 s_i s_j) / 2s */
    IF2(for (int i = 0; i < J_nv; ++i) for (int j = 0; j < J_nv; ++j) r.h[i][j] = a.h[i][j] * i2s - r.g[i] * r.g[j] / s;)
    return r;
}
#define jc SFX(jc)
#define jvar SFX(jvar)
#define jadd SFX(jadd)
#define jsub SFX(jsub)
#define jmul SFX(jmul)
#define jmulc SFX(jmulc)
#define jdiv SFX(jdiv)
#define jsqrt SFX(jsqrt)

/* jnp.minimum / maximum: NaN-propagating, a tie splits the derivative evenly (first order only: used after the loop) */
static inline JET SFX(jmin)(JET a, JET b) {
    if (a.v != a.v || b.v != b.v) { JET r = jc((REAL)NAN); for (int i = 0; i < J_nv; ++i) r.g[i] = NAN; return r; }
    if (a.v < b.v) return a;
    if (b.v < a.v) return b;
    JET r = a;
    for (int i = 0; i < J_nv; ++i) r.g[i] = (TREAL)0.5 * (a.g[i] + b.g[i]);
    return r;
}
static inline JET SFX(jmax)(JET a, JET b) {
    if (a.v != a.v || b.v != b.v) { JET r = jc((REAL)NAN); for (int i = 0; i < J_nv; ++i) r.g[i] = NAN; return r; }
    if (a.v > b.v) return a;
    if (b.v > a.v) return b;
    JET r = a;
    for (int i = 0; i < J_nv; ++i) r.g[i] = (TREAL)0.5 * (a.g[i] + b.g[i]);
    return r;
}
#define jmin SFX(jmin)
#define jmax SFX(jmax)

typedef struct {
    int kind;
    REAL ox, oy, dx, dy, tx_, ty_, nx, ny, p1x, p1y, p2x, p2y, sinp, cosp;
} SFX(obj_t);
#define OBJ SFX(obj_t)

static void SFX(make_obj)(OBJ* w, int kind, const double* xy, const double* sc, REAL patch) {
    w->kind = kind;
    w->ox = (REAL)xy[0]; w->oy = (REAL)xy[1]; w->dx = (REAL)xy[2]; w->dy = (REAL)xy[3];
    w->tx_ = w->dx - w->ox; w->ty_ = w->dy - w->oy;
    REAL vx = w->ty_, vy = -w->tx_; /* geometry.py:561-573 */
    REAL len = SQRT(vx * vx + vy * vy);
    if (len == (REAL)0) len = (REAL)1;
    w->nx = vx / len; w->ny = vy / len;
    w->p1x = w->ox - patch * w->tx_; w->p1y = w->oy - patch * w->ty_; /* geometry.py:632-636 */
    w->p2x = w->dx + patch * w->tx_; w->p2y = w->dy + patch * w->ty_;
    w->sinp = (REAL)sc[0]; w->cosp = (REAL)sc[1];
}

/* geometry.py:206-230; a zero-length vector: where(length == 0, 1, length) -- and, when the result is differentiated, the
 * reference's reverse mode meets sqrt'(0) behind that where: NaN (rule of the header) */
static inline void SFX(normalize2)(JET vx, JET vy, JET* ox, JET* oy) {
    JET len = jsqrt(jadd(jmul(vx, vx), jmul(vy, vy)));
    if (len.v == (REAL)0) { len = jc((REAL)1); J_poison = 1; }
    *ox = jdiv(vx, len);
    *oy = jdiv(vy, len);
}
#define normalize2 SFX(normalize2)

/* geometry.py:641-650 / 698-711 / 416-419 */
static inline JET SFX(evaluate)(const OBJ* w, JET p0x, JET p0y, JET p1x, JET p1y, JET p2x, JET p2y) {
    if (w->kind == ORC_VERTEX) return jc((REAL)0);
    if (w->kind == ORC_RIS) {
        JET rx, ry;
        normalize2(jsub(p2x, p1x), jsub(p2y, p1y), &rx, &ry);
        JET mx = jmulc((REAL)-1, rx), my = jmulc((REAL)-1, ry); /* -r */
        JET sin_a = jsub(jmulc(w->ny, mx), jmulc(w->nx, my));
        JET cos_a = jadd(jmulc(w->nx, mx), jmulc(w->ny, my));
        JET ds = jsub(sin_a, jc(w->sinp)), dc = jsub(cos_a, jc(w->cosp));
        return jadd(jmul(ds, ds), jmul(dc, dc));
    }
    JET ix, iy, rx, ry;
    normalize2(jsub(p1x, p0x), jsub(p1y, p0y), &ix, &iy);
    normalize2(jsub(p2x, p1x), jsub(p2y, p1y), &rx, &ry);
    JET din = jadd(jmulc(w->nx, ix), jmulc(w->ny, iy));
    JET two_din = jmulc((REAL)2, din);
    JET ex = jsub(rx, jsub(ix, jmulc(w->nx, two_din)));
    JET ey = jsub(ry, jsub(iy, jmulc(w->ny, two_din)));
    return jadd(jmul(ex, ex), jmul(ey, ey));
}
#define evaluate SFX(evaluate)

/* geometry.py:1077-1084 */
static JET SFX(path_loss)(const OBJ* O, const int32_t* cand, int k, const JET* px, const JET* py) {
    JET loss = jc((REAL)0);
    for (int i = 0; i < k; ++i) loss = jadd(loss, evaluate(&O[cand[i]], px[i], py[i], px[i + 1], py[i + 1], px[i + 2], py[i + 2]));
    return loss;
}
/* geometry.py:176-203 */
static JET SFX(path_length)(int k, const JET* px, const JET* py) {
    JET r = jc((REAL)0);
    for (int i = 0; i <= k; ++i) {
        JET vx = jadd(jsub(px[i + 1], px[i]), jc((REAL)REAL_EPS));
        JET vy = jadd(jsub(py[i + 1], py[i]), jc((REAL)REAL_EPS));
        JET ln = jsqrt(jadd(jmul(vx, vx), jmul(vy, vy)));
        r = (i == 0) ? ln : jadd(r, ln);
    }
    return r;
}
#define path_loss SFX(path_loss)
#define path_length SFX(path_length)

/* geometry.py:988-1010: points 1..k from the unknowns (one per Wall / RIS, none per Vertex) */
static void SFX(p2c)(const OBJ* O, const int32_t* cand, int k, const JET* theta, JET* px, JET* py) {
    int j = 0;
    for (int i = 0; i < k; ++i) {
        const OBJ* w = &O[cand[i]];
        if (w->kind == ORC_VERTEX) {
            px[i + 1] = jc(w->ox); py[i + 1] = jc(w->oy);
        } else {
            px[i + 1] = jadd(jc(w->ox), jmulc(w->tx_, theta[j]));
            py[i + 1] = jadd(jc(w->oy), jmulc(w->ty_, theta[j]));
            ++j;
        }
    }
}
#define p2c SFX(p2c)

static inline int SFX(n_unknowns)(const OBJ* O, const int32_t* cand, int k) {
    int n = 0;
    for (int i = 0; i < k; ++i) n += (O[cand[i]].kind != ORC_VERTEX);
    return n;
}
#define n_unknowns SFX(n_unknowns)

static inline JET SFX(objective)(const OBJ* O, const int32_t* cand, int k, int solver, const JET* px, const JET* py) {
    return solver == 2 ? path_length(k, px, py) : path_loss(O, cand, k, px, py);
}
#define objective SFX(objective)

int SFX(orc_opt_objective)(const double* xys, const uint8_t* kind, const double* sincos, int N, int solver, const double* tx,
                           const double* rx, const int32_t* cand, int k, const double* theta, double* value, double* g) {
    OBJ* O = (OBJ*)malloc(sizeof(OBJ) * (N > 0 ? N : 1));
    for (int j = 0; j < N; ++j) SFX(make_obj)(&O[j], kind[j], xys + 4 * j, sincos + 2 * j, (REAL)0);
    const int n = n_unknowns(O, cand, k);
    J_nv = n; J_second = 0; J_poison = 0;
    JET th[ORC_MAX_ORDER], px[ORC_MAX_ORDER + 2], py[ORC_MAX_ORDER + 2];
    for (int i = 0; i < n; ++i) th[i] = jvar((REAL)theta[i], i);
    px[0] = jc((REAL)tx[0]); py[0] = jc((REAL)tx[1]); px[k + 1] = jc((REAL)rx[0]); py[k + 1] = jc((REAL)rx[1]);
    p2c(O, cand, k, th, px, py);
    JET f = objective(O, cand, k, solver, px, py);
    *value = (double)f.v;
    for (int i = 0; i < n; ++i) g[i] = (double)(REAL)f.g[i];
    free(O);
    return 0;
}

/* optax.scale_by_adam + scale(-lr) in oracle/ref.py:616-638's order; t = 1, 2, ...  Returns the update's pieces in place. */
static inline void SFX(adam_consts)(const orc_opt_params* p, int t, REAL* c1, REAL* c2) {
    *c1 = (REAL)(1.0 - pow(p->b1, (double)t));
    *c2 = (REAL)(1.0 - pow(p->b2, (double)t));
}
int SFX(orc_opt_adam_step)(const orc_opt_params* p, int t, double g_, double* x_, double* mu_, double* nu_) {
    REAL c1, c2, g = (REAL)g_, x = (REAL)*x_, mu = (REAL)*mu_, nu = (REAL)*nu_;
    SFX(adam_consts)(p, t, &c1, &c2);
    mu = (REAL)p->b1 * mu + (REAL)(1.0 - p->b1) * g;
    nu = (REAL)p->b2 * nu + (REAL)(1.0 - p->b2) * (g * g);
    REAL mh = mu / c1, nh = nu / c2;
    x = x + (REAL)(-p->lr) * (mh / (SQRT(nh) + (REAL)p->eps));
    *x_ = (double)x; *mu_ = (double)mu; *nu_ = (double)nu;
    return 0;
}

/* ---- validity and path function on first-order duals w.r.t. the cell (J_nv = 2 or 0), as oracle/d2d_oracle_grad.c ---- */
static inline JET SFX(activation)(JET x, const orc_opt_params* p) {
    JET z = jmulc((REAL)p->alpha, x);
    if (p->act == 0) return jdiv(jmin(jmax(jadd(z, jc((REAL)3)), jc((REAL)0)), jc((REAL)6)), jc((REAL)6));
    JET r = jc((REAL)1 / ((REAL)1 + EXP(-z.v))); /* lax.logistic; JVP y (1 - y) */
    const TREAL gg = (TREAL)r.v * ((TREAL)1 - (TREAL)r.v);
    for (int i = 0; i < J_nv; ++i) r.g[i] = gg * z.g[i];
    return r;
}
#define activation SFX(activation)
static inline JET SFX(t_and)(JET a, JET b, int approx) { return approx ? jmin(a, b) : jc((a.v != 0 && b.v != 0) ? (REAL)1 : (REAL)0); }
static inline JET SFX(t_or)(JET a, JET b, int approx) { return approx ? jmax(a, b) : jc((a.v != 0 || b.v != 0) ? (REAL)1 : (REAL)0); }
static inline JET SFX(t_not)(JET a, int approx) { return approx ? jsub(jc((REAL)1), a) : jc(a.v != 0 ? (REAL)0 : (REAL)1); }
static inline JET SFX(t_ge)(JET x, JET y, const orc_opt_params* p) { return p->approx ? activation(jsub(x, y), p) : jc(x.v >= y.v ? (REAL)1 : (REAL)0); }
static inline JET SFX(t_le)(JET x, JET y, const orc_opt_params* p) { return p->approx ? activation(jsub(y, x), p) : jc(x.v <= y.v ? (REAL)1 : (REAL)0); }
static inline JET SFX(t_lt)(JET x, JET y, const orc_opt_params* p) { return p->approx ? activation(jsub(y, x), p) : jc(x.v < y.v ? (REAL)1 : (REAL)0); }
#define t_and SFX(t_and)
#define t_or SFX(t_or)
#define t_not SFX(t_not)
#define t_ge SFX(t_ge)
#define t_le SFX(t_le)
#define t_lt SFX(t_lt)

static inline JET SFX(seg_test)(JET num, JET den, const orc_opt_params* p) { /* geometry.py:163-171 */
    const int den_is_zero = (den.v == (REAL)0);
    JET t = den_is_zero ? jc((REAL)INFINITY) : jdiv(num, den);
    return t_and(t_ge(t, jc((REAL)(-p->seg_tol)), p), t_le(t, jc((REAL)1 + (REAL)p->seg_tol), p), p->approx);
}
#define seg_test SFX(seg_test)
static inline JET SFX(wall_hits)(const OBJ* w, JET p3x, JET p3y, JET p4x, JET p4y, const orc_opt_params* p) { /* geometry.py:82-173 */
    const REAL Ax = w->p2x - w->p1x, Ay = w->p2y - w->p1y;
    JET Bx = jsub(p3x, p4x), By = jsub(p3y, p4y);
    JET Cx = jsub(jc(w->p1x), p3x), Cy = jsub(jc(w->p1y), p3y);
    JET a = jsub(jmul(By, Cx), jmul(Bx, Cy));
    JET b = jsub(jmulc(Ax, Cy), jmulc(Ay, Cx));
    JET d = jsub(jmulc(Ay, Bx), jmulc(Ax, By));
    return t_and(seg_test(a, d, p), seg_test(b, d, p), p->approx);
}
#define wall_hits SFX(wall_hits)

static inline REAL SFX(ipow)(REAL x, int n) { /* lax.integer_pow */
    if (n == 0) return (REAL)1;
    REAL acc = 0; int have = 0;
    while (n > 0) {
        if (n & 1) { acc = have ? acc * x : x; have = 1; }
        n >>= 1;
        if (n > 0) x = x * x;
    }
    return acc;
}

/* One (cell, candidate): the contribution valid * fun as a first-order dual w.r.t. the cell.  cellv: 0 = the cell is the
 * receiver (rx), 1 = the transmitter.  With with_grad the loop carries (theta, cell) second-order jets. */
static JET SFX(eval_candidate)(const OBJ* O, int N, const int32_t* cand, int k, REAL txx, REAL txy, REAL rxx, REAL rxy, int cell_is_tx,
                               const double* theta0, const orc_opt_params* p, int with_grad, double* pts_out, double* loss_out,
                               const int32_t* snaps, int n_snap) {
    const int n = n_unknowns(O, cand, k);
#if !JSECOND
    with_grad = 0; /* (the lean instance: value maps only) */
#endif
    const int nc = with_grad ? 2 : 0; /* cell variables sit behind the thetas */
    /* theta_t, mu_t, nu_t as first-order duals w.r.t. the cell: value + d / d cell (the only history the loop carries) */
    REAL th[ORC_MAX_ORDER], mu[ORC_MAX_ORDER], nu[ORC_MAX_ORDER];
    TREAL dth[ORC_MAX_ORDER][2], dmu[ORC_MAX_ORDER][2], dnu[ORC_MAX_ORDER][2];
    for (int i = 0; i < n; ++i) {
        th[i] = (REAL)theta0[i]; mu[i] = nu[i] = (REAL)0;
        dth[i][0] = dth[i][1] = dmu[i][0] = dmu[i][1] = dnu[i][0] = dnu[i][1] = 0.0;
    }
    JET px[ORC_MAX_ORDER + 2], py[ORC_MAX_ORDER + 2];
    REAL last_loss = (REAL)0;
    TREAL dlast[2] = {0.0, 0.0};
    int poison = 0;
    /* hard validity is a bool and fun = 1 ignores the path: nothing of the contribution is differentiated (no NaN either) */
    const int differentiated = with_grad && (p->approx || p->fun_id != 3);
    /* the objective inside the loop reaches the contribution through theta (n > 0) or, MinPath in the approx modes, through the
     * recorded loss alone */
    const int loop_differentiated = differentiated && (n > 0 || (p->approx && p->solver == 1));
    if (k > 0) {
        for (int t = 1; t <= p->steps; ++t) {
            /* objective at theta_t as a jet in (theta (independent), cell) */
            J_nv = n + nc; J_second = with_grad; J_poison = 0;
            JET thj[ORC_MAX_ORDER];
            for (int i = 0; i < n; ++i) thj[i] = jvar(th[i], i);
            if (with_grad) {
                px[0] = cell_is_tx ? jvar(txx, n) : jc(txx); py[0] = cell_is_tx ? jvar(txy, n + 1) : jc(txy);
                px[k + 1] = cell_is_tx ? jc(rxx) : jvar(rxx, n); py[k + 1] = cell_is_tx ? jc(rxy) : jvar(rxy, n + 1);
            } else {
                px[0] = jc(txx); py[0] = jc(txy); px[k + 1] = jc(rxx); py[k + 1] = jc(rxy);
            }
            p2c(O, cand, k, thj, px, py);
            JET f = objective(O, cand, k, p->solver, px, py);
            if (J_poison && loop_differentiated) poison = 1; /* normalize() of a zero-length vector inside a differentiated objective */
            last_loss = f.v;
            if (with_grad)
                for (int c = 0; c < 2; ++c) { /* total derivative: d f / d cell + sum_i d f / d theta_i  d theta_i / d cell */
                    dlast[c] = f.g[n + c];
                    for (int i = 0; i < n; ++i) dlast[c] += f.g[i] * dth[i][c];
                }
            REAL c1, c2;
            SFX(adam_consts)(p, t, &c1, &c2);
            TREAL dgs[ORC_MAX_ORDER][2]; /* d g_i / d cell, with d theta_t / d cell of THIS step for every unknown */
            for (int i = 0; i < n; ++i) {
                dgs[i][0] = dgs[i][1] = 0.0;
#if JSECOND
                if (with_grad)
                    for (int c = 0; c < 2; ++c) {
                        dgs[i][c] = f.h[i][n + c];
                        for (int j = 0; j < n; ++j) dgs[i][c] += f.h[i][j] * dth[j][c];
                    }
#endif
            }
            for (int i = 0; i < n; ++i) {
                REAL g = (REAL)f.g[i];
                for (int u = 0; u < (p->g_ulps < 0 ? -p->g_ulps : p->g_ulps); ++u)
                    g = (sizeof(REAL) == 4) ? (REAL)nextafterf((float)g, p->g_ulps > 0 ? INFINITY : -INFINITY)
                                            : (REAL)nextafter((double)g, p->g_ulps > 0 ? INFINITY : -INFINITY);
                const TREAL* dg = dgs[i];
                /* oracle/ref.py:633-637 */
                const REAL b1 = (REAL)p->b1, b2 = (REAL)p->b2, ob1 = (REAL)(1.0 - p->b1), ob2 = (REAL)(1.0 - p->b2);
                mu[i] = b1 * mu[i] + ob1 * g;
                const REAL gg = g * g;
                nu[i] = b2 * nu[i] + ob2 * gg;
                const REAL mh = mu[i] / c1, nh = nu[i] / c2;
                const REAL sq = SQRT(nh), den = sq + (REAL)p->eps;
                th[i] = th[i] + (REAL)(-p->lr) * (mh / den);
                if (with_grad)
                    for (int c = 0; c < 2; ++c) {
                        dmu[i][c] = (TREAL)b1 * dmu[i][c] + (TREAL)ob1 * dg[c];
                        dnu[i][c] = (TREAL)b2 * dnu[i][c] + (TREAL)ob2 * ((TREAL)2 * (TREAL)g * dg[c]);
                        const TREAL dmh = dmu[i][c] / (TREAL)c1, dnh = dnu[i][c] / (TREAL)c2;
                        /* sqrt'(0) = inf: the reference's reverse mode meets 0 * inf when g has been 0 exactly all along */
                        TREAL dsq;
                        if (nh == (REAL)0) { dsq = 0; if (differentiated) poison = 1; }
                        else dsq = dnh / ((TREAL)2 * (TREAL)sqrt((double)nh));
                        const TREAL dupd = (dmh - ((TREAL)mh / (TREAL)den) * dsq) / (TREAL)den;
                        dth[i][c] += (TREAL)(REAL)(-p->lr) * dupd;
                    }
            }
            /* the interaction points after t updates, for the steps the caller wants to see (trajectory agreement) */
            for (int sn = 0; pts_out && sn < n_snap; ++sn)
                if (snaps[sn] == t) {
                    double* o = pts_out + (size_t)sn * 2 * ORC_MAX_ORDER;
                    int j = 0;
                    for (int i = 0; i < ORC_MAX_ORDER; ++i) {
                        if (i >= k) { o[2 * i] = o[2 * i + 1] = 0.0; continue; }
                        const OBJ* w = &O[cand[i]];
                        if (w->kind == ORC_VERTEX) { o[2 * i] = (double)w->ox; o[2 * i + 1] = (double)w->oy; }
                        else { o[2 * i] = (double)(w->ox + w->tx_ * th[j]); o[2 * i + 1] = (double)(w->oy + w->ty_ * th[j]); ++j; }
                    }
                }
        }
    }
    /* final points and recorded loss as first-order duals w.r.t. the cell */
    J_nv = nc; J_second = 0; J_poison = 0;
    JET thf[ORC_MAX_ORDER];
    for (int i = 0; i < n; ++i) {
        thf[i] = jc(th[i]);
        for (int c = 0; c < nc; ++c) thf[i].g[c] = dth[i][c];
    }
    if (with_grad) {
        px[0] = cell_is_tx ? jvar(txx, 0) : jc(txx); py[0] = cell_is_tx ? jvar(txy, 1) : jc(txy);
        px[k + 1] = cell_is_tx ? jc(rxx) : jvar(rxx, 0); py[k + 1] = cell_is_tx ? jc(rxy) : jvar(rxy, 1);
    } else {
        px[0] = jc(txx); py[0] = jc(txy); px[k + 1] = jc(rxx); py[k + 1] = jc(rxy);
    }
    p2c(O, cand, k, thf, px, py);
    JET loss;
    if (k == 0) loss = jc((REAL)0);
    else if (p->solver == 2) {
        loss = path_loss(O, cand, k, px, py); /* geometry.py:1204 */
        if (J_poison && differentiated && p->approx) poison = 1;
    } else {
        loss = jc(last_loss); /* geometry.py:1284-1288 */
        for (int c = 0; c < nc; ++c) loss.g[c] = dlast[c];
    }
    if (pts_out && k == 0)
        for (int i = 0; i < 2 * ORC_MAX_ORDER * n_snap; ++i) pts_out[i] = 0.0;
    if (loss_out) *loss_out = (double)loss.v;
    /* on_objects, geometry.py:821-854 (a Vertex contains every point, :397-403) */
    JET on = jc((REAL)1);
    for (int i = 0; i < k; ++i) {
        const OBJ* w = &O[cand[i]];
        JET c;
        if (w->kind == ORC_VERTEX) c = jc((REAL)1);
        else {
            JET ox_ = jsub(px[i + 1], jc(w->ox)), oy_ = jsub(py[i + 1], jc(w->oy));
            REAL sq = w->tx_ * w->tx_ + w->ty_ * w->ty_;
            if (sq == (REAL)0) sq = (REAL)1;
            JET s = jdiv(jadd(jmulc(w->tx_, ox_), jmulc(w->ty_, oy_)), jc(sq));
            c = t_and(t_ge(s, jc((REAL)0), p), t_le(s, jc((REAL)1), p), p->approx);
        }
        on = t_and(on, c, p->approx);
    }
    /* intersects_with_objects, geometry.py:856-906 (a Vertex intersects nothing, :407-414) */
    JET hit = jc((REAL)0);
    for (int i = 0; i <= k; ++i) {
        const int ig0 = (i == 0) ? -1 : cand[i - 1];
        const int ig1 = (i == k) ? -1 : cand[i];
        for (int j = 0; j < N; ++j) {
            if (j == ig0 || j == ig1) continue;
            JET h = O[j].kind == ORC_VERTEX ? jc((REAL)0) : wall_hits(&O[j], px[i], py[i], px[i + 1], py[i + 1], p);
            hit = t_or(hit, h, p->approx);
        }
    }
    JET ok = t_lt(loss, jc((REAL)p->tol), p);
    JET valid = t_and(t_and(on, t_not(hit, p->approx), p->approx), ok, p->approx);
    if (valid.v != valid.v) valid = jc((REAL)0); /* jnp.nan_to_num */
    JET r = path_length(k, px, py);
    JET f;
    switch (p->fun_id) {
        case 0: f = jdiv(jc(SFX(ipow)((REAL)p->r_coef, k)), jadd(jc((REAL)p->height * (REAL)p->height), jmul(r, r))); break;
        case 1: f = jmul(r, r); break;
        case 2: f = r; break;
        default: f = jc((REAL)1); break;
    }
    JET out = jmul(valid, f);
    if (poison)
        for (int c = 0; c < nc; ++c) out.g[c] = NAN;
    return out;
}

int SFX(orc_opt_power_map)(const double* xys, const uint8_t* kind, const double* sincos, int N, const orc_opt_params* p,
                           const double* fixed, const double* X, const double* Y, long ncell, const int32_t* cands,
                           const int32_t* cand_k, long C, const double* theta0, double* value, double* grad, double* pts,
                           double* loss, const int32_t* snaps, int n_snap, int nthreads) {
    if (N < 0 || p->steps < 1 || (p->solver != 1 && p->solver != 2)) return -1;
    OBJ* O = (OBJ*)malloc(sizeof(OBJ) * (N > 0 ? N : 1));
    for (int j = 0; j < N; ++j) SFX(make_obj)(&O[j], kind[j], xys + 4 * j, sincos + 2 * j, (REAL)p->patch);
#ifdef _OPENMP
    /* (a num_threads clause, not omp_set_num_threads: the count must not stick to later calls that ask for the default) */
    const int nt_ = nthreads > 0 ? nthreads : omp_get_max_threads();
#endif
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt_)
    for (long c = 0; c < ncell; ++c) {
        REAL acc = (REAL)0;
        double gx = 0.0, gy = 0.0;
        const REAL cx = (REAL)X[c], cy = (REAL)Y[c], fx = (REAL)fixed[0], fy = (REAL)fixed[1];
        for (long ci = 0; ci < C; ++ci) {
            JET t = p->grid_is_tx
                        ? SFX(eval_candidate)(O, N, cands + ORC_MAX_ORDER * ci, cand_k[ci], cx, cy, fx, fy, 1, theta0 + ORC_MAX_ORDER * ci, p,
                                              grad != NULL, pts ? pts + ((size_t)c * C + ci) * n_snap * 2 * ORC_MAX_ORDER : NULL,
                                              loss ? loss + (size_t)c * C + ci : NULL, snaps, n_snap)
                        : SFX(eval_candidate)(O, N, cands + ORC_MAX_ORDER * ci, cand_k[ci], fx, fy, cx, cy, 0, theta0 + ORC_MAX_ORDER * ci, p,
                                              grad != NULL, pts ? pts + ((size_t)c * C + ci) * n_snap * 2 * ORC_MAX_ORDER : NULL,
                                              loss ? loss + (size_t)c * C + ci : NULL, snaps, n_snap);
            acc = acc + t.v; /* scene.py:1909 */
            if (grad) { gx += t.g[0]; gy += t.g[1]; }
        }
        value[c] = (double)acc;
        if (grad) { grad[2 * c] = gx; grad[2 * c + 1] = gy; }
    }
    free(O);
    return 0;
}

#undef NV
#undef IF2
#undef JET
#undef J_nv
#undef J_second
#undef J_poison
#undef jc
#undef jvar
#undef jadd
#undef jsub
#undef jmul
#undef jmulc
#undef jdiv
#undef jsqrt
#undef jmin
#undef jmax
#undef OBJ
#undef normalize2
#undef evaluate
#undef path_loss
#undef path_length
#undef p2c
#undef n_unknowns
#undef objective
#undef activation
#undef t_and
#undef t_or
#undef t_not
#undef t_ge
#undef t_le
#undef t_lt
#undef seg_test
#undef wall_hits

#endif
