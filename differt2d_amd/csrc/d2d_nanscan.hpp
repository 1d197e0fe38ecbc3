// NaN scan of the value+grad sweeps: where does the reference's reverse-mode autodiff return NaN?
//
// jax.grad through ImagePath.from_tx_objects_rx yields NaN for a (cell, candidate) pair in three situations, whether
// or not the candidate is valid for the cell (DESIGN.md "NaN parity"):
//   (1) un == 0 in some step of the backward scan: jnp.where(un == 0, 0, vn * u / un) sends a zero cotangent through the
//       untaken division by zero, 0 * inf = NaN (differt2d/geometry.py:1105);
//   (2) approx modes only: a zero-length segment of the path inside the differentiated loss, normalize()'s sqrt'(0) * 0
//       (differt2d/geometry.py:227-228, 647-648); in hard mode the loss only feeds a boolean;
//   (3) a segment vector of exactly (-eps, -eps): path_length adds eps to both components (geometry.py:199-200) and
//       differentiates the norm of (0, 0) -- found by the C gradient oracle on lattice scenes (every path function but 1).
// All depend on the image chain and the backward scan of the candidate alone -- not on its wall loop (> 95 % of a
// candidate's work) -- and all are exact-zero events of fp32 expressions whose real-number zero sets are LINES in the
// cell's plane.  The culled value+grad sweep never evaluates the candidates that the tile culling proves invalid, so it
// cannot see their NaN; this pass finds them all:
//   * lanes = candidates: a conservative test against the patch's bounding box, "can any of the zero events happen for any
//     cell of the box?" (nan_possible_*), with explicit rounding margins -- a line crosses ~1.5 % of the 8 x 8 patches of a
//     1024^2 grid;
//   * lanes = cells, for the survivors: the backward scan itself, operation for operation what eval_candidate<GRAD> and the
//     exhaustive power_vg_kernel compute (nan_probe), and the flags they would raise;
//   * flagged cells get a NaN gradient, the patch's row of scene-VJP partial sums a NaN for the fixed end point and for the
//     walls of the flagged candidates -- what power_vg_kernel (d2d_params.strict_nan) writes, position for position.
// Launched behind the culled value+grad sweep, in front of the VJP reduction.
#pragma once
#include "d2d_kernels.hpp"

namespace d2d {

// The backward scan of eval_candidate<K, MODE, STATS, GRAD = true> (geometry.py:1093-1110), same operations, same order.
// (txx, txy): px[0]; (rxx, rxy): px[K + 1] where the scan starts; img: the image chain of px[0].
template <int K, bool APPROX>
__device__ __forceinline__ bool nan_probe(const SweepArgs& a, const int (&cand)[D2D_MAX_ORDER], const float (&imgx)[D2D_MAX_ORDER],
                                          const float (&imgy)[D2D_MAX_ORDER], float txx, float txy, float rxx, float rxy) {
    const bool plen = a.fun_id != D2D_FUN_ONE;  // the path function goes through path_length (rule 3 below)
    float px[K + 2], py[K + 2];
    px[0] = txx;
    py[0] = txy;
    px[K + 1] = rxx;
    py[K + 1] = rxy;
    bool znan = false;
    float ptx = rxx, pty = rxy;
#pragma unroll
    for (int i = K - 1; i >= 0; --i) {
        const float4 r0 = ldc4(a.refl, 2 * cand[i]);
        float ux = ptx - imgx[i], uy = pty - imgy[i];
        float vx = r0.x - ptx, vy = r0.y - pty;
        float un = ux * r0.z + uy * r0.w;
        float vn = vx * r0.z + vy * r0.w;
        bool z = (un == 0.0f);
        float den = z ? 1.0f : un;
        float incx, incy;
        div2_exact(vn * ux, vn * uy, den, incx, incy);
        incx = z ? 0.0f : incx;
        incy = z ? 0.0f : incy;
        ptx = ptx + incx;
        pty = pty + incy;
        px[i + 1] = ptx;
        py[i + 1] = pty;
        // (hard validity with fun = 1: nothing is differentiated through the path)
        if (APPROX || plen) znan = znan || z;
    }
    if (APPROX) {
#pragma unroll
        for (int i = 0; i <= K; ++i) znan = znan || (px[i + 1] == px[i] && py[i + 1] == py[i]);
    }
    if (plen) {
        // (3) path_length: a segment vector of exactly (-eps, -eps), geometry.py:199-200 (eval_candidate<GRAD> has the same lines)
#pragma unroll
        for (int i = 0; i <= K; ++i) znan = znan || (((px[i + 1] - px[i]) + D2D_EPS == 0.0f) && ((py[i + 1] - py[i]) + D2D_EPS == 0.0f));
    }
    return znan;
}

__device__ __forceinline__ float4 wallc_r0(const WallC& w) { return make_float4(w.ox, w.oy, w.nx, w.ny); }

// Thin quad around the part [sa, sb] of wall w's line that an fp32 interaction point may occupy (as cull_candidate builds
// it, without clipping to the wall: a NaN does not care whether the point is on the wall)
__device__ __forceinline__ void nan_quad(const WallC& w, float sa, float sb, float E, float (&qx)[4], float (&qy)[4]) {
    const float eps = 1.1920929e-07f;
    const float d = 64.0f * eps * 2.0f * E;  // the fp32 point may sit this far off the wall's line
    const float eax = __builtin_fmaf(sa, w.tx, w.ox), eay = __builtin_fmaf(sa, w.ty, w.oy);
    const float ebx = __builtin_fmaf(sb, w.tx, w.ox), eby = __builtin_fmaf(sb, w.ty, w.oy);
    const float ddx = d * w.nx, ddy = d * w.ny;
    const float tl = fabsf(w.tx) + fabsf(w.ty);
    const float px_ = d * w.tx * w.rsq * tl + 4.0f * eps * (fabsf(sa) + fabsf(sb)) * fabsf(w.tx);
    const float py_ = d * w.ty * w.rsq * tl + 4.0f * eps * (fabsf(sa) + fabsf(sb)) * fabsf(w.ty);
    qx[0] = eax - px_ + ddx; qy[0] = eay - py_ + ddy;
    qx[1] = eax - px_ - ddx; qy[1] = eay - py_ - ddy;
    qx[2] = ebx + px_ + ddx; qy[2] = eby + py_ + ddy;
    qx[3] = ebx + px_ - ddx; qy[3] = eby + py_ - ddy;
}

// RX grids.  Candidate (w[0] .. w[K-1]) with the transmitter's image chain (Ix, Iy); (bx, by): corners of the cells' box.
// true = some cell of the box MAY raise a flag in nan_probe; false = certainly none does.
// Level lvl (from the cell inwards) knows a convex region Q that contains the fp32 point the scan step of wall lvl starts
// from (the box, then thin quads around the previous wall's line): un = (q - I).n and vn = (o - q).n are affine in q, so a
// common sign at the 4 vertices with a margin above the expressions' rounding holds throughout Q; un != 0 there also makes
// the step's point a linear-fractional function of q without a pole in Q, whose range over Q the vertices span.
// plen: the path function goes through path_length, whose own trap -- a segment vector of exactly (-eps, -eps), rule (3) --
// needs two consecutive points within eps sqrt 2 of each other (an ABSOLUTE distance: NAN_ABS below) and, the step being
// along u, u_x == u_y to rounding; in the approx modes rule (2)'s test (any near-coincidence) covers it, in hard mode the
// direction condition keeps the survivors few.
constexpr float NAN_ABS = 1e-6f;
template <int K, bool APPROX>
// near_last: what is known about the LAST wall's near-coincidence test from the cells themselves (nan_near_bits): 0 = no cell
// of the box can have its interaction point within the bound of itself, 1 = some cell may, -1 = not known (the box's corners decide)
__device__ __forceinline__ bool nan_possible_rx(const float (&bx)[4], const float (&by)[4], const WallC (&w)[K], const float (&Ix)[K],
                                                const float (&Iy)[K], float fx, float fy, bool plen, int near_last = -1) {
    const float eps = 1.1920929e-07f;
    const float abs3 = plen ? NAN_ABS : 0.0f;
    const bool near_test = APPROX || plen;
    float qx[4], qy[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        qx[j] = bx[j];
        qy[j] = by[j];
    }
#pragma unroll
    for (int lvl = K - 1; lvl >= 0; --lvl) {
        const WallC& wl = w[lvl];
        bool pos = true, neg = true, vpos = true, vneg = true, dpos = true, dneg = true, fin = true;
        float smin = __builtin_inff(), smax = -__builtin_inff(), E = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ux = qx[j] - Ix[lvl], uy = qy[j] - Iy[lvl];
            const float vx = wl.ox - qx[j], vy = wl.oy - qy[j];
            const float un = __builtin_fmaf(ux, wl.nx, uy * wl.ny);
            const float vn = __builtin_fmaf(vx, wl.nx, vy * wl.ny);
            const float du = 16.0f * eps * __builtin_fmaf(fabsf(wl.nx), fabsf(qx[j]) + fabsf(Ix[lvl]), fabsf(wl.ny) * (fabsf(qy[j]) + fabsf(Iy[lvl])));
            pos = pos && (un > du);
            neg = neg && (un < -du);
            if (near_test) {
                // the step's point differs from q by inc = vn u / un, |inc|_inf >= |vn| / sqrt 2: above a few ulps of q's
                // coordinates the sum cannot round back to q (nor, plen, stay within eps sqrt 2 of it)
                const float mq = (fabsf(qx[j]) + fabsf(qy[j])) + (fabsf(wl.ox) + fabsf(wl.oy));
                const float dv = 32.0f * eps * mq + abs3;
                vpos = vpos && (vn > dv);
                vneg = vneg && (vn < -dv);
                if (!APPROX) {
                    // rule (3): fl(pt + inc) - pt == (eps, eps) needs both components of inc = (vn / un) u within [eps / 2,
                    // 3 eps / 2] (the sum is rounded to a grid no coarser than eps, else the rule cannot fire at all): u lies in
                    // the double cone u_x u_y > 0, 1/3 <= u_x / u_y <= 3 -- bounded by f1 = u_y - u_x / 3.5, f2 = 3.5 u_x - u_y,
                    // which have the same sign inside it and opposite signs throughout a region that avoids it
                    const float f1 = uy - ux * (1.0f / 3.5f), f2 = 3.5f * ux - uy;
                    const float dm = 64.0f * eps * ((fabsf(qx[j]) + fabsf(qy[j])) + (fabsf(Ix[lvl]) + fabsf(Iy[lvl])));
                    dpos = dpos && (f1 > dm) && (f2 < -dm);
                    dneg = dneg && (f1 < -dm) && (f2 > dm);
                }
            }
            const float g = vn * __builtin_amdgcn_rcpf(un);
            const float dx = __builtin_fmaf(g, ux, -vx), dy = __builtin_fmaf(g, uy, -vy);
            const float s = __builtin_fmaf(wl.ty, dy, wl.tx * dx) * wl.rsq;
            const float mag = __builtin_fmaf(fabsf(g), fabsf(ux) + fabsf(uy), fabsf(vx) + fabsf(vy));
            fin = fin && (mag < 1e18f);
            smin = fminf(smin, s);
            smax = fmaxf(smax, s);
            E = fmaxf(E, mag);
        }
        if (!((pos || neg) && fin)) return true;
        const bool near = (lvl == K - 1 && near_last >= 0) ? (near_last != 0) : !(vpos || vneg);
        if (near_test && near && (APPROX || !(dpos || dneg))) return true;
        E = E + (fabsf(wl.ox) + fabsf(wl.oy)) + (fabsf(Ix[lvl]) + fabsf(Iy[lvl]));
        const float M = __builtin_fmaf(64.0f * eps * wl.rsq * (fabsf(wl.tx) + fabsf(wl.ty)), 2.0f * E, 1e-30f);
        if (!(E < 1e18f) || !(fabsf(smin) < 1e18f) || !(fabsf(smax) < 1e18f)) return true;
        if (lvl == 0) {
            if (near_test) {
                // first segment: the fixed end point on (the fp32 neighbourhood of) the first wall's line?
                const float fn = __builtin_fmaf(fx - wl.ox, wl.nx, (fy - wl.oy) * wl.ny);
                const float df = 64.0f * eps * 2.0f * E + 32.0f * eps * ((fabsf(fx) + fabsf(fy)) + (fabsf(wl.ox) + fabsf(wl.oy))) + abs3;
                if (!(fabsf(fn) > 2.0f * df)) return true;
            }
            break;
        }
        nan_quad(wl, smin - M, smax + M, E, qx, qy);
    }
    return false;
}

// TX grids (scene.py:1489-1648): the cells are transmitters -- the image chain is the CELL's (per lane in the exact chain),
// the scan starts from the fixed receiver F.  Over a box of cells the images J_lvl(c) fill the parallelogram of the
// corners' images (reflections are affine), the scan's point a region P (F itself, then thin quads), and the two are
// bounded independently: un = pt.n - J.n is a difference of two affine functions, each spanned by its 4 vertices; the
// step's point is linear-fractional in pt for fixed J and in J for fixed pt, pole-free once un cannot vanish, so its range
// over P x J(box) is spanned by the 16 vertex pairs.  Looser than the RX-grid test (pt and c are treated as unrelated), never wrong.
template <int K, bool APPROX>
__device__ __forceinline__ bool nan_possible_txg(const float (&cx)[4], const float (&cy)[4], const WallC (&w)[K], float fx, float fy, bool plen) {
    const float eps = 1.1920929e-07f;
    const float abs3 = plen ? NAN_ABS : 0.0f;
    const bool near_test = APPROX || plen;
    float Jx[K][4], Jy[K][4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        float x = cx[v], y = cy[v];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            image_of(wallc_r0(w[i]), x, y, Jx[i][v], Jy[i][v]);
            x = Jx[i][v];
            y = Jy[i][v];
        }
    }
    float Px[4], Py[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        Px[j] = fx;
        Py[j] = fy;
    }
    float dP_prev = 0.0f;  // how far the fp32 point of the previous level may sit from its exact value (the thin quad's allowances)
#pragma unroll
    for (int lvl = K - 1; lvl >= 0; --lvl) {
        const WallC& wl = w[lvl];
        float amin = __builtin_inff(), amax = -__builtin_inff(), bmin = __builtin_inff(), bmax = -__builtin_inff();
        float magP = 0.0f, magJ = 0.0f, vnabs = 0.0f;
        float cmin = __builtin_inff(), cmax = -__builtin_inff(), emin = __builtin_inff(), emax = -__builtin_inff();
        float c2min = __builtin_inff(), c2max = -__builtin_inff(), e2min = __builtin_inff(), e2max = -__builtin_inff();
        bool vpos = true, vneg = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pa = __builtin_fmaf(Px[j], wl.nx, Py[j] * wl.ny), jb = __builtin_fmaf(Jx[lvl][j], wl.nx, Jy[lvl][j] * wl.ny);
            amin = fminf(amin, pa);
            amax = fmaxf(amax, pa);
            bmin = fminf(bmin, jb);
            bmax = fmaxf(bmax, jb);
            magP = fmaxf(magP, fabsf(Px[j]) + fabsf(Py[j]));
            magJ = fmaxf(magJ, fabsf(Jx[lvl][j]) + fabsf(Jy[lvl][j]));
            vnabs = fmaxf(vnabs, fabsf(__builtin_fmaf(wl.ox - Px[j], wl.nx, (wl.oy - Py[j]) * wl.ny)));
            if (near_test) {
                const float vn = __builtin_fmaf(wl.ox - Px[j], wl.nx, (wl.oy - Py[j]) * wl.ny);
                const float dv = 32.0f * eps * ((fabsf(Px[j]) + fabsf(Py[j])) + (fabsf(wl.ox) + fabsf(wl.oy))) + abs3;
                vpos = vpos && (vn > dv);
                vneg = vneg && (vn < -dv);
                if (!APPROX) {
                    // (rule (3)'s double cone, nan_possible_rx: f1 = u_y - u_x / 3.5 and f2 = 3.5 u_x - u_y with u = P - J, each a
                    // difference of an affine function of P and one of J)
                    const float p1 = Py[j] - Px[j] * (1.0f / 3.5f), p2 = 3.5f * Px[j] - Py[j];
                    const float j1 = Jy[lvl][j] - Jx[lvl][j] * (1.0f / 3.5f), j2 = 3.5f * Jx[lvl][j] - Jy[lvl][j];
                    cmin = fminf(cmin, p1); cmax = fmaxf(cmax, p1);
                    emin = fminf(emin, j1); emax = fmaxf(emax, j1);
                    c2min = fminf(c2min, p2); c2max = fmaxf(c2max, p2);
                    e2min = fminf(e2min, j2); e2max = fmaxf(e2max, j2);
                }
            }
        }
        if (!(magP < 1e18f) || !(magJ < 1e18f)) return true;
        // (the cell's own fp32 images sit within a few ulps per reflection of the corners' hull: inside this margin)
        const float du = 64.0f * eps * (fabsf(wl.nx) + fabsf(wl.ny)) * (magP + magJ);
        bool tight = false;
        float E_tight = 0.0f;
        if (K == 2 && lvl == 0) {
            // Order 2, the second step.  Bounded independently, the point pt1 (an interval of wall 1's line) and the image J0(c)
            // (the parallelogram of the corners' images) leave un0 = (pt1 - J0) . n0 a wide range; but they move TOGETHER with the
            // cell: pt1 - J1 = mu (F - J1) with mu = (o1 - J1) . n1 / un1 (nan_probe's first step), pt1 is its own mirror image in
            // wall 1 and J0 that of J1, hence pt1 - J0 = mu R1(F - J1) and
            //     un0 = - mu A,   A = (J1(c) - F) . R1 n0     (R1 x = x - 2 (x . n1) n1)
            // with A AFFINE in the cell (reflections are affine): one sign at the 4 corners, beyond what rounding can undo
            // (the fp32 pt1 within dP_prev of the exact one, J1 a few ulps from the mirror image of J0, the dot product's own
            // rounding -- divided by the smallest |mu|), holds for every cell of the box.  (un1 has one sign over the box: the
            // level above passed.)
            const WallC& w1 = w[1];
            const float nn = __builtin_fmaf(wl.nx, w1.nx, wl.ny * w1.ny);
            const float px_ = __builtin_fmaf(-2.0f * nn, w1.nx, wl.nx), py_ = __builtin_fmaf(-2.0f * nn, w1.ny, wl.ny);  // R1 n0
            float unmax = 0.0f, vjmin = __builtin_inff(), aamin = __builtin_inff(), magJ1 = 0.0f;
            bool vjp = true, vjn = true, ap = true, an = true;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float un1 = __builtin_fmaf(fx - Jx[1][v], w1.nx, (fy - Jy[1][v]) * w1.ny);
                const float vj = __builtin_fmaf(w1.ox - Jx[1][v], w1.nx, (w1.oy - Jy[1][v]) * w1.ny);
                const float av = __builtin_fmaf(Jx[1][v] - fx, px_, (Jy[1][v] - fy) * py_);
                unmax = fmaxf(unmax, fabsf(un1));
                vjmin = fminf(vjmin, fabsf(vj));
                aamin = fminf(aamin, fabsf(av));
                magJ1 = fmaxf(magJ1, fabsf(Jx[1][v]) + fabsf(Jy[1][v]));
                vjp = vjp && vj > 0.0f; vjn = vjn && vj < 0.0f;
                ap = ap && av > 0.0f; an = an && av < 0.0f;
            }
            const float n1l = fabsf(wl.nx) + fabsf(wl.ny);
            const float mj = 32.0f * eps * (magJ1 + (fabsf(w1.ox) + fabsf(w1.oy)) + (fabsf(fx) + fabsf(fy)));
            if (!(vjp || vjn) || !(ap || an) || !(vjmin > 2.0f * mj) || !(unmax < 1e18f) || !(magJ1 < 1e18f)) return true;
            const float mu_lo = 0.98f * (vjmin - mj) / (unmax + mj);                   // |mu| >= this
            const float err = (dP_prev + 32.0f * eps * magJ1) * n1l + du;              // fl(un0) against -mu A
            const float ma = 1.02f * err / mu_lo + 2.0f * mj * (fabsf(px_) + fabsf(py_));  // ... and fl(A) against A
            if (!(aamin > ma) || !(mu_lo > 0.0f)) return true;
            const float u0lo = mu_lo * (aamin - ma) - err;                              // |fl(un0)| >= this over the box
            if (!(u0lo > 0.0f)) return true;
            E_tight = __builtin_fmaf(1.02f * vnabs / u0lo, magP + magJ, (fabsf(wl.ox) + fabsf(wl.oy)) + magP);
            tight = true;
        } else if (!((amin - bmax > du) || (amax - bmin < -du))) return true;
        if (near_test && !(vpos || vneg)) {
            if (APPROX) return true;
            const float dm = 256.0f * eps * (magP + magJ);  // hard mode, rule (3): can u = P - J enter the double cone?
            const float f1lo = cmin - emax, f1hi = cmax - emin, f2lo = c2min - e2max, f2hi = c2max - e2min;
            if (!((f1lo > dm && f2hi < -dm) || (f1hi < -dm && f2lo > dm))) return true;
        }
        bool fin = true;
        float smin = __builtin_inff(), smax = -__builtin_inff(), E = 0.0f;
        if (tight) {
            E = E_tight;  // (the last level: only the magnitudes are asked for below)
            smin = smax = 0.0f;
        } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float vx = wl.ox - Px[j], vy = wl.oy - Py[j];
            const float vn = __builtin_fmaf(vx, wl.nx, vy * wl.ny);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float ux = Px[j] - Jx[lvl][v], uy = Py[j] - Jy[lvl][v];
                const float un = __builtin_fmaf(ux, wl.nx, uy * wl.ny);
                const float g = vn * __builtin_amdgcn_rcpf(un);
                const float dx = __builtin_fmaf(g, ux, -vx), dy = __builtin_fmaf(g, uy, -vy);
                const float s = __builtin_fmaf(wl.ty, dy, wl.tx * dx) * wl.rsq;
                const float mag = __builtin_fmaf(fabsf(g), fabsf(ux) + fabsf(uy), fabsf(vx) + fabsf(vy));
                fin = fin && (mag < 1e18f);
                smin = fminf(smin, s);
                smax = fmaxf(smax, s);
                E = fmaxf(E, mag);
            }
        }
        }
        if (!fin) return true;
        E = E + (fabsf(wl.ox) + fabsf(wl.oy)) + magJ + magP;
        const float M = __builtin_fmaf(256.0f * eps * wl.rsq * (fabsf(wl.tx) + fabsf(wl.ty)), 2.0f * E, 1e-30f);
        if (!(E < 1e18f) || !(fabsf(smin) < 1e18f) || !(fabsf(smax) < 1e18f)) return true;
        if (lvl == 0) {
            if (near_test) {
                // first segment cell -> first wall: a cell on (the fp32 neighbourhood of) the wall's line?
                bool cpos = true, cneg = true;
                const float dq = 64.0f * eps * 2.0f * E + abs3;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float cn = __builtin_fmaf(cx[v] - wl.ox, wl.nx, (cy[v] - wl.oy) * wl.ny);
                    const float dc = 2.0f * (dq + 32.0f * eps * ((fabsf(cx[v]) + fabsf(cy[v])) + (fabsf(wl.ox) + fabsf(wl.oy))));
                    cpos = cpos && (cn > dc);
                    cneg = cneg && (cn < -dc);
                }
                if (!(cpos || cneg)) return true;
            }
            break;
        }
        // (what nan_quad allows the fp32 point around the exact one: 64 eps 2 E off the line, M along it)
        dP_prev = 1.5f * (64.0f * eps * 2.0f * E) + M * (fabsf(wl.tx) + fabsf(wl.ty));
        nan_quad(wl, smin - M, smax + M, E, Px, Py);
    }
    return false;
}

// One wave per 8 x 8 patch.  Prefix (w_0 .. w_{K-2}) wave-uniform, lanes = last wall (the odometer of sweep_order_culled
// without any skip: a NaN does not care about shadows), conservative test per lane, exact probe of the survivors with
// lanes = cells.  Order of the candidates is irrelevant here: the flags are OR-ed.
//   flags[0]: a cell of this lane raised a flag; wallnan (LDS, [N]): walls of flagged candidates.
template <int K, bool APPROX, bool TXG>
__device__ __forceinline__ void nan_scan_order(const SweepArgs& a, const float4* tab, int* wallnan, const float (&bx)[4], const float (&by)[4],
                                               float cx, float cy, bool force, bool& cell_nan, bool& any_nan, unsigned long long& n_probe) {
    static_assert(K >= 1, "order 0 has no scan");
    const int lane = threadIdx.x & 63;
    const bool plen = a.fun_id != D2D_FUN_ONE;
    const int Nc = a.Nc;
    if (Nc < 1 || (K >= 2 && Nc < 2)) return;
    const int n_chunks = (Nc + 63) >> 6;
    int cand[D2D_MAX_ORDER] = {-1, -1, -1, -1};
    float imgx[D2D_MAX_ORDER], imgy[D2D_MAX_ORDER];
    int pos[D2D_MAX_ORDER] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 1; d < K - 1; ++d) pos[d] = (pos[d - 1] == 0) ? 1 : 0;  // no equal neighbours
    while (true) {
        WallC wu[K];  // [0, K-1): the prefix' walls (wave-uniform)
#pragma unroll
        for (int d = 0; d < K - 1; ++d) {
            cand[d] = cmem(a.cw)[pos[d]];
            const float4 r0 = ldc4(a.refl, 2 * cand[d]);
            wu[d] = make_wallc(r0, ldc4(a.refl, 2 * cand[d] + 1), ldc4(a.flt, cand[d]), cand[d]);
            // RX grids: the transmitter's images (the exact chain's own, geometry.py:1086-1091); TX grids: per cell, below
            if (!TXG) image_of(r0, d == 0 ? a.txx : imgx[d > 0 ? d - 1 : 0], d == 0 ? a.txy : imgy[d > 0 ? d - 1 : 0], imgx[d], imgy[d]);
        }
        const float pIx = (K == 1) ? a.txx : imgx[K >= 2 ? K - 2 : 0];
        const float pIy = (K == 1) ? a.txy : imgy[K >= 2 ? K - 2 : 0];
        const int last_prefix_pos = (K == 1) ? -1 : pos[K >= 2 ? K - 2 : 0];
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            const int lp = chunk * 64 + lane;
            bool alive = (lp < Nc) && (lp != last_prefix_pos);
            if (!force) {
                const int wl = cmem(a.cw)[lp < Nc ? lp : 0];
                const float4 r0 = tab[2 * wl], r1 = tab[2 * wl + 1], fc = tab[2 * a.N + wl];
                WallC w[K];
#pragma unroll
                for (int d = 0; d < K - 1; ++d) w[d] = wu[d];
                w[K - 1] = make_wallc(r0, r1, fc, wl);
                if constexpr (TXG) {
                    if (alive) alive = nan_possible_txg<K, APPROX>(bx, by, w, a.txx, a.txy, plen);
                } else {
                    float Ix[K], Iy[K];
#pragma unroll
                    for (int d = 0; d < K - 1; ++d) {
                        Ix[d] = imgx[d];
                        Iy[d] = imgy[d];
                    }
                    image_of(r0, pIx, pIy, Ix[K - 1], Iy[K - 1]);
                    if (alive) alive = nan_possible_rx<K, APPROX>(bx, by, w, Ix, Iy, a.txx, a.txy, plen);
                }
            }
            unsigned long long mask = __ballot(alive);
            n_probe += (unsigned long long)__builtin_popcountll(mask);
            while (mask) {
                const int b = __builtin_ctzll(mask);
                mask &= mask - 1;
                cand[K - 1] = cmem(a.cw)[chunk * 64 + b];
                bool z;
                if constexpr (TXG) {
                    // images of the lane's cell (eval_candidate<TXG>: sweep_order_culled_txg builds them the same way)
                    float ex[D2D_MAX_ORDER], ey[D2D_MAX_ORDER];
#pragma unroll
                    for (int d = 0; d < K; ++d)
                        image_of(ldc4(a.refl, 2 * cand[d]), d == 0 ? cx : ex[d > 0 ? d - 1 : 0], d == 0 ? cy : ey[d > 0 ? d - 1 : 0], ex[d], ey[d]);
                    z = nan_probe<K, APPROX>(a, cand, ex, ey, cx, cy, a.txx, a.txy);
                } else {
                    image_of(ldc4(a.refl, 2 * cand[K - 1]), pIx, pIy, imgx[K - 1], imgy[K - 1]);
                    z = nan_probe<K, APPROX>(a, cand, imgx, imgy, a.txx, a.txy, cx, cy);
                }
                cell_nan = cell_nan || z;
                if (wave_any(z)) {
                    any_nan = true;
                    if (lane == 0) {
#pragma unroll
                        for (int d = 0; d < K; ++d) wallnan[cand[d]] = 1;
                    }
                }
            }
        }
        if (K == 1) break;
        bool carry = true;
        int stop = -1;
#pragma unroll
        for (int d = K - 2; d >= 0; --d) {
            if (carry) {
                pos[d] += 1;
                if (d > 0 && pos[d] == pos[d - 1]) pos[d] += 1;
                if (pos[d] < Nc) {
                    carry = false;
                    stop = d;
                }
            }
        }
        if (carry) break;
#pragma unroll
        for (int e = 1; e < K - 1; ++e)
            if (e > stop) pos[e] = (pos[e - 1] == 0) ? 1 : 0;
    }
}

// a.grad: [m][n][2] the culled sweep's per-cell gradient; a.partial: its per-patch rows of the scene VJP, or null.
// stats (may be null): [0] (patch, candidate) pairs probed, [1] flagged cells, [2] patches with a flag; the region kernel also
// [3] probes made by the wave itself (queue full), [4] refused queue items (must be 0), [5] rounds.
template <bool APPROX, bool TXG, int MAXK>
__global__ void __launch_bounds__(64) nan_scan_kernel(SweepArgs a, unsigned long long* __restrict__ stats) {
    extern __shared__ float4 tab[];  // [2N] refl, [N] flt, then [N] int flags
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 2 * a.N; i += 64) tab[i] = ldc4(a.refl, i);
    for (int i = lane; i < a.N; i += 64) tab[2 * a.N + i] = ldc4(a.flt, i);
    int* wallnan = reinterpret_cast<int*>(tab + 3 * a.N);
    for (int i = lane; i < a.N; i += 64) wallnan[i] = 0;
    __syncthreads();
    const int tiles_x = (a.n + TILE_W - 1) / TILE_W;
    const long tile = blockIdx.x;
    const int tcol = (int)(tile % tiles_x), trow = (int)(tile / tiles_x);
    const int col = tcol * TILE_W + (lane & (TILE_W - 1));
    const int row = trow * TILE_H + (lane / TILE_W);
    const bool in_range = (col < a.n) && (row < a.m);
    const int ccol = col < a.n ? col : a.n - 1;
    const int crow = row < a.m ? row : a.m - 1;
    const long idx = (long)crow * a.n + ccol;
    const float cx = a.X[idx], cy = a.Y[idx];
    const bool lane_bad = !(fabsf(cx) < 1e18f) || !(fabsf(cy) < 1e18f) || !(fabsf(a.txx) < 1e18f) || !(fabsf(a.txy) < 1e18f);
    float x0 = cx, x1 = cx, y0 = cy, y1 = cy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, off, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, off, 64));
        y0 = fminf(y0, __shfl_xor(y0, off, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
    }
    const float bx[4] = {x0, x1, x1, x0};
    const float by[4] = {y0, y0, y1, y1};
    // a coordinate that is not comfortably finite: no bound holds, every candidate is probed
    const bool force = wave_any(lane_bad);
    bool cell_nan = false, any_nan = false;
    unsigned long long n_probe = 0ull;
    if (a.min_order <= 0 && a.max_order >= 0 && a.fun_id != D2D_FUN_ONE) {
        // order 0 has no scan, but its one segment has rule (3): px[0] = transmitter, px[1] = receiver
        cell_nan = TXG ? (((a.txx - cx) + D2D_EPS == 0.0f) && ((a.txy - cy) + D2D_EPS == 0.0f)) : (((cx - a.txx) + D2D_EPS == 0.0f) && ((cy - a.txy) + D2D_EPS == 0.0f));
        any_nan = wave_any(cell_nan);
    }
    if (a.min_order <= 1 && a.max_order >= 1) nan_scan_order<1, APPROX, TXG>(a, tab, wallnan, bx, by, cx, cy, force, cell_nan, any_nan, n_probe);
    if (a.min_order <= 2 && a.max_order >= 2) nan_scan_order<2, APPROX, TXG>(a, tab, wallnan, bx, by, cx, cy, force, cell_nan, any_nan, n_probe);
    if (MAXK >= 3 && a.min_order <= 3 && a.max_order >= 3) nan_scan_order<3, APPROX, TXG>(a, tab, wallnan, bx, by, cx, cy, force, cell_nan, any_nan, n_probe);
    if (MAXK >= 4 && a.min_order <= 4 && a.max_order >= 4) nan_scan_order<4, APPROX, TXG>(a, tab, wallnan, bx, by, cx, cy, force, cell_nan, any_nan, n_probe);
    const float qnan = __builtin_nanf("");
    const unsigned long long flagged = __ballot(cell_nan && in_range);
    if (a.nan_cell_bits != nullptr) {
        // beside the sweep: leave the flags for nan_apply_kernel (every patch writes its words: nothing to zero beforehand)
        if (lane == 0) a.nan_cell_bits[tile] = flagged;
        if (a.nan_row_bits != nullptr) {
            __syncthreads();
            unsigned* dst = a.nan_row_bits + tile * a.nan_row_words;
            if (lane == 0) dst[0] = any_nan ? 1u : 0u;
            for (int wd = lane; wd + 1 < a.nan_row_words; wd += 64) {
                unsigned b = 0u;
                for (int k = 0; k < 32 && 32 * wd + k < a.N; ++k) b |= (any_nan && wallnan[32 * wd + k]) ? (1u << k) : 0u;
                dst[1 + wd] = b;
            }
        }
    } else {
    if (cell_nan && in_range) {
        a.grad[2 * idx] = qnan;
        a.grad[2 * idx + 1] = qnan;
    }
    if (any_nan && a.partial != nullptr) {
        __syncthreads();
        float* dst = a.partial + tile * (4 * a.N + 2);  // the patch's row (fwd_patch / txg_patch / power_vg_kernel)
        for (int i = lane; i < a.N; i += 64)
            if (wallnan[i]) dst[4 * i] = dst[4 * i + 1] = dst[4 * i + 2] = dst[4 * i + 3] = qnan;
        if (lane == 0) dst[4 * a.N] = dst[4 * a.N + 1] = qnan;
    }
    }
    if (stats && lane == 0) {
        atomicAdd(&stats[0], n_probe);
        atomicAdd(&stats[1], (unsigned long long)__builtin_popcountll(flagged));
        if (any_nan) atomicAdd(&stats[2], 1ull);
    }
}

// ---- the same scan, two levels ------------------------------------------------------------------------------------------
// One workgroup of NAN_W = 16 waves per region of 4 x 4 patches.  A zero line that crosses a 32 x 32-cell region is ~4 x as
// likely as one that crosses an 8 x 8 patch, but the region-level test is shared by 16 patches: the batches of an order (prefix,
// chunk of last walls) are dealt to the waves round-robin and tested against the REGION's box, the survivors go to a list in
// LDS (unordered: the flags are OR-ed), and every wave then tests only that list against its own patch and probes what is left.
// Rounds of NAN_LCAP / 64 batches: the list cannot overflow, whatever survives (non-finite coordinates: everything does).
// (NAN_W, NAN_R, NAN_LCAP, NAN_RB: d2d_kernels.hpp)

// batch `id` of order K: prefix positions (into cw[], no equal neighbours, lexicographic) and the chunk of last walls
template <int K>
__device__ __forceinline__ void nan_decode_batch(long long id, int Nc, int n_chunks, int (&pos)[D2D_MAX_ORDER], int& chunk) {
    if (K == 1) {
        chunk = (int)id;
        return;
    }
    long long L;
    if (id < 0x7fffffffLL) {
        const unsigned u = (unsigned)id;
        chunk = (int)(u % (unsigned)n_chunks);
        L = (long long)(u / (unsigned)n_chunks);
    } else {
        chunk = (int)(id % n_chunks);
        L = id / n_chunks;
    }
    int dig[D2D_MAX_ORDER] = {0, 0, 0, 0};
#pragma unroll
    for (int d = K - 2; d >= 1; --d) {
        if (L < 0x7fffffffLL) {
            const unsigned u = (unsigned)L;
            dig[d] = (int)(u % (unsigned)(Nc - 1));
            L = (long long)(u / (unsigned)(Nc - 1));
        } else {
            dig[d] = (int)(L % (Nc - 1));
            L = L / (Nc - 1);
        }
    }
    pos[0] = (int)L;
#pragma unroll
    for (int d = 1; d <= K - 2; ++d) pos[d] = dig[d] + (dig[d] >= pos[d - 1] ? 1 : 0);
}

template <int K, bool APPROX, bool TXG, bool DBG>
__device__ __forceinline__ void nan_region_order(const SweepArgs& a, const float4* tab, unsigned long long* list, int* lcount, unsigned* wallbits,
                                                 unsigned long long* wq, int* wqn, const float2* cells, unsigned long long* cellmask, unsigned* wallbits_all, int nwords,
                                                 const unsigned* nearbits, const unsigned* region_near, const float (&rbx)[4], const float (&rby)[4], const float (&pbx)[4], const float (&pby)[4],
                                                 float cx, float cy, bool force, bool patch_exists, bool& cell_nan, bool& any_nan,
                                                 unsigned long long& n_probe, int& round, unsigned* dbg_counts) {  // (dbg_counts: DBG builds only)
    static_assert(K >= 1, "order 0 has no scan");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Entries of the probe queue / batches per round IN USE.  Product build (DBG = false): the compile-time sizes.  DBG = true -- chosen
    // by the host when a test asked for smaller buffers ("nan_scan_wqcap" / "nan_scan_rb": a full queue and a full list as the rule)
    // or for the counters ("nan_scan_stats") --: a.nan_wqcap in [1, NAN_WQCAP], a.nan_rb in [1, NAN_RB], and the counters through
    // LDS.  Same code either way; as run-time values in the one kernel they cost the hard-mode scan 8 % (measured, round 6).
    const int rb = DBG ? a.nan_rb : NAN_RB;
    const int wqcap = DBG ? a.nan_wqcap : NAN_WQCAP;
    if (!DBG) dbg_counts = nullptr;
    const bool plen = a.fun_id != D2D_FUN_ONE;
    const int Nc = a.Nc;
    if (Nc < 1 || (K >= 2 && Nc < 2)) return;  // (workgroup-uniform)
    const int n_chunks = (Nc + 63) >> 6;
    long long nb = n_chunks;
    if (K >= 2) {
        nb *= Nc;
#pragma unroll
        for (int d = 1; d <= K - 2; ++d) nb *= (Nc - 1);
    }
    // the exact probe of one (patch, candidate) item, lanes = the cells of patch p; flags go to the patch's words in LDS
    auto probe_item = [&](unsigned long long cu, int p, float pcx, float pcy) {
        int ce[D2D_MAX_ORDER] = {-1, -1, -1, -1};
        float ex[D2D_MAX_ORDER], ey[D2D_MAX_ORDER];
        // (wave-uniform) an item names a patch of this region and K objects of the scene.  Whatever it holds, it never becomes an
        // address outside the tables: the patch index is taken modulo NAN_W, an object index is capped at N - 1 (one scalar min
        // each; the scene's tables are a few KB and an index of 4095 would read 128 KB past them).  With the counters on
        // ("nan_scan_stats": the tests) an item that NEEDED the cap is refused and counted instead -- an internal error, always 0.
        if (dbg_counts != nullptr) {
            bool item_ok = (unsigned)p < (unsigned)NAN_W;
#pragma unroll
            for (int d = 0; d < K; ++d) item_ok = item_ok && (int)((cu >> (12 * d)) & 0xfffull) < a.N;
            if (!item_ok) {
                if (lane == 0) atomicAdd(&dbg_counts[1], 1u);
                return;
            }
        }
        p &= NAN_W - 1;
#pragma unroll
        for (int d = 0; d < K; ++d) {
            ce[d] = min((int)((cu >> (12 * d)) & 0xfffull), a.N - 1);
            // RX grids: the transmitter's image chain (wave-uniform); TX grids: the lane's cell's (geometry.py:1086-1091)
            image_of(ldc4(a.refl, 2 * ce[d]), d == 0 ? (TXG ? pcx : a.txx) : ex[d > 0 ? d - 1 : 0], d == 0 ? (TXG ? pcy : a.txy) : ey[d > 0 ? d - 1 : 0],
                     ex[d], ey[d]);
        }
        const bool z = TXG ? nan_probe<K, APPROX>(a, ce, ex, ey, pcx, pcy, a.txx, a.txy) : nan_probe<K, APPROX>(a, ce, ex, ey, a.txx, a.txy, pcx, pcy);
        const unsigned long long zm = __ballot(z);
        if (zm != 0ull && lane == 0) {
            atomicOr(&cellmask[p], zm);
#pragma unroll
            for (int d = 0; d < K; ++d) atomicOr(&wallbits_all[p * nwords + (ce[d] >> 5)], 1u << (ce[d] & 31));
        }
    };
    for (long long r0 = 0; r0 < nb; r0 += rb, ++round) {  // (workgroup-uniform)
        int* const cnt_p = lcount + (round & 1);
        if (threadIdx.x == 0) *wqn = 0;  // (the previous round's queue was emptied before its last barrier)
#ifdef D2D_NAN_QUEUE_R5
        for (int i = threadIdx.x; i < NAN_WQCAP; i += 64 * NAN_W) wq[i] = ~0ull;  // (probe build: the sentinel; ordered by the barrier below)
#endif
        // ---- region level: batches [r0, r0 + rb) of the order, batch r0 + t * NAN_W + wv to wave wv: at most 64 rb <= NAN_LCAP
        // survivors per round, the list cannot overflow
        const int nbr = (nb - r0 < (long long)rb) ? (int)(nb - r0) : rb;  // batches of this round
#pragma unroll 1
        for (int t = 0; t < (NAN_RB + NAN_W - 1) / NAN_W; ++t) {
            if (t * NAN_W + wv >= nbr) break;
            const long long id = r0 + (long long)t * NAN_W + wv;
            int pos[D2D_MAX_ORDER] = {0, 0, 0, 0};
            int chunk = 0;
            nan_decode_batch<K>(id, Nc, n_chunks, pos, chunk);
            int cand[D2D_MAX_ORDER] = {-1, -1, -1, -1};
            float imgx[D2D_MAX_ORDER], imgy[D2D_MAX_ORDER];
            WallC wu[K];
#pragma unroll
            for (int d = 0; d < K - 1; ++d) {
                cand[d] = cmem(a.cw)[pos[d]];
                const float4 r0_ = ldc4(a.refl, 2 * cand[d]);
                wu[d] = make_wallc(r0_, ldc4(a.refl, 2 * cand[d] + 1), ldc4(a.flt, cand[d]), cand[d]);
                if (!TXG) image_of(r0_, d == 0 ? a.txx : imgx[d > 0 ? d - 1 : 0], d == 0 ? a.txy : imgy[d > 0 ? d - 1 : 0], imgx[d], imgy[d]);
            }
            const float pIx = (K == 1) ? a.txx : imgx[K >= 2 ? K - 2 : 0];
            const float pIy = (K == 1) ? a.txy : imgy[K >= 2 ? K - 2 : 0];
            const int last_prefix_pos = (K == 1) ? -1 : pos[K >= 2 ? K - 2 : 0];
            const int lp = chunk * 64 + lane;
            bool alive = (lp < Nc) && (lp != last_prefix_pos);
            const int wl = cmem(a.cw)[lp < Nc ? lp : 0];
            if (!force) {
                const float4 q0 = tab[2 * wl], q1 = tab[2 * wl + 1], fc = tab[2 * a.N + wl];
                WallC w[K];
#pragma unroll
                for (int d = 0; d < K - 1; ++d) w[d] = wu[d];
                w[K - 1] = make_wallc(q0, q1, fc, wl);
                if constexpr (TXG) {
                    if (alive) alive = nan_possible_txg<K, APPROX>(rbx, rby, w, a.txx, a.txy, plen);
                } else {
                    float Ix[K], Iy[K];
#pragma unroll
                    for (int d = 0; d < K - 1; ++d) {
                        Ix[d] = imgx[d];
                        Iy[d] = imgy[d];
                    }
                    image_of(q0, pIx, pIy, Ix[K - 1], Iy[K - 1]);
                    if (alive) alive = nan_possible_rx<K, APPROX>(rbx, rby, w, Ix, Iy, a.txx, a.txy, plen, (int)((region_near[wl >> 5] >> (wl & 31)) & 1u));
                }
            }
            const unsigned long long mask = __ballot(alive);
            const int cnt = __builtin_popcountll(mask);
            if (cnt) {
                int base = 0;
                if (lane == 0) base = atomicAdd(cnt_p, cnt);
                base = __builtin_amdgcn_readfirstlane(base);
                if (alive) {
                    unsigned long long code = (unsigned long long)wl << (12 * (K - 1));
#pragma unroll
                    for (int d = 0; d < K - 1; ++d) code |= (unsigned long long)cand[d] << (12 * d);
                    list[base + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = code;
                }
            }
        }
        __syncthreads();
        const int n = *cnt_p;
        if (threadIdx.x == 0) lcount[(round + 1) & 1] = 0;  // (the other counter: nobody uses it before the barrier below)
        // ---- patch level: the region's survivors against this wave's own patch, then the cells
        if (patch_exists) {
            for (int off = 0; off < n; off += 64) {
                const bool have = off + lane < n;
                const unsigned long long code = list[have ? off + lane : off];
                bool alive = have;
                if (!force) {
                    WallC w[K];
                    float Ix[K], Iy[K];
                    float ix = a.txx, iy = a.txy;
#pragma unroll
                    for (int d = 0; d < K; ++d) {
                        const int wd = (int)((code >> (12 * d)) & 0xfffull);
                        const float4 q0 = tab[2 * wd], q1 = tab[2 * wd + 1], fc = tab[2 * a.N + wd];
                        w[d] = make_wallc(q0, q1, fc, wd);
                        if (!TXG) {
                            image_of(q0, ix, iy, Ix[d], Iy[d]);
                            ix = Ix[d];
                            iy = Iy[d];
                        }
                    }
                    if constexpr (TXG) {
                        if (alive) alive = nan_possible_txg<K, APPROX>(pbx, pby, w, a.txx, a.txy, plen);
                    } else {
                        if (alive) alive = nan_possible_rx<K, APPROX>(pbx, pby, w, Ix, Iy, a.txx, a.txy, plen, (int)((nearbits[w[K - 1].idx >> 5] >> (w[K - 1].idx & 31)) & 1u));
                    }
                }
                unsigned long long mask = __ballot(alive);
                const int cnt = __builtin_popcountll(mask);
                n_probe += (unsigned long long)cnt;
                if (cnt) {
                    // the probes of a region are dealt to ALL its waves (the patches a zero line crosses are few, and the slowest wave
                    // of a region probed 1.45 - 1.7 x the average): (patch, candidate) items go to a queue of the workgroup.  The
                    // reservation is MONOTONE -- the counter only grows, no wave ever takes its claim back -- so that a slot below the
                    // cap belongs to exactly one lane whatever the interleaving, and the slots below min(counter, cap) have all been
                    // written once the barrier is passed.  (Round 5 reserved and rolled back when the queue was full: a successful
                    // reservation between another wave's add and its subtraction left the final count covering slots nobody wrote.)
#ifdef D2D_NAN_QUEUE_R5
                    // PROBE BUILD ONLY (scripts/nan_queue_race.py): round 5's reserve-and-roll-back, kept to SHOW its race -- the
                    // queue is filled with a sentinel every round (below), and a drained sentinel is a slot the final count covers
                    // but nobody wrote; the item check in probe_item refuses it and counts it in stats[4].
                    int base = 0;
                    if (lane == 0) {
                        base = atomicAdd(wqn, cnt);
                        if (base + cnt > wqcap) {
                            atomicSub(wqn, cnt);
                            base = -1;
                        }
                    }
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= 0) {
                        if (alive) wq[base + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = code | ((unsigned long long)wv << 56);
                        mask = 0ull;
                    }
                    if (dbg_counts != nullptr && lane == 0 && mask != 0ull) atomicAdd(&dbg_counts[0], (unsigned)__builtin_popcountll(mask));
#else
                    int base = 0;
                    if (lane == 0) base = atomicAdd(wqn, cnt);
                    base = __builtin_amdgcn_readfirstlane(base);
                    const int slot = base + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                    const bool queued = alive && slot < wqcap;
                    if (queued) wq[slot] = code | ((unsigned long long)wv << 56);
                    mask = __ballot(alive && !queued);
                    if (dbg_counts != nullptr && lane == 0 && mask != 0ull) atomicAdd(&dbg_counts[0], (unsigned)__builtin_popcountll(mask));
#endif
                }
                // ... unless it is full: then the wave probes its own
                while (mask) {
                    const int b = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    const unsigned lo32 = (unsigned)__builtin_amdgcn_readlane((int)(code & 0xffffffffull), b);
                    const unsigned hi32 = (K >= 3) ? (unsigned)__builtin_amdgcn_readlane((int)(code >> 32), b) : 0u;
                    const unsigned long long cu = ((unsigned long long)hi32 << 32) | lo32;
                    probe_item(cu, wv, cx, cy);
                }
            }
        }
        __syncthreads();
        {
            // ---- the queue: item i to wave i mod NAN_W, the cells of the item's patch from LDS
            const int nq_all = *wqn;
            const int nq = nq_all < wqcap ? nq_all : wqcap;  // (the counter keeps counting past the cap: those lanes probed their own)
            for (int i = wv; i < nq; i += NAN_W) {
                const unsigned long long it = wq[i];
                const int p = (int)(it >> 56);
                const float2 c = cells[p * 64 + lane];
                probe_item(it & 0x00ffffffffffffffull, p, c.x, c.y);
            }
        }
        __syncthreads();  // the list is rewritten by the next round
    }
}

#ifndef D2D_NAN_MIN_WAVES
#define D2D_NAN_MIN_WAVES 1  // A/B: waves per SIMD the region scan must leave room for (8: two of its workgroups per CU, <= 64 VGPRs)
#endif
template <bool APPROX, bool TXG, int MAXK, bool DBG = false>
__global__ void __launch_bounds__(64 * NAN_W, D2D_NAN_MIN_WAVES) nan_scan_region_kernel(SweepArgs a, unsigned long long* __restrict__ stats) {
    extern __shared__ float4 tab[];  // [2N] refl, [N] flt, the region's list [NAN_LCAP], then [2 NAN_W + 1][ceil(N / 32)] flag bits
    __shared__ float pbox[NAN_W][4];
    __shared__ int lcount[2];
    __shared__ int sbad;
    __shared__ unsigned long long swq[NAN_WQCAP];    // the region's probes: (patch, candidate) items dealt to all waves (nan_region_order)
    __shared__ int swqn;
    __shared__ float2 scells[NAN_W * 64];            // the region's cells, patch by patch
    __shared__ unsigned long long scellmask[NAN_W];  // flagged cells per patch
    __shared__ unsigned sdbg[2];                     // counters of the stats build (nan_region_order)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 2 * a.N; i += 64 * NAN_W) tab[i] = ldc4(a.refl, i);
    for (int i = threadIdx.x; i < a.N; i += 64 * NAN_W) tab[2 * a.N + i] = ldc4(a.flt, i);
    unsigned long long* list = reinterpret_cast<unsigned long long*>(tab + 3 * a.N);
    const int nwords = (a.N + 31) >> 5;
    unsigned* wallbits_all = reinterpret_cast<unsigned*>(list + NAN_LCAP);
    for (int i = threadIdx.x; i < (2 * NAN_W + 1) * nwords; i += 64 * NAN_W) wallbits_all[i] = 0u;
    // [NAN_W][nwords] per patch, then [nwords] of the region: walls whose line some CELL is within the near-coincidence bound of
    unsigned* nearbits_all = wallbits_all + NAN_W * nwords;
    if (threadIdx.x == 0) {
        lcount[0] = lcount[1] = 0;
        sbad = 0;
        swqn = 0;
        sdbg[0] = sdbg[1] = 0u;
    }
    if (threadIdx.x < NAN_W) scellmask[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned* wallbits = wallbits_all + wv * nwords;
    const int tiles_x = (a.n + TILE_W - 1) / TILE_W, tiles_y = (a.m + TILE_H - 1) / TILE_H;
    const int regions_x = (tiles_x + NAN_R - 1) / NAN_R;
    const int rx = (int)(blockIdx.x % regions_x), ry = (int)(blockIdx.x / regions_x);
    const int tcol = rx * NAN_R + (wv & (NAN_R - 1)), trow = ry * NAN_RY + (wv / NAN_R);
    const bool patch_exists = tcol < tiles_x && trow < tiles_y;
    const long tile = (long)trow * tiles_x + tcol;
    const int col = tcol * TILE_W + (lane & (TILE_W - 1));
    const int row = trow * TILE_H + (lane / TILE_W);
    const bool in_range = (col < a.n) && (row < a.m);
    const int ccol = col < a.n ? col : a.n - 1;  // (a patch beyond the grid's edge repeats cells of this region's last column / row)
    const int crow = row < a.m ? row : a.m - 1;
    const long idx = (long)crow * a.n + ccol;
    const float cx = a.X[idx], cy = a.Y[idx];
    scells[wv * 64 + lane] = make_float2(cx, cy);
    const bool lane_bad = !(fabsf(cx) < 1e18f) || !(fabsf(cy) < 1e18f) || !(fabsf(a.txx) < 1e18f) || !(fabsf(a.txy) < 1e18f);
    float x0 = cx, x1 = cx, y0 = cy, y1 = cy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, off, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, off, 64));
        y0 = fminf(y0, __shfl_xor(y0, off, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
    }
    const float pbx[4] = {x0, x1, x1, x0};
    const float pby[4] = {y0, y0, y1, y1};
    if (lane == 0) {
        pbox[wv][0] = x0; pbox[wv][1] = x1; pbox[wv][2] = y0; pbox[wv][3] = y1;
    }
    if (wave_any(lane_bad) && lane == 0) atomicOr(&sbad, 1);
    unsigned* nearbits = nearbits_all + wv * nwords;
    unsigned* region_near = nearbits_all + NAN_W * nwords;
    if (!TXG) {
        // The last segment of a path (interaction point on the candidate's last wall -> the cell) can only vanish, or be the
        // (eps, eps) of rule (3), for a cell within a few ulps (+ NAN_ABS) of that wall's line: |vn| <= dv in nan_possible_rx's
        // terms.  A box is crossed by a line far more often than a cell sits on it: the cells themselves are asked, once per
        // wall (lanes = cells), and the candidate tests look the answer up.
        const float eps = 1.1920929e-07f;
        const float abs3 = (a.fun_id != D2D_FUN_ONE) ? NAN_ABS : 0.0f;
        for (int w0 = 0; w0 < a.N; ++w0) {
            const float4 q = ldc4(a.refl, 2 * w0);
            const float vn = __builtin_fmaf(q.x - cx, q.z, (q.y - cy) * q.w);
            const float dv = 32.0f * eps * ((fabsf(cx) + fabsf(cy)) + (fabsf(q.x) + fabsf(q.y))) + abs3;
            if (wave_any(!(fabsf(vn) > dv)) && lane == 0) {
                nearbits[w0 >> 5] |= 1u << (w0 & 31);
                atomicOr(&region_near[w0 >> 5], 1u << (w0 & 31));
            }
        }
    }
    __syncthreads();
    float r0 = pbox[lane & (NAN_W - 1)][0], r1 = pbox[lane & (NAN_W - 1)][1], r2 = pbox[lane & (NAN_W - 1)][2], r3 = pbox[lane & (NAN_W - 1)][3];
#pragma unroll
    for (int off = NAN_W / 2; off > 0; off >>= 1) {
        r0 = fminf(r0, __shfl_xor(r0, off, 64));
        r1 = fmaxf(r1, __shfl_xor(r1, off, 64));
        r2 = fminf(r2, __shfl_xor(r2, off, 64));
        r3 = fmaxf(r3, __shfl_xor(r3, off, 64));
    }
    const float rbx[4] = {r0, r1, r1, r0};
    const float rby[4] = {r2, r2, r3, r3};
    // a coordinate that is not comfortably finite anywhere in the region: no bound holds, every candidate is probed
    const bool force = sbad != 0;
    bool cell_nan = false, any_nan = false;
    unsigned long long n_probe = 0ull;
    unsigned* const dbg_counts = (DBG && stats != nullptr) ? sdbg : nullptr;  // [0] self probes, [1] refused items: through LDS, DBG builds only
    int round = 0;
    if (a.min_order <= 0 && a.max_order >= 0 && a.fun_id != D2D_FUN_ONE) {
        // order 0 has no scan, but its one segment has rule (3): px[0] = transmitter, px[1] = receiver
        cell_nan = TXG ? (((a.txx - cx) + D2D_EPS == 0.0f) && ((a.txy - cy) + D2D_EPS == 0.0f)) : (((cx - a.txx) + D2D_EPS == 0.0f) && ((cy - a.txy) + D2D_EPS == 0.0f));
        any_nan = wave_any(cell_nan);
    }
    if (a.min_order <= 1 && a.max_order >= 1)
        nan_region_order<1, APPROX, TXG, DBG>(a, tab, list, lcount, wallbits, swq, &swqn, scells, scellmask, wallbits_all, nwords, nearbits, region_near, rbx, rby, pbx, pby, cx, cy, force, patch_exists, cell_nan, any_nan, n_probe, round, dbg_counts);
    if (a.min_order <= 2 && a.max_order >= 2)
        nan_region_order<2, APPROX, TXG, DBG>(a, tab, list, lcount, wallbits, swq, &swqn, scells, scellmask, wallbits_all, nwords, nearbits, region_near, rbx, rby, pbx, pby, cx, cy, force, patch_exists, cell_nan, any_nan, n_probe, round, dbg_counts);
    if (MAXK >= 3 && a.min_order <= 3 && a.max_order >= 3)
        nan_region_order<3, APPROX, TXG, DBG>(a, tab, list, lcount, wallbits, swq, &swqn, scells, scellmask, wallbits_all, nwords, nearbits, region_near, rbx, rby, pbx, pby, cx, cy, force, patch_exists, cell_nan, any_nan, n_probe, round, dbg_counts);
    if (MAXK >= 4 && a.min_order <= 4 && a.max_order >= 4)
        nan_region_order<4, APPROX, TXG, DBG>(a, tab, list, lcount, wallbits, swq, &swqn, scells, scellmask, wallbits_all, nwords, nearbits, region_near, rbx, rby, pbx, pby, cx, cy, force, patch_exists, cell_nan, any_nan, n_probe, round, dbg_counts);
    {
        // what the region's probes found in this wave's patch (nan_region_order ends on a barrier)
        const unsigned long long cm = scellmask[wv];
        cell_nan = cell_nan || (((cm >> lane) & 1ull) != 0ull);
        any_nan = any_nan || (cm != 0ull);
    }
    const float qnan = __builtin_nanf("");
    const unsigned long long flagged = __ballot(cell_nan && in_range && patch_exists);
    if (a.nan_cell_bits != nullptr) {
        // beside the sweep: leave the flags for nan_apply_kernel (every patch writes its words: nothing to zero beforehand)
        if (patch_exists) {
            if (lane == 0) a.nan_cell_bits[tile] = flagged;
            if (a.nan_row_bits != nullptr) {
                __builtin_amdgcn_wave_barrier();
                unsigned* dst = a.nan_row_bits + tile * a.nan_row_words;
                if (lane == 0) dst[0] = any_nan ? 1u : 0u;
                for (int wd = lane; wd + 1 < a.nan_row_words; wd += 64) dst[1 + wd] = any_nan ? wallbits[wd] : 0u;
            }
        }
    } else {
    if (cell_nan && in_range && patch_exists) {
        a.grad[2 * idx] = qnan;
        a.grad[2 * idx + 1] = qnan;
    }
    if (any_nan && patch_exists && a.partial != nullptr) {
        __builtin_amdgcn_wave_barrier();
        float* dst = a.partial + tile * (4 * a.N + 2);  // the patch's row (fwd_patch / txg_patch / power_vg_kernel)
        for (int i = lane; i < a.N; i += 64)
            if ((wallbits[i >> 5] >> (i & 31)) & 1u) dst[4 * i] = dst[4 * i + 1] = dst[4 * i + 2] = dst[4 * i + 3] = qnan;
        if (lane == 0) dst[4 * a.N] = dst[4 * a.N + 1] = qnan;
    }
    }
    if (stats && lane == 0) {
        atomicAdd(&stats[0], n_probe);
        atomicAdd(&stats[1], (unsigned long long)__builtin_popcountll(flagged));
        if (any_nan && patch_exists) atomicAdd(&stats[2], 1ull);
        if (wv == 0) {  // (nan_region_order ends on a barrier: the workgroup's counters are complete)
            atomicAdd(&stats[3], (unsigned long long)sdbg[0]);  // probes a wave made itself because the region's queue was full
            atomicAdd(&stats[4], (unsigned long long)sdbg[1]);  // queue items refused because they named no patch / object (must stay 0)
            atomicAdd(&stats[5], (unsigned long long)round);    // rounds (list fills) of this region
        }
    }
}

// What a scan that ran BESIDE the sweep found, applied once both are through: one wave per patch (four to a workgroup); the flagged
// cells get their NaN gradient, the patch's row of VJP partial sums a NaN for the fixed end point and the flagged objects --
// exactly what the scan writes itself when it runs behind the sweep.
__global__ void __launch_bounds__(256) nan_apply_kernel(float* __restrict__ grad, float* __restrict__ partial,
                                                        const unsigned long long* __restrict__ cell_bits, const unsigned* __restrict__ row_bits,
                                                        int row_words, int N, int m, int n, long tiles) {
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    const unsigned long long bits = cell_bits[tile];
    const float qnan = __builtin_nanf("");
    if (bits != 0ull) {
        const int tiles_x = (n + TILE_W - 1) / TILE_W;
        const int col = (int)(tile % tiles_x) * TILE_W + (lane & (TILE_W - 1));
        const int row = (int)(tile / tiles_x) * TILE_H + (lane / TILE_W);
        if (((bits >> lane) & 1ull) && col < n && row < m) {
            const long idx = (long)row * n + col;
            grad[2 * idx] = qnan;
            grad[2 * idx + 1] = qnan;
        }
    }
    if (partial != nullptr && row_bits != nullptr) {
        const unsigned* src = row_bits + tile * row_words;
        if (src[0] != 0u) {
            float* dst = partial + tile * (4 * N + 2);
            for (int i = lane; i < N; i += 64)
                if ((src[1 + (i >> 5)] >> (i & 31)) & 1u) dst[4 * i] = dst[4 * i + 1] = dst[4 * i + 2] = dst[4 * i + 3] = qnan;
            if (lane == 0) dst[4 * N] = dst[4 * N + 1] = qnan;
        }
    }
}

}  // namespace d2d
