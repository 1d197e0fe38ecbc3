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
//   * a candidate is never evaluated at all when conservative geometry proves that exact zero for a whole 8 x 8 patch:
//     the tile culling (cull_candidate), the first-segment shadow masks (shadow_tx_kernel / shadow_fill_kernel), the
//     wall-to-wall masks (pair_shadow_kernel) and the last-segment masks of the leaf regions (hidden_region_kernel), each
//     with explicit rounding bounds.
//
// File map: SweepArgs / eval_candidate (exact evaluation + hand-derived adjoint) / sweep_order (plain nested loops) /
// s_range, cull_candidate, sweep_order_culled (prefix odometer, two-stage culling, survivors in candidate order) /
// region lists (EmitSink, cull_batch, sweep_order_listed, region_list_kernel, region_refine_kernel) /
// power_fwd_kernel (one wave per patch; dearest patches cut in four) / power_fwd_split_kernel (small launches: every
// patch shared by 4 waves prefix by prefix) / power_fwd_coop_kernel (the smallest: 4 / 8 / 16 waves, candidate by candidate) /
// sweep_order_culled_txg + power_fwd_txg_kernel (TX grids) / patch_* (dearest-first schedule) / shadow_fill_kernel,
// pair_shadow_kernel, hidden_region_kernel (occlusion masks) / power_vg_kernel (exhaustive value+grad) / trace_kernel and
// the literal object code (any mix of Wall / RIS / Vertex) / power_opt_*_kernel (MinPath / FermatPath sweeps).
//
// Reference lines followed (DiffeRT2d v0.4.0): see include/d2d.h and oracle/ref.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/d2d.h"

namespace d2d {

// Tables that no kernel writes while it reads them (scene constants, masks and lists built by the kernels in front) are
// read through the constant address space: a wave-uniform index then always becomes a scalar load (s_load, scalar
// cache), also behind stores the compiler cannot tell apart from the table (a plain global load is only scalarised when
// nothing in front of it may have clobbered it).
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* cmem(const T* p) {
    return (const __attribute__((address_space(4))) T*)(unsigned long long)p;
}

// (class types such as float4 cannot be copied out of another address space: go through the built-in vector types)
__device__ __forceinline__ float4 ldc4(const float4* p, long i) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = ((const __attribute__((address_space(4))) v4f*)(unsigned long long)p)[i];
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int4 ldc4i(const int4* p, long i) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i v = ((const __attribute__((address_space(4))) v4i*)(unsigned long long)p)[i];
    return make_int4(v.x, v.y, v.z, v.w);
}

// A field of the kernel's own argument block, read WHERE IT IS NEEDED (the compiler loads every argument at the top of a
// kernel and keeps it in a scalar register until its last use: the pointers a sweep only needs for its final stores -- map,
// work counters, the cut patches' hand-over -- then occupy registers through the whole candidate loop, and what does not fit
// is parked in VGPR lanes: 82 parked scalars in the hot kernel before, 58 with these late reads).  Valid in kernels whose
// FIRST parameter is the struct, by value (kernarg offset 0).
template <typename T, unsigned OFF>
__device__ __forceinline__ T late_kernarg() {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "one or two dwords");
    const auto base = __builtin_amdgcn_kernarg_segment_ptr();
    T v;
    if constexpr (sizeof(T) == 8) asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "i"(OFF));
    else asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(base), "i"(OFF));
    return v;
}
#define D2D_LATE_ARG(T, field) late_kernarg<T, (unsigned)__builtin_offsetof(SweepArgs, field)>()

enum Mode { MODE_HARD = 0, MODE_HSIG = 1, MODE_SIG = 2 };

// Region candidate lists: for every order K >= 2, every region of R x R patches and every slice of first-wall positions,
// the candidates that the tile culling cannot prove invalid for the region's bounding box, in candidate order.  The
// patches of a region then test and evaluate those instead of enumerating all prefixes themselves.  Two levels: big
// regions are listed by enumeration (region_list_kernel), the regions the sweep kernels read by filtering their parent's
// lists (region_refine_kernel).  A list is a chain of 128-entry chunks of one pool; an entry holds 12 bits per wall
// index, first wall lowest (bit 60: see sweep_order_culled).
constexpr int RL_CHUNK = 128;
struct ListPool {
    unsigned long long* pool;  // [max_chunks][RL_CHUNK]
    int* next;                 // [max_chunks] the chunk that continues a chunk
    int* head;                 // continuation chunks handed out so far (may run past the pool: those lists are not listed)
    int n_static;              // chunks [0, n_static) are the lists' first chunks; continuation chunks follow
    int max_chunks;
};
struct RegionLevel {
    int* cnt[D2D_MAX_ORDER + 1];    // [K]: [regions][S] entries; < 0: not listed (pool exhausted, non-finite cell)
    int chunk0[D2D_MAX_ORDER + 1];  // [K]: list i of order K starts in chunk chunk0[K] + i
    int S;                          // lists per region: slices of first-wall positions (first_wall_range(s, S)); 1 at the leaf level
    int R;                          // a region is R x R patches
    int regions_x, regions_y;
    const float4* box;              // [regions] bounding boxes of the regions' cells (region_box_kernel)
    // [regions][N] last-segment masks (hidden_region_kernel; leaf level only, null: none): bit b of word [r][w] = the segment
    // from any point within hidden_dperp of bin b of wall w to ANY point of region r's box is certainly reported as
    // intersecting some object by the exact path.  Scene, grid and validity mode only: no end point involved.
    const unsigned long long* hidden;
    float hidden_dperp;
};
struct RegionLists {
    RegionLevel leaf;  // the level the sweep kernels read (one list per region and order)
    ListPool lp;
    int* flag;         // [leaf regions] != 0: some list of the region is not listed
};

struct SweepArgs {
    const RegionLists* __restrict__ rl;  // LISTED kernels: device copy of the lists' descriptor
    // Patches a LISTED kernel cannot take (a list of their region is not listed, or a cell is not comfortably finite) are
    // queued here and swept by the enumerating kernel launched right behind it (fb_n != null there: workgroups walk the queue)
    int* fb_n;
    int* fb_list;
    // scene tables (device, read-only, wave-uniform indexing -> scalar loads)
    const float4* __restrict__ occl;  // [N]  {p1x, p1y, Ax, Ay}: patched origin and P2-P1 (geometry.py:632-636)
    const float4* __restrict__ refl;  // [2N] {ox, oy, nx, ny}, {tx, ty, sq, 0}: reflection data
    const float4* __restrict__ flt;   // [N]  {1/sq, margin * 1/sq, 0, 0}: constants of the on_objects pre-filter
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
    float on_lo, on_hi;        // parametric coordinate certainly outside the wall: s < on_lo or s > on_hi => on_objects == 0
    float loss_skip;           // a loss certainly below this cannot change less(loss, tol) (see eval_candidate); < 0: never
    float sig_l2f;             // MODE_SIG: log2 of an upper bound of |fun| over the launch's orders, 1e30: none (sig_zc_of)
    int sig_mono;              // MODE_SIG: fun >= 0 throughout, so a cell's running sum never shrinks
    // first-segment shadow culling (shadow_tx_kernel): bit b of shadow[w] = every point of wall w with parametric
    // coordinate in [b/64, (b+1)/64] (and within shadow_dperp of the wall's line) is certainly hidden from the fixed
    // end point by some other object
    const unsigned long long* __restrict__ shadow;  // [N] or null
    float shadow_dperp;
    float shadow_lo, shadow_inv;  // bin b covers parametric coordinates shadow_lo + [b, b+1] / shadow_inv
    int shadow_prefix_ok;         // a fully covered first wall kills its whole prefix (see sweep_order_culled)
    // wall-to-wall masks (pair_shadow_kernel): bit (be + 8 * bl) of pair[we * N + wl] = every segment from a point in
    // bin be of wall we (the earlier interaction, P3) to a point in bin bl of wall wl (the later one, P4) -- 8 bins over
    // the same parametric window as the 64 shadow bins -- is certainly occluded by some third object
    const unsigned long long* __restrict__ pair;  // [N * N] or null
    float pair_dperp;
    int pair_prefix_ok;  // two consecutive prefix walls that cannot see each other at all kill the prefix (orders >= 3)
    float fnum[D2D_MAX_ORDER + 1];  // r_coef ** k (lax.integer_pow), k = 0..D2D_MAX_ORDER
    float h2;                  // height * height
    int fun_id;
    int out_mode;
    float patch;               // geometry.py:916 (gradient kernel: chain rule through the patched end points)
    // value+grad kernel only
    float* __restrict__ grad;        // [m][n][2] d Z / d rx per cell
    const float* __restrict__ cot;   // [m][n] cotangent for the scene VJP, or null (= ones)
    float* __restrict__ partial;     // [n_waves][4 N + 2] per-wave partial sums of the scene VJP, or null
    // fun_id == D2D_FUN_CUSTOM (exhaustive value+grad kernel only): a path function the host evaluated on the traced paths --
    // its values and its derivatives w.r.t. the path's points, per candidate (the sweep's own order) and cell
    const float* __restrict__ cust_f;   // [C][m * n]
    const float* __restrict__ cust_pb;  // [C][m * n][D2D_MAX_ORDER + 2][2]
    long cust_cells;                    // m * n

    // patch schedule (patch_cost_kernel / patch_order_kernel): workgroup b takes patch sched[b]; null = identity
    const int* __restrict__ sched;
    int cullq_off;  // byte offset, from the start of dynamic LDS, of the culling queues (512 B per wave of the workgroup)
    // Dearest patches cut in four (power_fwd_kernel, max_order == 2, needs the schedule): workgroups [0, 4 * n_heavy)
    // are the quarters of patches sched[0 .. n_heavy), the others take sched[n_heavy ..) one patch each
    int n_heavy;
    int heavy_cap;                    // list entries per lane and quarter (>= the candidates a quarter can evaluate)
    float* __restrict__ heavy_list;   // [n_heavy][4][heavy_cap][64] non-zero contributions in candidate order
    int* __restrict__ heavy_cnt;      // [n_heavy][4][64] entries per lane, then [n_heavy][4] work of the quarter
    int* __restrict__ heavy_done;     // [n_heavy] quarters finished (zero between launches)
    unsigned* __restrict__ cost_out;  // [n_patches] work this patch took (feeds the next launch's schedule), or null

    // NaN scan beside the sweep (d2d_nanscan.hpp; null: the scan writes into grad / partial itself, behind the sweep): what the
    // scan found, per patch -- applied by nan_apply_kernel once both are through
    unsigned long long* nan_cell_bits;  // [patches] bit l: the cell of lane l gets a NaN gradient
    unsigned* nan_row_bits;             // [patches][nan_row_words] word 0: the patch has a flag (the fixed end point's entries), then one bit per object
    int nan_row_words;
    int nan_wqcap, nan_rb;  // region scan: queue entries / batches per round in use, in [1, NAN_WQCAP] / [1, NAN_RB] (the host fills them in)
#ifdef D2D_AB_TIMELINE
    unsigned long long* tl_ring;  // diagnostic build: [256][2] first start / end (100 MHz real-time counter) of the last 256 forward launches
    int tl_seq;
#endif
    unsigned long long* stats; // [D2D_NUM_STATS] executed-work counters (STATS build only), may be null
    unsigned long long* wave_cycles;  // [n_patches] shader clock ticks spent per patch (STATS build only), may be null
};

// Work counter of the patch a wave is sweeping, in units of ~25 wave-instructions (wave-uniform: one s_add).  It feeds
// the next launch's dearest-first schedule.  (A/B on one MI355X: ordering by elapsed ticks instead costs nothing to
// measure but schedules 5 % worse at 1024^2; this counter costs 4 % at 4096^2, where the schedule hardly matters.)
#define D2D_WORK(x) (st.work += (x))
#define D2D_EPS 1.1920929e-07f  // jnp.finfo(float32).eps, geometry.py:200

// Executed-work counters of one wave (wave-uniform, live in SGPRs). Only the STATS build of the
// kernel touches them; the timed kernel is compiled without.
//   [0] candidates evaluated (points + on_objects)      [1] candidates that reached the loss stage
//   [2] candidates that reached the occlusion loop      [3] candidates that reached valid*fun
//   [4] segment/wall tests evaluated (filter)           [5] tests that took the exact-divide path
//   [6] sum over [0] of the candidate order k           [7] sum over [1] of k   [8] sum over [3] of (k+1)
struct WaveStats {
    // (32-bit, and kept in vector registers by stat_add: as 16 wave-uniform 64-bit values they took 32 scalar registers of a
    // kernel that already parks scalars in VGPR lanes; one wave never counts past 2^32)
    unsigned c[16];  // [9] tile-culling levels evaluated; [10..15] shader-clock ticks per phase (diagnostic)
    int shadow;                // wave state, not a counter: the wall that occluded the wave's previous candidate
    unsigned work;             // every build: work done for this patch in units of ~25 wave-instructions (feeds the schedule)
};

__device__ __forceinline__ bool wave_any(bool p) { return __any(p); }

// The same wave-uniform index, unknown to the optimiser: a table row loaded through it is loaded HERE, not kept in scalar
// registers from an earlier load of the same row (the candidate's walls are read before the wall loop and again behind it,
// where few candidates arrive: held across the loop they are 4 scalars per wall that the loop's own state then has to do without)
#ifndef D2D_LATE_INDEX
#define D2D_LATE_INDEX 1
#endif
__device__ __forceinline__ int late_index(int i) {
#if D2D_LATE_INDEX
    asm volatile("" : "+s"(i));
#endif
    return i;
}

// counter I += x in a vector register (see WaveStats)
#ifndef D2D_WALL_PAIRS_MODES
#define D2D_WALL_PAIRS_MODES 3  // bit m: validity mode m takes two walls per trip (A/B)
#endif
#ifndef D2D_WALL_PAIRS
#define D2D_WALL_PAIRS 1  // A/B: 0 = the wall loop of eval_candidate takes one wall per trip (rounds 1 - 3)
#endif
#ifndef D2D_STAT_MASK  // (A/B: the counters an instrumented build keeps)
#define D2D_STAT_MASK 0xffff
#endif
template <int I>
__device__ __forceinline__ void stat_add(WaveStats& st, unsigned long long x) {
    const unsigned x32 = (unsigned)x;
    if ((D2D_STAT_MASK >> I) & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(st.c[I]) : "v"(x32));
}

// ---- correctly rounded fp32 division without the range scaling of the generic expansion --------------
// hipcc expands x / y (with -fhip-fp32-correctly-rounded-divide-sqrt) into v_div_scale x2, v_rcp, a Newton
// step on the reciprocal, two residual corrections of the quotient (the last one in v_div_fmas) and
// v_div_fixup.  When neither operand needs scaling (|x|, |y| in [2^-62, 2^62] or x == 0) the scale factors
// are 1 and the fixup is the identity, so the bare fma chain below returns the same correctly rounded
// quotient; it costs 8 VALU ops instead of 11 and lets several numerators share one refined reciprocal.
// d2d_selftest_div checks bit-equality with x / y on the GPU (tests/test_gpu_selftest.py).
// Measured again in round 2 (scripts/ab_build.sh "nofastdiv:-DD2D_FAST_DIV=0"), now that the kernel is bound by its
// compare -> lane-mask -> branch round trips rather than by vector issue: the wave-uniform range check and branch in front
// of the bare chain cost more than the three instructions it saves -- the generic expansion is faster (hard_sigmoid -5 %,
// cfg4 -15 %, cfg2 hard +-0).  Off by default; the bare chain and its self-test stay for A/B.
#ifndef D2D_FAST_DIV
#define D2D_FAST_DIV 0
#endif
__device__ __forceinline__ bool div_in_range(float x) {
    float ax = fabsf(x);
    return ax >= 2.168404345e-19f && ax <= 4.611686018e18f;  // [2^-62, 2^62]
}
__device__ __forceinline__ float rcp_refined(float y) {
    float r = __builtin_amdgcn_rcpf(y);
    float e = __builtin_fmaf(-y, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_with_rcp(float x, float y, float r) {
    float q = x * r;
    float e = __builtin_fmaf(-y, q, x);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-y, q, x);
    return __builtin_fmaf(e, r, q);
}
// x1 / y and x2 / y, correctly rounded; wave-uniform choice between the bare chain and the generic expansion
__device__ __forceinline__ void div2_exact(float x1, float x2, float y, float& q1, float& q2) {
#if D2D_FAST_DIV
    const bool ok = div_in_range(y) && (x1 == 0.0f || div_in_range(x1)) && (x2 == 0.0f || div_in_range(x2));
    if (!wave_any(!ok)) {
        const float r = rcp_refined(y);
        q1 = div_with_rcp(x1, y, r);
        q2 = div_with_rcp(x2, y, r);
        return;
    }
#endif
    q1 = x1 / y;
    q2 = x2 / y;
}
__device__ __forceinline__ float div1_exact(float x, float y) {
#if D2D_FAST_DIV
    const bool ok = div_in_range(y) && (x == 0.0f || div_in_range(x));
    if (!wave_any(!ok)) return div_with_rcp(x, y, rcp_refined(y));
#endif
    return x / y;
}

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

// f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>): a loop whose index is a compile-time constant inside the body
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Pre-division hard_sigmoid: clamp(alpha*x + 3, 0, 6); hard_sigmoid(x) = clampact(x) / 6.
__device__ __forceinline__ float clampact(float x, float alpha) {
    float z = alpha * x;
    return fminf(fmaxf(z + 3.0f, 0.0f), 6.0f);
}

// expf as the C library of the reference's host evaluates it -- glibc >= 2.27 (sysdeps/ieee754/flt-32/e_expf.c, from ARM's
// optimized-routines): x N / ln 2 = k + r in double, exp(x) = 2^(k/N) * (C0 r^3 + C1 r^2 + C2 r + 1) with N = 32 and a table
// of 2^(i/N), rounded to float once at the end.  jax.nn.sigmoid on XLA-CPU is its own polynomial and cannot be reproduced
// here; what CAN be made identical is this repository's oracle (oracle/d2d_oracle.c calls libm's expf) and the device: the
// device library's expf differs from libm's by an ulp in a few percent of the arguments, and the sigmoid mode's validity is a
// product of such terms.  The same double operations in the same order, no contraction: checked bit for bit against libm over
// 800 000 arguments on the host (scripts/check_expf_model.py) and on the device (tests/test_gpu_selftest.py).
static __device__ __constant__ unsigned long long EXPF_TAB[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
__device__ __forceinline__ float expf_libm(float x) {
    if (x != x) return x + x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();   // log(0x1p128)
    if (x < -0x1.9fe368p6f) return 0.0f;              // log(0x1p-150)
    if (x < -0x1.9d1d9ep6f) return 0x1p-149f;         // log(0x1p-149): __math_may_uflowf
    const double z = (0x1.71547652b82fep+0 * 32.0) * (double)x;
    double kd = z + 0x1.8p+52;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd = kd - 0x1.8p+52;
    const double r = z - kd;
    const unsigned long long t = EXPF_TAB[ki & 31ull] + (ki << 47);
    const double sc = __longlong_as_double((long long)t);
    const double zz = (0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0) * r + (0x1.ebfce50fac4f3p-3 / 32.0 / 32.0);
    const double r2 = r * r;
    double y = (0x1.62e42ff0c52d6p-1 / 32.0) * r + 1.0;
    y = zz * r2 + y;
    y = y * sc;
    return (float)y;
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf_libm(-z)); }

// Sigmoid validity (MODE_SIG): the pre-activation z below which a candidate's contribution valid * fun <= exp(z) * f_max
// is certainly less than a quarter ulp of `acc` -- adding it would leave acc unchanged under round to nearest, whatever
// its sign.  l2f = log2 of an upper bound of |fun| (SweepArgs::sig_l2f; huge: no bound, nothing may be skipped); 0.01 in
// log2 units covers the rounding of expf, of the products and of this estimate itself.  acc == 0 or denormal: < -89.
__device__ __forceinline__ float sig_zc_of(float l2f, float acc) {
    const int E = (int)((__float_as_uint(acc) >> 23) & 0xffu) - 127;
    return ((float)(E - 25) - l2f - 0.01f) * 0.69314718f;
}

// Per-lane gradient state of the value+grad kernel (GRAD build of eval_candidate).
struct GradCtx {
    float grx, gry;  // d acc / d rx of this cell (seed 1), scene.py:1920-1923 (argnums=1)
    float tbx, tby;  // cot * d acc / d tx, summed over candidates (scene-parameter VJP)
    float cot;       // cotangent of this cell's accumulated value
    float* wl;       // LDS [4 N] of this wave: sum over lanes/candidates of cot * (d/d origin.xy, d/d dest.xy)
    bool scene;      // accumulate tbx/tby/wl ?
    int ci;          // D2D_FUN_CUSTOM: ordinal of the next candidate in the sweep's order (wave-uniform)
    long cell;       // D2D_FUN_CUSTOM: this lane's cell
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// d activation(x) / d x as JAX differentiates it: hard_sigmoid = relu6(alpha x + 3) / 6 -> alpha/6 strictly inside
// the ramp; sigmoid = lax.logistic(alpha x) -> alpha * s * (1 - s).
template <int MODE>
__device__ __forceinline__ float dact(float x, float alpha) {
    float z = alpha * x;
    if (MODE == MODE_HSIG) {
        float y = z + 3.0f;
        return (y > 0.0f && y < 6.0f) ? alpha / 6.0f : 0.0f;
    }
    float s = sigmoidf_(z);
    return alpha * (s * (1.0f - s));
}

// adjoint of (ox, oy) = normalize2(vx, vy): vbar = (obar - (obar . o) o) / len   (len guarded like the forward)
__device__ __forceinline__ void normalize2_bwd(float vx, float vy, float obx, float oby, float& vbx, float& vby) {
    float len = sqrtf(vx * vx + vy * vy);
    bool z = (len == 0.0f);
    len = z ? 1.0f : len;
    float ox = vx / len, oy = vy / len;
    float d = z ? 0.0f : (obx * ox + oby * oy);
    vbx = (obx - d * ox) / len;
    vby = (oby - d * oy) / len;
}

// (txx, txy) / (rxx, rxy): the path's end points; exactly one of the two is the lane's grid cell, the other is
// wave-uniform (RX grid: scene.py:1803-1953; TX grid, TXG = true: scene.py:1489-1648, where the per-cell gradient is
// taken w.r.t. the transmitter, scene.py:1617-1620).
template <int K, int MODE, bool STATS, bool GRAD = false, bool PREF = true, bool TXG = false>
__device__ __forceinline__ void eval_candidate(const SweepArgs& a, const int (&cand)[D2D_MAX_ORDER],
                                               const float (&imgx)[D2D_MAX_ORDER], const float (&imgy)[D2D_MAX_ORDER],
                                               float txx, float txy, float rxx, float rxy, bool lane_bad, float& acc,
                                               WaveStats& st, GradCtx* g = nullptr, float acc_floor = -1.0f) {
    // acc_floor >= 0 (MODE_SIG, fun >= 0): the caller adds this candidate to a sum it does not hold -- `acc` is a scratch
    // that receives the contribution alone -- but knows that sum to be at least acc_floor when the addition happens
    const float zc_acc = (acc_floor >= 0.0f) ? acc_floor : acc;
    int on_i = 0, on_w = 0, hit_i = 0, hit_j = -1;  // GRAD: which activation carries the min / max
    float hit2 = -3.0e38f;  // GRAD, approx modes: the largest occlusion test OTHER than the one that carries the max (ties: JAX splits)
    bool znan = false;  // GRAD: the reference's autodiff yields NaN for this (cell, candidate), see below
    long cust = 0;      // GRAD, D2D_FUN_CUSTOM: this (candidate, cell)'s row of the host-evaluated path function
    if (GRAD && a.fun_id == D2D_FUN_CUSTOM) cust = (long)(g->ci++) * a.cust_cells + g->cell;
    float px[K + 2], py[K + 2];
    px[0] = txx;
    py[0] = txy;
    px[K + 1] = rxx;
    py[K + 1] = rxy;

    // ---- pre-filter on the LAST interaction point (the first one the backward scan produces) -----------
    // With an approximate quotient (v_rcp, a few ulp) and an explicit error bound M, decide whether the point's
    // parametric coordinate s on its wall is certainly outside the range where on_objects is not exactly 0.
    // If that holds in every lane, valid == 0 for the whole wave whatever the other points are: skip the
    // candidate before any exact division.  (Not in the GRAD build: the reference's autodiff NaN traps depend
    // on all interaction points.)
    if (PREF && !GRAD && K > 0) {
        const float4 r0 = ldc4(a.refl, 2 * cand[K - 1]);
        const float4 r1 = ldc4(a.refl, 2 * cand[K - 1] + 1);
        const float4 fc = ldc4(a.flt, cand[K - 1]);
        float ux = rxx - imgx[K - 1], uy = rxy - imgy[K - 1];
        float vx = r0.x - rxx, vy = r0.y - rxy;
        float un = ux * r0.z + uy * r0.w;
        float vn = vx * r0.z + vy * r0.w;
        float gq = vn * __builtin_amdgcn_rcpf(un);
        float dx = __builtin_fmaf(gq, ux, -vx), dy = __builtin_fmaf(gq, uy, -vy);  // p - origin
        float sa = __builtin_fmaf(r1.y, dy, r1.x * dx) * fc.x;
        // magnitudes that enter p - origin on the exact path (incl. the rounding of p = pt + inc at |origin| scale)
        float ex = __builtin_fmaf(fabsf(gq), fabsf(ux), fabsf(vx)) + fabsf(r0.x);
        float ey = __builtin_fmaf(fabsf(gq), fabsf(uy), fabsf(vy)) + fabsf(r0.y);
        float M = __builtin_fmaf(__builtin_fmaf(fabsf(r1.y), ey, fabsf(r1.x) * ex), fc.y, 1e-30f);
        bool cull = !lane_bad && (un != 0.0f) && (ex < 1e18f) && (ey < 1e18f) && ((sa + M < a.on_lo) || (sa - M > a.on_hi));
        if (STATS) stat_add<9>(st, 1);
        if (!wave_any(!cull)) return;
    }

    D2D_WORK(6 + 2 * K);  // backward scan + on_objects
    if (STATS) {
        stat_add<0>(st, 1);
        stat_add<6>(st, K);
    }
    float sv[K + 1];  // parametric coordinates of the interaction points on their walls (fp32, as on_objects computes them)
    // What the later stages will want from memory is asked for now (scalar loads, wave-uniform): the masks of the candidate's
    // first wall and wall pairs and the first wall of the occlusion loop.  A candidate is one dependent chain; a round trip
    // that is in flight during the scan is a round trip the wave does not sit out later.
    constexpr bool PRE = !GRAD && !TXG && K >= 1;
    unsigned long long sh_pre = 0ull, pm_pre[K + 1];
    constexpr bool HOIST = PRE && K <= 2;  // (orders >= 3: the extra live scalars cost more than the waits: cfg4 +5 %)
    if (HOIST) {
        if (a.shadow != nullptr) sh_pre = cmem(a.shadow)[cand[0]];
#pragma unroll
        for (int i = 0; i + 1 < K; ++i) pm_pre[i] = (a.pair != nullptr) ? cmem(a.pair)[(size_t)cand[i] * a.N + cand[i + 1]] : 0ull;
    }
    const int sh = (st.shadow >= 0 && st.shadow < a.N) ? st.shadow : -1;
    int j = sh >= 0 ? sh : 0;
    float4 w = ldc4(a.occl, a.N > 0 ? j : 0);
    // ---- backward scan of the image method, geometry.py:1093-1110 -------------------------
    {
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
            // jnp.where(un == 0, 0, vn*u/un): reverse mode sends a zero cotangent through the untaken
            // division by zero -> 0/0 = NaN (geometry.py:1105)
            // (hard validity is a bool: the contribution is differentiated through `fun` alone, and fun = 1 ignores the path --
            // nothing then reaches the division, the gradient is simply 0)
            if (GRAD && (MODE != MODE_HARD || a.fun_id != D2D_FUN_ONE)) znan = znan || z;
            if (!GRAD) {
                // This wall's parametric coordinate (on_objects, geometry.py:589-621) right away: a candidate whose point
                // is exactly off this wall in every lane is invalid whatever its other walls say, and most candidates
                // that reach this function die so -- the remaining steps of the scan are then never computed.  (The
                // value+grad build must see every interaction point: its NaN artefacts depend on all of them.)
                const float4 r1 = ldc4(a.refl, 2 * cand[i] + 1);
                const float dx = ptx - r0.x, dy = pty - r0.y;
                const float sw = div1_exact(r1.x * dx + r1.y * dy, r1.z);
                sv[i] = sw;
                if (i > 0) {
                    bool off;
                    if (MODE == MODE_HARD) off = !((sw >= 0.0f) && (sw <= 1.0f));
                    else if (MODE == MODE_HSIG) off = fminf(clampact(sw - 0.0f, a.alpha), clampact(1.0f - sw, a.alpha)) == 0.0f;
                    else off = fminf(a.alpha * (sw - 0.0f), a.alpha * (1.0f - sw)) <= fmaxf(-89.0f, sig_zc_of(a.sig_l2f, zc_acc));
                    const bool pt_bad = lane_bad || !(fabsf(ptx) < 1e18f) || !(fabsf(pty) < 1e18f);
                    if (!wave_any(!off || pt_bad)) {
                        if (STATS && i == K - 1) stat_add<15>(st, 1);  // died at the last wall (the first the scan reaches)
                        return;
                    }
                }
            }
        }
    }
    if (GRAD && MODE != MODE_HARD && K > 0) {
        // normalize() of a zero-length segment inside the differentiated loss: sqrt'(0) * 0 = NaN
        // (geometry.py:227-228, 647-648); in hard mode the loss only feeds a boolean and is not differentiated
#pragma unroll
        for (int i = 0; i <= K; ++i) znan = znan || (px[i + 1] == px[i] && py[i + 1] == py[i]);
    }
    if (GRAD && a.fun_id != D2D_FUN_ONE && a.fun_id != D2D_FUN_CUSTOM) {
        // path_length's guard fails where it is needed: a segment vector of exactly (-eps, -eps) becomes (0, 0) once eps is
        // added to both components (geometry.py:199-200), and jnp.linalg.norm's derivative at 0 is 0 / 0 -- NaN whatever
        // cotangent reaches it, a valid candidate's or an invalid one's zero (every path function but the constant one
        // goes through the length)
#pragma unroll
        for (int i = 0; i <= K; ++i) znan = znan || (((px[i + 1] - px[i]) + D2D_EPS == 0.0f) && ((py[i + 1] - py[i]) + D2D_EPS == 0.0f));
    }
    if (GRAD && wave_any(znan)) {
        const float qnan = __builtin_nanf("");
        if (znan) {
            g->grx = qnan;
            g->gry = qnan;
        }
        if (g->scene) {
            if (znan) g->tbx = g->tby = qnan;
            if ((threadIdx.x & 63) == 0) {
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    float* w4 = g->wl + 4 * cand[i];
                    w4[0] = w4[1] = w4[2] = w4[3] = qnan;
                }
            }
        }
    }

    // Lanes whose coordinates are not comfortably finite never take part in a skip decision
    // (0 * fun must then be evaluated for real: it may be NaN).
    bool bad = lane_bad;
#pragma unroll
    for (int i = 1; i <= K; ++i) bad = bad || !(fabsf(px[i]) < 1e18f) || !(fabsf(py[i]) < 1e18f);
    if (GRAD && a.fun_id == D2D_FUN_CUSTOM) {
        // a host-evaluated path function that is inf / NaN for this (candidate, cell) -- or whose derivative is -- must meet the
        // candidate's valid = 0 for real (0 * NaN = NaN in the reference's sum, scene.py:1909, and in its gradient)
        const float* pb = a.cust_pb + cust * (2 * (D2D_MAX_ORDER + 2));
        bad = bad || !(fabsf(a.cust_f[cust]) < 3.0e38f);
#pragma unroll
        for (int i = 0; i < 2 * (K + 2); ++i) bad = bad || !(fabsf(pb[i]) < 3.0e38f);
    }

    // ---- on_objects, geometry.py:821-854 / 589-621 ----------------------------------------
    bool on_b = true;     // MODE_HARD
    float on_c = 6.0f;    // MODE_HSIG: min of clamped pre-activations (true_value = 6/6)
    float on_z = 3.0e38f; // MODE_SIG: min of alpha*x (true_value = 1.0 handled at the end)
    bool nanflag = false;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        float s;
        if (GRAD) {
            const float4 r0 = ldc4(a.refl, 2 * cand[i]);
            const float4 r1 = ldc4(a.refl, 2 * cand[i] + 1);
            float dx = px[i + 1] - r0.x, dy = py[i + 1] - r0.y;
            s = div1_exact(r1.x * dx + r1.y * dy, r1.z);
            sv[i] = s;
        } else {
            s = sv[i];  // (computed in the backward scan)
        }
        if (MODE == MODE_HARD) {
            on_b = on_b && (s >= 0.0f) && (s <= 1.0f);
        } else if (MODE == MODE_HSIG) {
            nanflag = nanflag || (s != s);
            float c1 = clampact(s - 0.0f, a.alpha), c2 = clampact(1.0f - s, a.alpha);
            float cm = fminf(c1, c2);
            if (GRAD && cm < on_c) {
                on_i = i;
                on_w = (c2 < c1) ? 1 : 0;
            }
            on_c = fminf(on_c, cm);
        } else {
            nanflag = nanflag || (s != s);
            float z1 = a.alpha * (s - 0.0f), z2 = a.alpha * (1.0f - s);
            float zm = fminf(z1, z2);
            if (GRAD && zm < on_z) {
                on_i = i;
                on_w = (z2 < z1) ? 1 : 0;
            }
            on_z = fminf(on_z, zm);
        }
    }
    // sigmoid(z) is exactly 0 only once exp(-z) overflows: z <= -89
    // ... or, for this lane's running sum, once the contribution is certainly below a quarter ulp of it (sig_zc_of; not in
    // the value+grad build: the adjoints are not absorbed by the sum)
    bool on_zero = (MODE == MODE_HARD) ? !on_b : (MODE == MODE_HSIG) ? (on_c == 0.0f)
                   : (on_z <= (GRAD ? -89.0f : fmaxf(-89.0f, sig_zc_of(a.sig_l2f, zc_acc))));
    if (K > 0 && !wave_any(!on_zero || bad)) return;  // valid == 0 in every lane: acc + 0.0

    // Lanes for which the occlusion result can still change the output.  (is_valid = all(on_objects,
    // not intersects, loss < tol): the three terms commute, so the cheap-to-refute occlusion comes before the loss.)
    const bool live = !on_zero || bad;
    D2D_WORK(1);

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
    // ---- per-lane look-up in the occlusion masks --------------------------------------------------------------------
    // The first-segment shadow masks and the wall-to-wall masks say, bin by bin, where a segment is CERTAINLY reported
    // as intersecting some object by the exact path (hard: hit; approx: exactly saturated) -- for every point within
    // dperp of the bin.  The tile culling consults them with the ranges a whole patch can reach; here each lane consults
    // them with the point it has actually computed (its fp32 parametric coordinate +- the rounding of that coordinate's
    // own evaluation, both neighbours when the pad straddles a bin boundary).  A lane they settle is occluded exactly as
    // if the wall loop had found its occluder; when they settle every lane, the wall loop is not entered at all.
    if (!GRAD && !TXG && K >= 1 && a.shadow != nullptr) {  // (TX grids: the masks belong to the other end of the path)
        const float eps = 1.1920929e-07f;
        bool occl = false;
        float Ei[K + 1];   // magnitudes that entered point i (how far its fp32 value may sit off its wall's line: 128 eps Ei)
        float padv[K + 1];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float4 r0 = ldc4(a.refl, 2 * cand[i]);
            const float4 r1 = ldc4(a.refl, 2 * cand[i] + 1);
            const float4 fc = ldc4(a.flt, cand[i]);
            const float ex = fabsf(px[i + 1] - px[i + 2]) + fabsf(r0.x - px[i + 2]) + fabsf(r0.x) + fabsf(imgx[i]);
            const float ey = fabsf(py[i + 1] - py[i + 2]) + fabsf(r0.y - py[i + 2]) + fabsf(r0.y) + fabsf(imgy[i]);
            Ei[i] = ex + ey;
            padv[i] = __builtin_fmaf(__builtin_fmaf(fabsf(r1.y), ey, fabsf(r1.x) * ex), fc.y, 1e-4f);
        }
        {
            const unsigned long long shm = HOIST ? sh_pre : cmem(a.shadow)[cand[0]];  // wave-uniform (asked for at the top)
            if (shm != 0ull) {
                const float fa_ = (sv[0] - padv[0] - a.shadow_lo) * a.shadow_inv, fb_ = (sv[0] + padv[0] - a.shadow_lo) * a.shadow_inv;
                const bool ok = fa_ >= 0.0f && fb_ < 64.0f && 256.0f * eps * Ei[0] <= a.shadow_dperp;
                const int ka = ok ? (int)fa_ : 0, kb = ok ? (int)fb_ : 0;
                occl = ok && (kb - ka <= 1) && ((shm >> ka) & 1ull) && ((shm >> kb) & 1ull);
            }
        }
        if (K >= 2 && a.pair != nullptr) {
#pragma unroll
            for (int i = 0; i + 1 < K; ++i) {
                // in path order the earlier point is P3 (rows of 8 bits), the later P4 (geometry.py:881-904)
                const unsigned long long m = HOIST ? pm_pre[i] : cmem(a.pair)[(size_t)cand[i] * a.N + cand[i + 1]];  // wave-uniform
                if (m != 0ull) {
                    const float k8 = 0.125f * a.shadow_inv;
                    const float ea_ = (sv[i] - padv[i] - a.shadow_lo) * k8, eb_ = (sv[i] + padv[i] - a.shadow_lo) * k8;
                    const float la_ = (sv[i + 1] - padv[i + 1] - a.shadow_lo) * k8, lb_ = (sv[i + 1] + padv[i + 1] - a.shadow_lo) * k8;
                    const bool ok = ea_ >= 0.0f && eb_ < 8.0f && la_ >= 0.0f && lb_ < 8.0f && 512.0f * eps * Ei[i] <= a.pair_dperp &&
                                    512.0f * eps * Ei[i + 1] <= a.pair_dperp;
                    const int ea = ok ? (int)ea_ : 0, eb = ok ? (int)eb_ : 0, la = ok ? (int)la_ : 0, lb = ok ? (int)lb_ : 0;
                    const bool all4 = ((m >> (ea + 8 * la)) & 1ull) && ((m >> (eb + 8 * la)) & 1ull) && ((m >> (ea + 8 * lb)) & 1ull) &&
                                      ((m >> (eb + 8 * lb)) & 1ull);
                    occl = occl || (ok && all4 && (eb - ea <= 1) && (lb - la <= 1));
                }
            }
        }
        if (wave_any(occl)) {
            if (MODE == MODE_HARD) hit_b = hit_b || occl;
            else if (MODE == MODE_HSIG) hit_c = occl ? 6.0f : hit_c;
            else {
                hit_z = occl ? 1.0e30f : hit_z;
                any_test = true;
            }
            active = active && (!occl || bad);
            D2D_WORK(2);
            if (!wave_any(active)) return;  // every lane occluded (or off its walls): valid == 0 whatever the loss is
        }
    }
    if (STATS) stat_add<2>(st, 1);  // candidates that enter the wall loop
    // The wall that finished off the previous candidate of this wave is tried first ("shadow cache"): or / max
    // do not depend on the order of the tests, and neighbouring candidates tend to share their occluder.
    // (the next wall's data are fetched while this one is tested: a lone wave would otherwise sit out one scalar-load
    // latency per wall)
    // (D2D_WALL_PAIRS_MODES: hard and hard_sigmoid.  hard_sigmoid LOST 3 % with two walls per trip -- 147 scalars parked in VGPR
    // lanes instead of 62 -- until the loss stage reloaded its walls' rows (late_index): 0.132 -> 0.125 ms at cfg2 with both;
    // sigmoid gains nothing, 5.0 ms either way.  How far ahead the pair loop loads its walls -- both next walls before the
    // filters, one, none -- makes no difference: 0.077 - 0.078)
    constexpr bool PAIRS = D2D_WALL_PAIRS && ((D2D_WALL_PAIRS_MODES >> MODE) & 1);
    if constexpr (PAIRS) {
    // Two walls per trip: their filters are independent instruction streams (a lone wave issues a dependent chain at a
    // fraction of its rate), and the wave-level question "does any lane need an exact test" is asked once for both.  The
    // exact tests recompute the three bilinear forms (the same expressions on the same operands: the same bits) instead of
    // keeping them in registers for the one wall in ten that needs them.
    auto filt = [&](const float4& ww, int jj) -> unsigned {
        unsigned wbits = 0u;  // bit i: this lane needs the exact test of segment i
#pragma unroll
        for (int i = 0; i <= K; ++i) {
            const int ig0 = (i == 0) ? -1 : cand[i - 1];
            const int ig1 = (i == K) ? -1 : cand[i];
            const bool skip = (jj == ig0 || jj == ig1);  // wave-uniform: a segment ignores the walls it joins
            float Cx = ww.x - px[i], Cy = ww.y - py[i];
            float fa = by[i] * Cx - bx[i] * Cy;   // geometry.py:157
            float fb = ww.z * Cy - ww.w * Cx;     // geometry.py:158
            float fd = ww.w * bx[i] - ww.z * by[i]; // geometry.py:159
            // divide-free filter: t = fl(num/fd) certainly outside [flt_lo, flt_hi]?  num/fd lies outside iff
            // (num - lo fd)(num - hi fd) > 0, whatever the sign of fd; each factor is one fma, so its sign is the exact
            // difference's, and a product that underflows to 0 (tiny fd, or a numerator on a window edge) counts as "not
            // certainly outside": the exact test decides.  One compare per segment instead of six and their mask algebra.
            float pa = __builtin_fmaf(-a.flt_lo, fd, fa) * __builtin_fmaf(-a.flt_hi, fd, fa);
            float pb = __builtin_fmaf(-a.flt_lo, fd, fb) * __builtin_fmaf(-a.flt_hi, fd, fb);
            bool miss = fmaxf(pa, pb) > 0.0f;
            if (MODE == MODE_SIG && !skip) any_test = true;
            if (STATS) stat_add<4>(st, skip ? 0 : 1);  // (branch-free on purpose: see the single-wall loop below)
            wbits |= (!skip && active && (!miss || bad)) ? (1u << i) : 0u;
        }
        return wbits;
    };
    auto exact = [&](const float4& ww, int jj, unsigned wbits) {
#pragma unroll
        for (int i = 0; i <= K; ++i) {
            if (!wave_any((wbits >> i) & 1u)) continue;
            if (STATS) stat_add<5>(st, 1);
            D2D_WORK(2);
            const float Cx = ww.x - px[i], Cy = ww.y - py[i];
            const float fa = by[i] * Cx - bx[i] * Cy;
            const float fb = ww.z * Cy - ww.w * Cx;
            const float fd = ww.w * bx[i] - ww.z * by[i];
            // exact path, geometry.py:163-171
            bool dz = (fd == 0.0f);
            float dd = dz ? 1.0f : fd;
            float ta, tb;
            div2_exact(fa, fb, dd, ta, tb);
            ta = dz ? __builtin_inff() : ta;
            tb = dz ? __builtin_inff() : tb;
            if (MODE == MODE_HARD) {
                bool h = (ta >= a.seg_lo) && (ta <= a.seg_hi) && (tb >= a.seg_lo) && (tb <= a.seg_hi);
                hit_b = hit_b || h;
            } else if (MODE == MODE_HSIG) {
                nanflag = nanflag || (ta != ta) || (tb != tb);
                float c = fminf(fminf(clampact(ta - a.seg_lo, a.alpha), clampact(a.seg_hi - ta, a.alpha)),
                                fminf(clampact(tb - a.seg_lo, a.alpha), clampact(a.seg_hi - tb, a.alpha)));
                // arg-max as the reference's ascending (j, i) scan finds it: the first of equal maxima
                if (GRAD && (c > hit_c || (c == hit_c && hit_j >= 0 && (jj < hit_j || (jj == hit_j && i < hit_i))))) {
                    hit_i = i;
                    hit_j = jj;
                }
                if (GRAD) hit2 = (c > hit_c) ? hit_c : fmaxf(hit2, c);
                hit_c = fmaxf(hit_c, c);
            } else {
                nanflag = nanflag || (ta != ta) || (tb != tb);
                float z = fminf(fminf(a.alpha * (ta - a.seg_lo), a.alpha * (a.seg_hi - ta)),
                                fminf(a.alpha * (tb - a.seg_lo), a.alpha * (a.seg_hi - tb)));
                if (GRAD && (z > hit_z || (z == hit_z && hit_j >= 0 && (jj < hit_j || (jj == hit_j && i < hit_i))))) {
                    hit_i = i;
                    hit_j = jj;
                }
                if (GRAD) hit2 = (z > hit_z) ? hit_z : fmaxf(hit2, z);
                hit_z = fmaxf(hit_z, z);
            }
        }
        // decided lanes: occlusion already makes valid exactly 0
        if (MODE == MODE_HARD) active = active && (!hit_b || bad);
        else if (MODE == MODE_HSIG) active = active && (hit_c != 6.0f || bad);
        else active = active && (hit_z < 17.5f || bad);
    };
    {
        bool done = a.N <= 0;
        if (!done && sh >= 0) {  // the cached occluder, on its own: it usually ends the loop
            const unsigned b0 = filt(w, sh);
            D2D_WORK(K + 1);
            if (wave_any(b0 != 0u)) {
                exact(w, sh, b0);
                if (!wave_any(active)) {
                    st.shadow = sh;
                    done = true;
                }
            }
        }
        auto next_idx = [&](int x) -> int {
            ++x;
            return x == sh ? x + 1 : x;
        };
        int jA = (sh == 0) ? 1 : 0;
        if (!done && jA < a.N) {
            int jB = next_idx(jA);
            float4 wA = (sh >= 0) ? ldc4(a.occl, jA) : w;  // (w holds wall 0 when there is no cached occluder)
            float4 wB = ldc4(a.occl, jB < a.N ? jB : jA);
            while (true) {
                const bool hasB = jB < a.N;
                const int nA = next_idx(jB), nB = next_idx(nA);
                // (the next trip's walls: their loads fly while this trip computes)
                const float4 wnA = ldc4(a.occl, nA < a.N ? nA : jA), wnB = ldc4(a.occl, nB < a.N ? nB : jA);
                const unsigned ba = filt(wA, jA);
                const unsigned bb = hasB ? filt(wB, jB) : 0u;
                D2D_WORK(hasB ? 2 * (K + 1) : (K + 1));
                if (wave_any((ba | bb) != 0u)) {
                    if (wave_any(ba != 0u)) {
                        exact(wA, jA, ba);
                        if (!wave_any(active)) {
                            st.shadow = jA;
                            break;
                        }
                    }
                    if (wave_any(bb != 0u)) {
                        exact(wB, jB, bb);
                        if (!wave_any(active)) {
                            st.shadow = jB;
                            break;
                        }
                    }
                }
                if (!(nA < a.N)) break;
                jA = nA;
                jB = nB;
                wA = wnA;
                wB = wnB;
            }
        }
    }
    } else {
    int nxt = sh >= 0 ? 0 : 1;
    if (nxt == sh) ++nxt;
    for (bool more = a.N > 0; more;) {
        const bool has_next = nxt < a.N;
        const float4 wn = ldc4(a.occl, has_next ? nxt : j);
        // The K + 1 segments against this wall: first the divide-free filter of all of them (pure vector work, no branch:
        // a segment that ends on this very wall is computed and masked, that is rarer than a branch is dear), then ONE
        // wave-level decision whether any lane needs an exact test at all (9 % of the tests at cfg2), and only then the
        // segments that do, one by one.
        float fa_[K + 1], fb_[K + 1], fd_[K + 1];
        unsigned wbits = 0u;  // bit i: this lane needs the exact test of segment i
#pragma unroll
        for (int i = 0; i <= K; ++i) {
            const int ig0 = (i == 0) ? -1 : cand[i - 1];
            const int ig1 = (i == K) ? -1 : cand[i];
            const bool skip = (j == ig0 || j == ig1);  // wave-uniform: a segment ignores the walls it joins
            float Cx = w.x - px[i], Cy = w.y - py[i];
            float fa = by[i] * Cx - bx[i] * Cy;   // geometry.py:157
            float fb = w.z * Cy - w.w * Cx;       // geometry.py:158
            float fd = w.w * bx[i] - w.z * by[i]; // geometry.py:159
            // divide-free filter: t = fl(num/fd) certainly outside [flt_lo, flt_hi]?  num/fd lies outside iff
            // (num - lo fd)(num - hi fd) > 0, whatever the sign of fd; each factor is one fma, so its sign is the exact
            // difference's, and a product that underflows to 0 (tiny fd, or a numerator on a window edge) counts as "not
            // certainly outside": the exact test decides.  One compare per segment instead of six and their mask algebra.
            float pa = __builtin_fmaf(-a.flt_lo, fd, fa) * __builtin_fmaf(-a.flt_hi, fd, fa);
            float pb = __builtin_fmaf(-a.flt_lo, fd, fb) * __builtin_fmaf(-a.flt_hi, fd, fb);
            bool miss = fmaxf(pa, pb) > 0.0f;
            if (MODE == MODE_SIG && !skip) any_test = true;
            // (branch-free on purpose: as `if (STATS && !skip) ++count` -- a branch with a side effect between the filter's compare
            // and its use -- the instrumented kernel's maps differed from the product kernel's at shadow boundaries, 0.5 % of the
            // cells at cfg2, from round 1 on; found in round 4, scripts/stats_cmp.py and tests/test_gpu_forward.py hold them equal)
            if (STATS) stat_add<4>(st, skip ? 0 : 1);
            wbits |= (!skip && active && (!miss || bad)) ? (1u << i) : 0u;
            fa_[i] = fa;
            fb_[i] = fb;
            fd_[i] = fd;
        }
        D2D_WORK(K + 1);
        if (wave_any(wbits != 0u)) {
#pragma unroll
            for (int i = 0; i <= K; ++i) {
                if (!wave_any((wbits >> i) & 1u)) continue;
                if (STATS) stat_add<5>(st, 1);
                D2D_WORK(2);
                const float fa = fa_[i], fb = fb_[i], fd = fd_[i];
                // exact path, geometry.py:163-171
                bool dz = (fd == 0.0f);
                float dd = dz ? 1.0f : fd;
                float ta, tb;
                div2_exact(fa, fb, dd, ta, tb);
                ta = dz ? __builtin_inff() : ta;
                tb = dz ? __builtin_inff() : tb;
                if (MODE == MODE_HARD) {
                    bool h = (ta >= a.seg_lo) && (ta <= a.seg_hi) && (tb >= a.seg_lo) && (tb <= a.seg_hi);
                    hit_b = hit_b || h;
                } else if (MODE == MODE_HSIG) {
                    nanflag = nanflag || (ta != ta) || (tb != tb);
                    float c = fminf(fminf(clampact(ta - a.seg_lo, a.alpha), clampact(a.seg_hi - ta, a.alpha)),
                                    fminf(clampact(tb - a.seg_lo, a.alpha), clampact(a.seg_hi - tb, a.alpha)));
                    // arg-max as the reference's ascending (j, i) scan finds it: the first of equal maxima
                    if (GRAD && (c > hit_c || (c == hit_c && hit_j >= 0 && (j < hit_j || (j == hit_j && i < hit_i))))) {
                        hit_i = i;
                        hit_j = j;
                    }
                    if (GRAD) hit2 = (c > hit_c) ? hit_c : fmaxf(hit2, c);
                    hit_c = fmaxf(hit_c, c);
                } else {
                    nanflag = nanflag || (ta != ta) || (tb != tb);
                    float z = fminf(fminf(a.alpha * (ta - a.seg_lo), a.alpha * (a.seg_hi - ta)),
                                    fminf(a.alpha * (tb - a.seg_lo), a.alpha * (a.seg_hi - tb)));
                    if (GRAD && (z > hit_z || (z == hit_z && hit_j >= 0 && (j < hit_j || (j == hit_j && i < hit_i))))) {
                        hit_i = i;
                        hit_j = j;
                    }
                    if (GRAD) hit2 = (z > hit_z) ? hit_z : fmaxf(hit2, z);
                    hit_z = fmaxf(hit_z, z);
                }
            }
            // decided lanes: occlusion already makes valid exactly 0 (only an exact test can change that: a wall whose
            // filters settle every lane costs one wave-level decision, not two)
            if (MODE == MODE_HARD) active = active && (!hit_b || bad);
            else if (MODE == MODE_HSIG) active = active && (hit_c != 6.0f || bad);
            else active = active && (hit_z < 17.5f || bad);
            if (!wave_any(active)) {
                st.shadow = j;
                break;
            }
        }
        more = has_next;
        j = nxt;
        w = wn;
        ++nxt;
        if (nxt == sh) ++nxt;
    }
    }
    // every lane occluded (or off its walls): valid == 0 whatever the loss is
    if (!wave_any(active)) return;

    if (STATS) stat_add<1>(st, 1);
    D2D_WORK(4 * K);
    // ---- path loss, geometry.py:1077-1084 / 641-650 ---------------------------------------
    // The loss of an image-method path is rounding noise (~1e-13) unless the path is degenerate, and it only
    // enters through less(loss, tol).  Certificate: evaluate the specular residuals e_i cheaply (v_rsq, fma);
    // from the same fp32 points the exact chain's e_i differs by at most c = 8e-6 per residual (both chains
    // carry < 35 unit roundoffs per component), so loss_exact <= sum (|e_i| + c)^2 =: bound.  If bound is below
    // loss_skip (hard: tol; approx: half the fp32 spacing below tol, where tol - loss rounds to tol) for every
    // lane that still matters, the exact loss cannot change any output bit and is not computed.
    float loss = 0.0f;
    bool loss_known = (K == 0);
    if (K > 0) {
        const float c = 8e-6f;
        float nvx[K + 1], nvy[K + 1];
#pragma unroll
        for (int i = 0; i <= K; ++i) {
            float vx = px[i + 1] - px[i], vy = py[i + 1] - py[i];
            float inv = __builtin_amdgcn_rsqf(__builtin_fmaf(vx, vx, vy * vy));
            nvx[i] = vx * inv;
            nvy[i] = vy * inv;
        }
        float bound = 0.0f;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float4 r0 = ldc4(a.refl, 2 * late_index(cand[i]));
            float s2 = 2.0f * __builtin_fmaf(nvx[i], r0.z, nvy[i] * r0.w);
            float ex = __builtin_fmaf(s2, r0.z, nvx[i + 1] - nvx[i]);
            float ey = __builtin_fmaf(s2, r0.w, nvy[i + 1] - nvy[i]);
            bound += __builtin_fmaf(2.0f * c, fabsf(ex) + fabsf(ey), __builtin_fmaf(ex, ex, ey * ey)) + c * c;
        }
        const bool certain = (bound * 1.00001f < a.loss_skip) && !bad;
        loss_known = !wave_any(!certain && active);
    }
    if (!loss_known) {
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float4 r0 = ldc4(a.refl, 2 * late_index(cand[i]));
            float ix, iy, rx_, ry_;
            normalize2(px[i + 1] - px[i], py[i + 1] - py[i], ix, iy);
            normalize2(px[i + 2] - px[i + 1], py[i + 2] - py[i + 1], rx_, ry_);
            float din = ix * r0.z + iy * r0.w;
            float s2 = 2.0f * din;
            float ex = rx_ - (ix - s2 * r0.z);
            float ey = ry_ - (iy - s2 * r0.w);
            loss = loss + (ex * ex + ey * ey);
        }
        if (STATS) stat_add<7>(st, K);
    }
    bool ok_b = loss < a.tol;                                    // hard: jnp.less
    float ok_x = a.tol - loss;                                   // approx: activation(tol - loss)
    if (MODE != MODE_HARD) nanflag = nanflag || (loss != loss);

    // ---- is_valid, geometry.py:947-963 -----------------------------------------------------
    float valid;
    float on_v = 1.0f, nh_v = 1.0f, ok_v = 1.0f;
    if (MODE == MODE_HARD) {
        valid = (on_b && !hit_b && ok_b) ? 1.0f : 0.0f;
    } else if (MODE == MODE_HSIG) {
        on_v = on_c / 6.0f;
        float hit_v = hit_c / 6.0f;
        ok_v = clampact(ok_x, a.alpha) / 6.0f;
        nh_v = 1.0f - hit_v;
        valid = fminf(fminf(on_v, nh_v), ok_v);
        valid = nanflag ? 0.0f : valid;  // NaN-propagating min, then nan_to_num
    } else {
        on_v = (K == 0) ? 1.0f : sigmoidf_(on_z);
        float hit_v = any_test ? fmaxf(0.0f, sigmoidf_(hit_z)) : 0.0f;
        ok_v = sigmoidf_(a.alpha * ok_x);
        nh_v = 1.0f - hit_v;
        valid = fminf(fminf(on_v, nh_v), ok_v);
        valid = nanflag ? 0.0f : valid;
    }

    // valid is exactly 0 in every lane (all occluded): acc + 0 * fun == acc, and every adjoint is 0
    if (!wave_any(valid != 0.0f || bad)) return;
    if (STATS) stat_add<3>(st, 1), stat_add<8>(st, K + 1);
    D2D_WORK(6);
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
    else if (GRAD && a.fun_id == D2D_FUN_CUSTOM) f = a.cust_f[cust];
    else f = 1.0f;
    acc = acc + valid * f;  // scene.py:1909

    if (GRAD) {
        // =========================== reverse mode, hand derived ===========================
        // acc += valid * f   ->   fbar = valid, vbar = f (valid is a constant in hard mode / after nan_to_num(NaN))
        const float fbar = valid;
        const float vbar = (MODE == MODE_HARD || nanflag) ? 0.0f : f;
        float rbar;
        if (a.fun_id == D2D_FUN_RECEIVED_POWER) {
            float Dn = a.h2 + r * r;
            rbar = -(fbar * (f / Dn)) * (2.0f * r);
        } else if (a.fun_id == D2D_FUN_LENGTH_SQUARED) rbar = fbar * (2.0f * r);
        else if (a.fun_id == D2D_FUN_LENGTH) rbar = fbar;
        else rbar = 0.0f;

        float pbx[K + 2], pby[K + 2];
#pragma unroll
        for (int i = 0; i < K + 2; ++i) pbx[i] = pby[i] = 0.0f;
        if (a.fun_id == D2D_FUN_CUSTOM) {
            // the host's d fun / d xys (a derivative w.r.t. the end points as arguments of `fun` folded into rows 0 and K + 1)
            const float* pb = a.cust_pb + cust * (2 * (D2D_MAX_ORDER + 2));
#pragma unroll
            for (int i = 0; i < K + 2; ++i) {
                pbx[i] = fbar * pb[2 * i];
                pby[i] = fbar * pb[2 * i + 1];
            }
        }
        // path_length -- for the path functions that go through it: with fun = 1 (or a host-evaluated function, whose own
        // derivative w.r.t. the points arrives in cust_pb) the reference never evaluates a length, and 0 * (w / |w|) is NaN, not 0,
        // for a segment vector of exactly (-eps, -eps) (found by scripts/fuzz_parity.py --grad, seed 3 case 168: a transmitter on
        // the end point of a wall, sigmoid validity, fun = 1 -- false NaN cells in both value+grad kernels since round 1)
        if (a.fun_id != D2D_FUN_ONE && a.fun_id != D2D_FUN_CUSTOM) {
#pragma unroll
            for (int i = 0; i <= K; ++i) {
                float wx = (px[i + 1] - px[i]) + D2D_EPS, wy = (py[i + 1] - py[i]) + D2D_EPS;
                float len = sqrtf(wx * wx + wy * wy);
                float gx = rbar * (wx / len), gy = rbar * (wy / len);
                pbx[i + 1] += gx; pby[i + 1] += gy;
                pbx[i] -= gx; pby[i] -= gy;
            }
        }
        // adjoints of the candidate's walls: origin, normal, direction t (the latter through on_objects only)
        constexpr int KK = (K > 0) ? K : 1;
        float obx[KK], oby[KK], nbx[KK], nby[KK], tbx[KK], tby[KK];
#pragma unroll
        for (int i = 0; i < KK; ++i) obx[i] = oby[i] = nbx[i] = nby[i] = tbx[i] = tby[i] = 0.0f;
        // adjoint of the selected occluder (per lane): patched origin P1 and A = P2 - P1
        float p1bx = 0.0f, p1by = 0.0f, abx = 0.0f, aby = 0.0f;
        bool occ = false;

        if (MODE != MODE_HARD) {
            // valid = minimum(minimum(on, not hit), ok) (logic.py:490-512, a left fold): jnp.minimum hands its cotangent to the
            // smaller argument and splits it evenly at a tie (logic.py:358)
            float w_on = (on_v < nh_v) ? 1.0f : (nh_v < on_v) ? 0.0f : 0.5f;
            float w_nh = 1.0f - w_on, w_ok = 0.0f;
            {
                const float m01 = fminf(on_v, nh_v);
                const float keep = (m01 < ok_v) ? 1.0f : (ok_v < m01) ? 0.0f : 0.5f;
                w_on *= keep;
                w_nh *= keep;
                w_ok = 1.0f - keep;
            }
            const float vb_on = vbar * w_on, vb_nh = vbar * w_nh, vb_ok = vbar * w_ok;
            // ---- ok = activation(tol - loss)
            float lossbar = (w_ok != 0.0f) ? -(vb_ok * dact<MODE>(ok_x, a.alpha)) : 0.0f;
            if (K > 0 && wave_any(lossbar != 0.0f)) {
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float4 r0 = ldc4(a.refl, 2 * cand[i]);
                    float v1x = px[i + 1] - px[i], v1y = py[i + 1] - py[i];
                    float v2x = px[i + 2] - px[i + 1], v2y = py[i + 2] - py[i + 1];
                    float ix, iy, rx_, ry_;
                    normalize2(v1x, v1y, ix, iy);
                    normalize2(v2x, v2y, rx_, ry_);
                    float din = ix * r0.z + iy * r0.w;
                    float s2 = 2.0f * din;
                    float ex = rx_ - (ix - s2 * r0.z), ey = ry_ - (iy - s2 * r0.w);
                    float ebx = 2.0f * ex * lossbar, eby = 2.0f * ey * lossbar;
                    // e = r - i + s2 n
                    float rbx = ebx, rby = eby;
                    float ibx = -ebx, iby = -eby;
                    float s2b = ebx * r0.z + eby * r0.w;
                    nbx[i] += s2 * ebx; nby[i] += s2 * eby;
                    float dinb = 2.0f * s2b;
                    ibx += dinb * r0.z; iby += dinb * r0.w;
                    nbx[i] += dinb * ix; nby[i] += dinb * iy;
                    float a1x, a1y, a2x, a2y;
                    normalize2_bwd(v1x, v1y, ibx, iby, a1x, a1y);
                    normalize2_bwd(v2x, v2y, rbx, rby, a2x, a2y);
                    pbx[i + 1] += a1x; pby[i + 1] += a1y; pbx[i] -= a1x; pby[i] -= a1y;
                    pbx[i + 2] += a2x; pby[i + 2] += a2y; pbx[i + 1] -= a2x; pby[i + 1] -= a2y;
                }
            }
            // ---- on_objects: the activation carrying the min
            if (K > 0) {
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float4 r0 = ldc4(a.refl, 2 * cand[i]);
                    const float4 r1 = ldc4(a.refl, 2 * cand[i] + 1);
                    float dx = px[i + 1] - r0.x, dy = py[i + 1] - r0.y;
                    float s = (r1.x * dx + r1.y * dy) / r1.z;
                    float x = on_w ? (1.0f - s) : (s - 0.0f);
                    float sb = (w_on != 0.0f && i == on_i) ? vb_on * dact<MODE>(x, a.alpha) * (on_w ? -1.0f : 1.0f) : 0.0f;
                    float q = sb / r1.z;
                    pbx[i + 1] += q * r1.x; pby[i + 1] += q * r1.y;
                    obx[i] -= q * r1.x; oby[i] -= q * r1.y;
                    // t enters the numerator and sq = t.t (a constant 1 when the wall is degenerate)
                    bool degenerate = (r1.x * r1.x + r1.y * r1.y == 0.0f);
                    float sqb = degenerate ? 0.0f : -(q * s);
                    tbx[i] += q * dx + 2.0f * sqb * r1.x;
                    tby[i] += q * dy + 2.0f * sqb * r1.y;
                }
            }
            // ---- not(intersects): the test carrying the max, and inside it the activation carrying the min
            occ = (w_nh != 0.0f) && (hit_j >= 0) && (vb_nh != 0.0f);
            // The adjoint of ONE occlusion test (segment i, object jj) that carries `wgt` of the cotangent of `hit`: inside the test,
            // minimum(minimum(ge(ta), le(ta)), minimum(ge(tb), le(tb))) (geometry.py:163-173), ties split evenly again.
            auto occluder_adjoint = [&](auto ic, const float4& w, float wgt, float& o_p1bx, float& o_p1by, float& o_abx, float& o_aby) {
                constexpr int i = decltype(ic)::value;
                const float qx = px[i], qy = py[i], q1x = px[i + 1], q1y = py[i + 1];  // P3, P4
                float Bx = qx - q1x, By = qy - q1y;
                float Cx = w.x - qx, Cy = w.y - qy;
                float fa = By * Cx - Bx * Cy, fb = w.z * Cy - w.w * Cx, fd = w.w * Bx - w.z * By;
                bool dz = (fd == 0.0f);
                float dd = dz ? 1.0f : fd;
                float ta = dz ? __builtin_inff() : fa / dd, tb = dz ? __builtin_inff() : fb / dd;
                float x0 = ta - a.seg_lo, x1 = a.seg_hi - ta, x2 = tb - a.seg_lo, x3 = a.seg_hi - tb;
                float m0, m1, m2, m3;
                if (MODE == MODE_HSIG) { m0 = clampact(x0, a.alpha); m1 = clampact(x1, a.alpha); m2 = clampact(x2, a.alpha); m3 = clampact(x3, a.alpha); }
                else { m0 = a.alpha * x0; m1 = a.alpha * x1; m2 = a.alpha * x2; m3 = a.alpha * x3; }
                // share of the test's cotangent per activation: minimum(minimum(ge(ta), le(ta)), minimum(ge(tb), le(tb))), the
                // reference compares the ACTIVATIONS (in sigmoid mode two pre-activations may round to one float) and splits ties
                if (MODE == MODE_SIG) { m0 = sigmoidf_(m0); m1 = sigmoidf_(m1); m2 = sigmoidf_(m2); m3 = sigmoidf_(m3); }
                const float a0 = (m0 < m1) ? 1.0f : (m1 < m0) ? 0.0f : 0.5f, b0 = (m2 < m3) ? 1.0f : (m3 < m2) ? 0.0f : 0.5f;
                const float mA = fminf(m0, m1), mB = fminf(m2, m3);
                const float ab = (mA < mB) ? 1.0f : (mB < mA) ? 0.0f : 0.5f;
                const float k0 = ab * a0, k1 = ab * (1.0f - a0), k2 = (1.0f - ab) * b0, k3 = (1.0f - ab) * (1.0f - b0);
                // valid = ... 1 - hit ... : d valid / d hit = -1
                const float hw = (wgt != 0.0f && !dz) ? -(vb_nh * wgt) : 0.0f;
                const float h0 = (k0 != 0.0f) ? hw * k0 * dact<MODE>(x0, a.alpha) : 0.0f, h1 = (k1 != 0.0f) ? hw * k1 * dact<MODE>(x1, a.alpha) : 0.0f;
                const float h2 = (k2 != 0.0f) ? hw * k2 * dact<MODE>(x2, a.alpha) : 0.0f, h3 = (k3 != 0.0f) ? hw * k3 * dact<MODE>(x3, a.alpha) : 0.0f;
                float tab = h0 - h1, tbb = h2 - h3;
                float fab = tab / dd, fbb = tbb / dd;
                float fdb = -(tab * ta + tbb * tb) / dd;
                fdb = (tab != 0.0f || tbb != 0.0f) ? fdb : 0.0f;
                // fa = By Cx - Bx Cy ; fb = Ax Cy - Ay Cx ; fd = Ay Bx - Ax By
                float Bbx = -fab * Cy + fdb * w.w, Bby = fab * Cx - fdb * w.z;
                float Cbx = fab * By - fbb * w.w, Cby = -fab * Bx + fbb * w.z;
                o_abx = fbb * Cy - fdb * By;
                o_aby = -fbb * Cx + fdb * Bx;
                o_p1bx = Cbx; o_p1by = Cby;
                pbx[i] += Bbx - Cbx; pby[i] += Bby - Cby;  // d/d P3
                pbx[i + 1] -= Bbx; pby[i + 1] -= Bby;      // d/d P4
            };
            // Do two or more tests tie for the maximum?  (hard_sigmoid: equal clamped pre-activations; sigmoid: pre-activations that
            // round to the same float under the activation -- common next to saturation, where the fp32 sigmoid has 2^-24 steps)
            bool tie = false;
            float vmax = 0.0f;
            if (wave_any(occ)) {
                // (tests that tie at a SATURATED value -- two occluders that each hide the lane completely -- have zero derivatives:
                // nothing to split, and they are common at shadow boundaries)
                if (MODE == MODE_HSIG) {
                    vmax = hit_c;
                    tie = occ && hit2 == hit_c && hit_c < 6.0f;
                } else {
                    vmax = sigmoidf_(hit_z);
                    // (... and so have tests that tie at a value saturated to exactly 0 -- hit2 starts at -3e38, sigmoid 0: without
                    // the lower guard an unoccluded lane whose only test rounds to 0 sent its wave through both passes for nothing)
                    tie = occ && vmax < 1.0f && vmax > 0.0f && sigmoidf_(hit2) == vmax;
                }
            }
            if (wave_any(tie)) {
                // jnp.maximum in a left fold over (segment, object) (geometry.py:881-904): of r tying tests the q-th in the fold's
                // order carries 2^-(r - q + 1) of the cotangent, the first as much as the second.  One pass counts, one applies;
                // lanes without a tie take their single test with weight 1 -- the same arithmetic as the path below.
                int r = 0, q = 0;
                for (int pass = 0; pass < 2; ++pass) {
                    static_for<0, K + 1>([&](auto ic) {
                        constexpr int i = decltype(ic)::value;
                        const int ig0 = (i == 0) ? -1 : cand[i - 1];
                        const int ig1 = (i == K) ? -1 : cand[i];
                        for (int jj = 0; jj < a.N; ++jj) {
                            if (jj == ig0 || jj == ig1) continue;
                            const float4 w = ldc4(a.occl, jj);
                            const float Cx = w.x - px[i], Cy = w.y - py[i];
                            const float fa = by[i] * Cx - bx[i] * Cy, fb = w.z * Cy - w.w * Cx, fd = w.w * bx[i] - w.z * by[i];
                            const bool dz = (fd == 0.0f);
                            float ta, tb;
                            div2_exact(fa, fb, dz ? 1.0f : fd, ta, tb);
                            ta = dz ? __builtin_inff() : ta;
                            tb = dz ? __builtin_inff() : tb;
                            float v;
                            if (MODE == MODE_HSIG)
                                v = fminf(fminf(clampact(ta - a.seg_lo, a.alpha), clampact(a.seg_hi - ta, a.alpha)),
                                          fminf(clampact(tb - a.seg_lo, a.alpha), clampact(a.seg_hi - tb, a.alpha)));
                            else
                                v = sigmoidf_(fminf(fminf(a.alpha * (ta - a.seg_lo), a.alpha * (a.seg_hi - ta)),
                                                    fminf(a.alpha * (tb - a.seg_lo), a.alpha * (a.seg_hi - tb))));
                            const bool t = occ && (v == vmax);
                            if (pass == 0) {
                                r += t ? 1 : 0;
                            } else if (wave_any(t)) {
                                q += t ? 1 : 0;
                                const int sh_ = (q == 1) ? r - 1 : r - q + 1;
                                const float wgt = t ? __builtin_ldexpf(1.0f, -sh_) : 0.0f;
                                float tp1x, tp1y, tax, tay;
                                occluder_adjoint(ic, w, wgt, tp1x, tp1y, tax, tay);
                                if (g->scene) {
                                    // P1 = (1 + patch) o - patch d ; P2 = (1 + patch) d - patch o ; A = P2 - P1
                                    const float P1bx = tp1x - tax, P1by = tp1y - tay, pa = a.patch;
                                    const float s0 = wave_sum(g->cot * ((1.0f + pa) * P1bx - pa * tax)), s1 = wave_sum(g->cot * ((1.0f + pa) * P1by - pa * tay));
                                    const float s2_ = wave_sum(g->cot * ((1.0f + pa) * tax - pa * P1bx)), s3 = wave_sum(g->cot * ((1.0f + pa) * tay - pa * P1by));
                                    if ((threadIdx.x & 63) == 0) {
                                        float* w4 = g->wl + 4 * jj;
                                        atomicAdd(&w4[0], s0); atomicAdd(&w4[1], s1); atomicAdd(&w4[2], s2_); atomicAdd(&w4[3], s3);
                                    }
                                }
                            }
                        }
                    });
                }
                occ = false;  // (applied, the objects' adjoints included)
            } else if (wave_any(occ)) {
                const int jj = occ ? hit_j : 0;
                const float4 w = ldc4(a.occl, jj);
                static_for<0, K + 1>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    if (wave_any(occ && hit_i == i)) {
                        float tp1x, tp1y, tax, tay;
                        occluder_adjoint(ic, w, (occ && hit_i == i) ? 1.0f : 0.0f, tp1x, tp1y, tax, tay);
                        if (occ && hit_i == i) { p1bx = tp1x; p1by = tp1y; abx = tax; aby = tay; }
                    }
                });
            }
        }

        // ---- image method backward scan (reverse of geometry.py:1093-1110) -----------------
        float ibx_[KK], iby_[KK];  // adjoints of the images
#pragma unroll
        for (int i = 0; i < KK; ++i) ibx_[i] = iby_[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float4 r0 = ldc4(a.refl, 2 * cand[i]);
            const float ptx = px[i + 2], pty = py[i + 2];  // the point the step started from
            float ux = ptx - imgx[i], uy = pty - imgy[i];
            float vx = r0.x - ptx, vy = r0.y - pty;
            float un = ux * r0.z + uy * r0.w;
            float vn = vx * r0.z + vy * r0.w;
            const float qbx = pbx[i + 1], qby = pby[i + 1];
            float ptbx = qbx, ptby = qby;  // p[i+1] = pt + inc
            if (un != 0.0f) {
                float incx = (vn * ux) / un, incy = (vn * uy) / un;
                float mbx = qbx / un, mby = qby / un;
                float unb = -(qbx * incx + qby * incy) / un;
                float vnb = mbx * ux + mby * uy;
                float ubx = mbx * vn + unb * r0.z, uby = mby * vn + unb * r0.w;
                nbx[i] += unb * ux + vnb * vx; nby[i] += unb * uy + vnb * vy;
                float vbx = vnb * r0.z, vby = vnb * r0.w;
                ptbx += ubx - vbx; ptby += uby - vby;
                ibx_[i] -= ubx; iby_[i] -= uby;
                obx[i] += vbx; oby[i] += vby;
            }
            pbx[i + 2] += ptbx; pby[i + 2] += ptby;
        }
        // ---- forward image chain backward (reverse of geometry.py:1086-1091) ----------------
        float txbx = pbx[0], txby = pby[0];
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
            const float4 r0 = ldc4(a.refl, 2 * cand[i]);
            const float prx = (i == 0) ? txx : imgx[i > 0 ? i - 1 : 0];
            const float pry = (i == 0) ? txy : imgy[i > 0 ? i - 1 : 0];
            float wx = prx - r0.x, wy = pry - r0.y;
            float dn = wx * r0.z + wy * r0.w;
            float s2 = 2.0f * dn;
            float gx = ibx_[i], gy = iby_[i];
            float s2b = -(gx * r0.z + gy * r0.w);
            nbx[i] += -s2 * gx; nby[i] += -s2 * gy;
            float dnb = 2.0f * s2b;
            float wbx = dnb * r0.z, wby = dnb * r0.w;
            nbx[i] += dnb * wx; nby[i] += dnb * wy;
            float prbx = gx + wbx, prby = gy + wby;
            obx[i] -= wbx; oby[i] -= wby;
            if (i == 0) { txbx += prbx; txby += prby; }
            else { ibx_[i > 0 ? i - 1 : 0] += prbx; iby_[i > 0 ? i - 1 : 0] += prby; }
        }
        g->grx += TXG ? txbx : pbx[K + 1];
        g->gry += TXG ? txby : pby[K + 1];

        if (g->scene) {
            g->tbx += g->cot * (TXG ? pbx[K + 1] : txbx);  // adjoint of the fixed (wave-uniform) end point
            g->tby += g->cot * (TXG ? pby[K + 1] : txby);
            // normals -> wall end points: n = m / len, m = (t_y, -t_x); t = dest - origin
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const float4 r0 = ldc4(a.refl, 2 * cand[i]);
                const float4 r1 = ldc4(a.refl, 2 * cand[i] + 1);
                float len = r1.w;  // |t| guarded to 1
                bool z = (r1.x * r1.x + r1.y * r1.y == 0.0f);
                float d = z ? 0.0f : (nbx[i] * r0.z + nby[i] * r0.w);
                float mbx = (nbx[i] - d * r0.z) / len, mby = (nby[i] - d * r0.w) / len;
                float ttx = tbx[i] - mby, tty = tby[i] + mbx;  // m_x = t_y, m_y = -t_x
                float dbx = ttx, dby = tty;
                float ox_ = obx[i] - ttx, oy_ = oby[i] - tty;
                float s0 = wave_sum(g->cot * ox_), s1 = wave_sum(g->cot * oy_);
                float s2_ = wave_sum(g->cot * dbx), s3 = wave_sum(g->cot * dby);
                if ((threadIdx.x & 63) == 0) {
                    float* w4 = g->wl + 4 * cand[i];
                    w4[0] += s0; w4[1] += s1; w4[2] += s2_; w4[3] += s3;
                }
            }
            if (MODE != MODE_HARD && wave_any(occ)) {
                // P1 = (1 + patch) o - patch d ; P2 = (1 + patch) d - patch o ; A = P2 - P1
                float P2bx = abx, P2by = aby;
                float P1bx = p1bx - abx, P1by = p1by - aby;
                float pa = a.patch;
                float ox_ = (1.0f + pa) * P1bx - pa * P2bx, oy_ = (1.0f + pa) * P1by - pa * P2by;
                float dx_ = (1.0f + pa) * P2bx - pa * P1bx, dy_ = (1.0f + pa) * P2by - pa * P1by;
                if (occ) {
                    float* w4 = g->wl + 4 * hit_j;
                    atomicAdd(&w4[0], g->cot * ox_);
                    atomicAdd(&w4[1], g->cot * oy_);
                    atomicAdd(&w4[2], g->cot * dx_);
                    atomicAdd(&w4[3], g->cot * dy_);
                }
            }
        }
    }
}

// All candidates of order K in lexicographic order (scene.py:122-175), images built incrementally
// (geometry.py:1086-1091, 1109).
template <int K, int MODE, bool STATS, bool GRAD = false, bool TXG = false>
__device__ __forceinline__ void sweep_order(const SweepArgs& a, float txx, float txy, float rxx, float rxy, bool lane_bad,
                                            float& acc, WaveStats& st, GradCtx* g = nullptr) {
    int cand[D2D_MAX_ORDER] = {-1, -1, -1, -1};
    float imgx[D2D_MAX_ORDER], imgy[D2D_MAX_ORDER];
    if (K == 0) {
        eval_candidate<0, MODE, STATS, GRAD, true, TXG>(a, cand, imgx, imgy, txx, txy, rxx, rxy, lane_bad, acc, st, g);
        return;
    }
    for (int i0 = 0; i0 < a.Nc; ++i0) {
        cand[0] = cmem(a.cw)[i0];
        image_of(ldc4(a.refl, 2 * cand[0]), txx, txy, imgx[0], imgy[0]);
        if (K == 1) {
            eval_candidate<K, MODE, STATS, GRAD, true, TXG>(a, cand, imgx, imgy, txx, txy, rxx, rxy, lane_bad, acc, st, g);
            continue;
        }
        for (int i1 = 0; i1 < a.Nc; ++i1) {
            cand[1] = cmem(a.cw)[i1];
            if (cand[1] == cand[0]) continue;
            image_of(ldc4(a.refl, 2 * cand[1]), imgx[0], imgy[0], imgx[1], imgy[1]);
            if (K == 2) {
                eval_candidate<K, MODE, STATS, GRAD, true, TXG>(a, cand, imgx, imgy, txx, txy, rxx, rxy, lane_bad, acc, st, g);
                continue;
            }
            for (int i2 = 0; i2 < a.Nc; ++i2) {
                cand[2] = cmem(a.cw)[i2];
                if (cand[2] == cand[1]) continue;
                image_of(ldc4(a.refl, 2 * cand[2]), imgx[1], imgy[1], imgx[2], imgy[2]);
                if (K == 3) {
                    eval_candidate<K, MODE, STATS, GRAD, true, TXG>(a, cand, imgx, imgy, txx, txy, rxx, rxy, lane_bad, acc, st, g);
                    continue;
                }
                for (int i3 = 0; i3 < a.Nc; ++i3) {
                    cand[3] = cmem(a.cw)[i3];
                    if (cand[3] == cand[2]) continue;
                    image_of(ldc4(a.refl, 2 * cand[3]), imgx[2], imgy[2], imgx[3], imgy[3]);
                    eval_candidate<(K >= 4 ? 4 : K), MODE, STATS, GRAD, true, TXG>(a, cand, imgx, imgy, txx, txy, rxx, rxy, lane_bad, acc, st, g);
                }
            }
        }
    }
}

// =====================================================================================
// Candidate-parallel tile culling.
//
// A wave owns an 8 x 8 patch of RX cells.  Before the cells (one per lane) walk the candidates of a
// prefix (w_0 .. w_{K-2}) one by one, the wave turns its lanes around: lane l takes the candidate whose
// LAST wall is the l-th allowed object, and decides -- conservatively, for the whole patch at once --
// whether on_objects can be anything but exactly 0 for any cell of the patch.  Survivors come back as a
// ballot mask and are then evaluated exactly, in the reference's order, by eval_candidate.
//
// The test: the parametric coordinate s of the point where the line (rx -> image) meets the wall's line
// is a linear-fractional function of rx; over a convex region that does not meet its pole line
// (un = (rx - image).n = 0) it is monotone along every segment, so its range over the region is spanned
// by the region's vertices.  Level 1 evaluates the 4 corners of the patch's bounding box against the last
// wall; level j > 1 takes the sub-segment of wall K-j+1 that level j-1 left possible (widened by the
// rounding bound, as a thin quad around the wall) and evaluates its 4 vertices against wall K-j.
// A candidate is dropped only if some level proves s outside the window where the activation is not
// exactly saturated to 0 (hard: [0,1]; hard_sigmoid: widened by 3/alpha; sigmoid: by 89/alpha), with an
// explicit bound M on |s_fp32(exact path) - s_real| -- so dropping it cannot change a bit of the output.
// =====================================================================================
struct WallC {  // what the culling needs of a wall
    float ox, oy, nx, ny, tx, ty, rsq;
    int idx;
};

__device__ __forceinline__ WallC make_wallc(const float4& r0, const float4& r1, const float4& fc, int idx) {
    return WallC{r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, fc.x, idx};
}

// Range of s over the convex hull of 4 points; returns false when the region may meet the pole line or
// anything is not comfortably finite (then nothing may be concluded).
__device__ __forceinline__ bool s_range(const float (&qx)[4], const float (&qy)[4], float Ix, float Iy, const WallC& w,
                                        float& smin, float& smax, float& M, float& E) {
    const float eps = 1.1920929e-07f;
    bool pos = true, neg = true, fin = true;
    smin = __builtin_inff();
    smax = -__builtin_inff();
    E = 0.0f;
    // partial unroll: the fully unrolled body needs ~35 more VGPRs (85 vs 51) and caps the kernel at 5 waves per SIMD
#pragma unroll 2
    for (int j = 0; j < 4; ++j) {
        float ux = qx[j] - Ix, uy = qy[j] - Iy;
        float vx = w.ox - qx[j], vy = w.oy - qy[j];
        float un = __builtin_fmaf(ux, w.nx, uy * w.ny);
        float vn = __builtin_fmaf(vx, w.nx, vy * w.ny);
        // bound on |un_fp32 - un_real| for any evaluation order of this expression
        float du = 8.0f * eps * __builtin_fmaf(fabsf(w.nx), fabsf(qx[j]) + fabsf(Ix), fabsf(w.ny) * (fabsf(qy[j]) + fabsf(Iy)));
        pos = pos && (un > du);
        neg = neg && (un < -du);
        float g = vn * __builtin_amdgcn_rcpf(un);
        float dx = __builtin_fmaf(g, ux, -vx), dy = __builtin_fmaf(g, uy, -vy);
        float s = __builtin_fmaf(w.ty, dy, w.tx * dx) * w.rsq;
        float mag = __builtin_fmaf(fabsf(g), fabsf(ux) + fabsf(uy), fabsf(vx) + fabsf(vy));
        fin = fin && (mag < 1e18f);
        smin = fminf(smin, s);
        smax = fmaxf(smax, s);
        E = fmaxf(E, mag);
    }
    E = E + (fabsf(w.ox) + fabsf(w.oy)) + (fabsf(Ix) + fabsf(Iy));
    M = __builtin_fmaf(64.0f * eps * w.rsq * (fabsf(w.tx) + fabsf(w.ty)), 2.0f * E, 1e-30f);
    return (pos || neg) && fin;
}

// The pole part of s_range on its own: may un = (rx - image) . n vanish (or be non-finite) somewhere in the patch?
__device__ __forceinline__ bool pole_possible(const float (&qx)[4], const float (&qy)[4], float Ix, float Iy, float nx, float ny) {
    const float eps = 1.1920929e-07f;
    bool pos = true, neg = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float ux = qx[j] - Ix, uy = qy[j] - Iy;
        float un = __builtin_fmaf(ux, nx, uy * ny);
        float du = 8.0f * eps * __builtin_fmaf(fabsf(nx), fabsf(qx[j]) + fabsf(Ix), fabsf(ny) * (fabsf(qy[j]) + fabsf(Iy)));
        pos = pos && (un > du);
        neg = neg && (un < -du);
    }
    return !(pos || neg);
}

// true = the candidate is certainly invalid for every cell of the patch.
// walls[j], images[j] for j = K-1 (last wall) down to 0; level 1 uses the patch box.
// WIDE: the exact path this culls for runs the chain in the other direction (TX grids: images of the CELL, backward scan
// from the fixed end point), so its interaction points are the same geometric points with a different rounding history:
// the bound M, derived for the chain evaluated here, is taken four times as wide.
template <int K, bool WIDE = false>
__device__ __forceinline__ bool cull_candidate(const float (&bx)[4], const float (&by)[4], const WallC (&w)[K],
                                               const float (&Ix)[K], const float (&Iy)[K], const SweepArgs& a,
                                               unsigned long long shadow0, float on_lo, float on_hi,
                                               unsigned long long hiddenK = 0ull, float hidden_dperp = 0.0f) {
    const float eps = 1.1920929e-07f;
    const float shadow_dperp = a.shadow_dperp, shadow_lo = a.shadow_lo, shadow_inv = a.shadow_inv;
    float qx[4], qy[4];
    // 8-bin range that the previous (later-in-path) wall's interaction point can occupy, for the wall-to-wall masks
    int pka = 0, pkb = -1;
    bool prev_ok = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        qx[j] = bx[j];
        qy[j] = by[j];
    }
#pragma unroll
    for (int lvl = K - 1; lvl >= 0; --lvl) {
        float smin, smax, M, E;
        bool ok = s_range(qx, qy, Ix[lvl], Iy[lvl], w[lvl], smin, smax, M, E);
        if (!ok) return false;
        if (WIDE) M *= 4.0f;
        if (smax + M < on_lo || smin - M > on_hi) return true;
        if (lvl == K - 1 && hiddenK != 0ull) {
            // The segment between the cell and the wall next to it (RX grids: last wall -> cell; TX grids, WIDE: cell -> first
            // wall, masks built with the roles swapped): if every point that interaction can occupy is hidden from the whole
            // region the box lies in, the segment is occluded in every lane: valid == 0 (hidden_region_kernel; the bins are
            // the shadow masks').
            float sa = fmaxf(smin - M, on_lo) - 1e-4f, sb = fminf(smax + M, on_hi) + 1e-4f;
            float fa_ = (sa - shadow_lo) * shadow_inv, fb_ = (sb - shadow_lo) * shadow_inv;
            if (fa_ >= 0.0f && fb_ < 64.0f && (WIDE ? 1024.0f : 256.0f) * eps * E <= hidden_dperp) {
                int ka = (int)fa_, kb = (int)fb_;
                ka = ka < 0 ? 0 : ka;
                kb = kb > 63 ? 63 : kb;
                unsigned long long need = (kb >= 63 ? ~0ull : ((1ull << (kb + 1)) - 1ull)) & ~((1ull << ka) - 1ull);
                if ((hiddenK & need) == need) return true;
            }
        }
        if (K >= 2 && a.pair) {
            // Segment between this wall's interaction point and the next one's (pair_shadow_kernel): if every pair of
            // bins the two points can occupy is certainly occluded by some third object, valid == 0 in every lane.
            float sa8 = fmaxf(smin - M, on_lo) - 1e-4f, sb8 = fminf(smax + M, on_hi) + 1e-4f;
            float fa8 = (sa8 - shadow_lo) * (0.125f * shadow_inv), fb8 = (sb8 - shadow_lo) * (0.125f * shadow_inv);
            const bool cur_ok = fa8 >= 0.0f && fb8 < 8.0f && 512.0f * eps * E <= a.pair_dperp;
            int ka = (int)fa8, kb = (int)fb8;
            ka = ka < 0 ? 0 : ka;
            kb = kb > 7 ? 7 : kb;
            if (lvl < K - 1 && prev_ok && cur_ok) {
                // in path order the earlier point is P3, the later P4 (geometry.py:881-904): rows = later wall's bins
                const int we = WIDE ? w[lvl + 1 < K ? lvl + 1 : lvl].idx : w[lvl].idx;   // earlier wall (P3 side)
                const int wl_ = WIDE ? w[lvl].idx : w[lvl + 1 < K ? lvl + 1 : lvl].idx;  // later wall (P4 side)
                const int ea = WIDE ? pka : ka, eb = WIDE ? pkb : kb;   // earlier wall's bin range
                const int la = WIDE ? ka : pka, lb = WIDE ? kb : pkb;   // later wall's bin range
                const unsigned long long m = cmem(a.pair)[(size_t)we * a.N + wl_];
                const unsigned long long row = (unsigned long long)(((1u << (eb + 1)) - 1u) & ~((1u << ea) - 1u));
                unsigned long long need = row * 0x0101010101010101ull;
                const unsigned long long rows = (lb >= 7 ? ~0ull : ((1ull << (8 * (lb + 1))) - 1ull)) & ~((1ull << (8 * la)) - 1ull);
                need &= rows;
                if ((m & need) == need) return true;
            }
            prev_ok = cur_ok;
            pka = ka;
            pkb = kb;
        }
        if (lvl == 0) {
            // First segment (fixed end point -> first wall): if every point the first interaction can occupy is hidden
            // from the fixed end point by some object, the segment is occluded in every lane: valid == 0.
            float sa = fmaxf(smin - M, on_lo) - 1e-4f, sb = fminf(smax + M, on_hi) + 1e-4f;  // lanes outside the window are invalid anyway
            float fa_ = (sa - shadow_lo) * shadow_inv, fb_ = (sb - shadow_lo) * shadow_inv;
            if (fa_ >= 0.0f && fb_ < 64.0f && 256.0f * eps * E <= shadow_dperp) {
                int ka = (int)fa_, kb = (int)fb_;
                ka = ka < 0 ? 0 : ka;
                kb = kb > 63 ? 63 : kb;
                unsigned long long need = (kb >= 63 ? ~0ull : ((1ull << (kb + 1)) - 1ull)) & ~((1ull << ka) - 1ull);
                if ((shadow0 & need) == need) return true;
            }
            break;
        }
        // what is left of wall `lvl` for the next level: sigma in [sa, sb], as a thin quad around the wall
        float sa = fmaxf(smin - M, on_lo), sb = fminf(smax + M, on_hi);
        float d = 64.0f * eps * 2.0f * E;  // the fp32 point may sit this far off the wall's line
        float eax = __builtin_fmaf(sa, w[lvl].tx, w[lvl].ox), eay = __builtin_fmaf(sa, w[lvl].ty, w[lvl].oy);
        float ebx = __builtin_fmaf(sb, w[lvl].tx, w[lvl].ox), eby = __builtin_fmaf(sb, w[lvl].ty, w[lvl].oy);
        float ddx = d * w[lvl].nx, ddy = d * w[lvl].ny;
        // also pad along the wall: the end points above carry their own rounding
        float px_ = d * w[lvl].tx * w[lvl].rsq * (fabsf(w[lvl].tx) + fabsf(w[lvl].ty)), py_ = d * w[lvl].ty * w[lvl].rsq * (fabsf(w[lvl].tx) + fabsf(w[lvl].ty));
        qx[0] = eax - px_ + ddx; qy[0] = eay - py_ + ddy;
        qx[1] = eax - px_ - ddx; qy[1] = eay - py_ - ddy;
        qx[2] = ebx + px_ + ddx; qy[2] = eby + py_ + ddy;
        qx[3] = ebx + px_ - ddx; qy[3] = eby + py_ - ddy;
    }
    return false;
}

// Ordered list of a wave's non-zero contributions, one column per lane, in LDS (power_fwd_split_kernel).
constexpr int SPLIT_LIST = 16;  // entries per lane; a wave that needs more raises `over` and its range is redone serially
struct ListSink {
    float* col;  // this lane's column: entry i at col[i * 64]
    int cnt;
    bool over;
    int cap = SPLIT_LIST;
    bool through = false;  // the list is read by another workgroup: store past this XCD's L2 (agent scope, sc1)
    __device__ __forceinline__ void push(float v) {
        if (cnt < cap) {
            if (through) __hip_atomic_store(reinterpret_cast<int*>(col + cnt * 64), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else col[cnt * 64] = v;
            ++cnt;
        } else {
            over = true;
        }
    }
};

// Survivors of a region's culling, in candidate order (region_list_kernel / region_refine_kernel): wave-uniform state
// of the list being written.
struct EmitSink {
    ListPool lp;
    int cur;    // chunk that takes the next entry (initially the list's static first chunk)
    int n;      // entries so far
    bool over;  // the pool ran out: the list is not listed
};

// Appends the codes of the lanes in `mask` (at most 64) in lane order.
__device__ __forceinline__ void emit_batch(EmitSink& e, unsigned long long code, bool alive, unsigned long long mask) {
    const int lane = threadIdx.x & 63;
    const int cntm = __builtin_popcountll(mask);
    if (cntm == 0 || e.over) return;
    const int start = e.n;
    const int in_cur = start & (RL_CHUNK - 1);
    const bool fresh = in_cur == 0 && start > 0;  // the current chunk is full
    int newc = -1;
    if (fresh || in_cur + cntm > RL_CHUNK) {
        int c = 0;
        if (lane == 0) c = atomicAdd(e.lp.head, 1);
        c = __builtin_amdgcn_readfirstlane(c) + e.lp.n_static;
        if (c >= e.lp.max_chunks) {
            e.over = true;
            return;
        }
        newc = c;
        if (lane == 0) e.lp.next[e.cur] = c;
    }
    if (alive) {
        const int pos = start + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
        const int chunk = (!fresh && (pos / RL_CHUNK) == (start / RL_CHUNK)) ? e.cur : newc;
        e.lp.pool[(size_t)chunk * RL_CHUNK + (pos & (RL_CHUNK - 1))] = code;
    }
    if (newc >= 0) e.cur = newc;
    e.n = start + cntm;
}

// All candidates of order K >= 1 with tile culling; `tab` = LDS copy of {refl[2N], flt[N]}.
// K >= 2: only the prefixes whose FIRST position lies in [p_lo, p_hi) (positions into cw[]).  LIST: instead of being
// added to acc, every contribution that is not exactly zero is appended to `sink` (adding an exact zero never changes
// acc: acc is never -0.0), so that another wave can add them later in the reference's order.
// EMIT (K >= 2): nothing is evaluated; the survivors of the full culling test are appended to `emit` instead (the box is
// then a region's, not a patch's).
template <int K, int MODE, bool STATS, bool GRAD = false, bool LIST = false, bool EMIT = false>
__device__ __forceinline__ void sweep_order_culled(const SweepArgs& a, const float4* tab, const float (&bx)[4],
                                                   const float (&by)[4], float rxx, float rxy, bool lane_bad, float& acc,
                                                   WaveStats& st, GradCtx* g = nullptr, int p_lo = 0,
                                                   int p_hi = 0x7fffffff, ListSink* sink = nullptr, EmitSink* emit = nullptr,
                                                   const unsigned long long* hidden_row = nullptr, float hidden_dperp = 0.0f) {
    static_assert(!EMIT || K >= 2, "lists exist for orders >= 2");
    const int lane = threadIdx.x & 63;
    int cand[D2D_MAX_ORDER] = {-1, -1, -1, -1};
    float imgx[D2D_MAX_ORDER], imgy[D2D_MAX_ORDER];
    const int Nc = a.Nc;
    // odometer over the K-1 prefix positions (wave-uniform)
    int pos[D2D_MAX_ORDER] = {0, 0, 0, 0};
    const int n_chunks = (Nc + 63) >> 6;
    // iterate prefixes in lexicographic order
    // first prefix
    const int p_end = p_hi < Nc ? p_hi : Nc;
    if (K >= 2) pos[0] = p_lo;
#pragma unroll
    for (int d = 1; d < K - 1; ++d) pos[d] = (pos[d - 1] == 0) ? 1 : 0;  // no equal neighbours
    if (K - 1 > 0 && (Nc < 2 || p_lo >= p_end)) return;
    if (Nc < 1) return;
    // Forward builds: the first walls that are not wholly in the fixed end point's shadow, as a bit mask over positions;
    // the odometer steps from set bit to set bit (a shadowed first wall costs nothing at all).
    const bool use_fmask = (K >= 2) && Nc <= 256 && a.shadow && a.shadow_prefix_ok;
    unsigned long long fmask[4] = {0ull, 0ull, 0ull, 0ull};
    if (use_fmask) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int p0 = c * 64 + lane;
            fmask[c] = __ballot(p0 < Nc && cmem(a.shadow)[cmem(a.cw)[p0 < Nc ? p0 : 0]] != ~0ull);
        }
    }
    auto next_first = [&](int from) -> int {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (from < (c + 1) * 64) {
                const int sh = from > c * 64 ? from - c * 64 : 0;
                const unsigned long long x = fmask[c] & (~0ull << sh);
                if (x) return c * 64 + __builtin_ctzll(x);
            }
        }
        return Nc;
    };
    if (use_fmask) {
        pos[0] = next_first(p_lo);
        if (pos[0] >= p_end) return;
#pragma unroll
        for (int d = 1; d < K - 1; ++d) pos[d] = (pos[d - 1] == 0) ? 1 : 0;
    }
    // Two-stage culling (forward builds, K >= 2).  Stage 1, per prefix and chunk of last walls: only the level next to
    // the patch (can the last wall's interaction point lie on the wall for any cell?) -- a third of the work at order 2, a
    // quarter at order 3, and most lanes die there.  The survivors are queued, in candidate order, in LDS; whenever 64 are
    // waiting, stage 2 runs the full multi-level test (shadow and wall-to-wall masks included) on a FULL wave of them,
    // and its survivors are evaluated exactly, still in candidate order.
    constexpr bool QUEUE = K >= 2;
    unsigned long long* cullq = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(const_cast<float4*>(tab)) + a.cullq_off) +
                                ((threadIdx.x >> 6) & 3) * 64;  // 64 slots per wave behind the kernel's other LDS data
    int qn = 0;  // wave-uniform
    auto flush = [&]() {
        if (qn == 0) return;
        __builtin_amdgcn_wave_barrier();
        bool alive2 = lane < qn;
        const unsigned long long code = cullq[lane < qn ? lane : 0];
        {
            WallC w[K];
            float Ix[K], Iy[K];
            float ix = a.txx, iy = a.txy;
#pragma unroll
            for (int d = 0; d < K; ++d) {
                const int wd = (int)((code >> (12 * d)) & 0xfffull);
                const float4 r0 = tab[2 * wd], r1 = tab[2 * wd + 1], fc = tab[2 * a.N + wd];
                w[d] = make_wallc(r0, r1, fc, wd);
                image_of(r0, ix, iy, Ix[d], Iy[d]);
                ix = Ix[d];
                iy = Iy[d];
            }
            const unsigned long long sh0 = a.shadow ? cmem(a.shadow)[w[0].idx] : 0ull;
            if (alive2 && cull_candidate<K>(bx, by, w, Ix, Iy, a, sh0, a.on_lo, a.on_hi)) alive2 = false;
        }
        unsigned long long mask = __ballot(alive2);
        if (STATS) stat_add<9>(st, K);
        D2D_WORK(5 * K);
        if constexpr (EMIT) {
            emit_batch(*emit, code, alive2, mask);
            mask = 0ull;
        }
        const unsigned long long te0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
        if constexpr (!EMIT) while (mask) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            const unsigned long long cb = cullq[b];
            const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(cb & 0xffffffffull));
            const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(cb >> 32));
            const unsigned long long cu = ((unsigned long long)hi32 << 32) | lo32;
            int ce[D2D_MAX_ORDER] = {-1, -1, -1, -1};
            float ex[D2D_MAX_ORDER], ey[D2D_MAX_ORDER];
#pragma unroll
            for (int d = 0; d < K; ++d) {
                ce[d] = (int)((cu >> (12 * d)) & 0xfffull);
                image_of(ldc4(a.refl, 2 * ce[d]), d == 0 ? a.txx : ex[d > 0 ? d - 1 : 0], d == 0 ? a.txy : ey[d > 0 ? d - 1 : 0], ex[d], ey[d]);
            }
            if (LIST) {
                float t = 0.0f;
                eval_candidate<K, MODE, STATS, GRAD, false, false>(a, ce, ex, ey, a.txx, a.txy, rxx, rxy, lane_bad, t, st, g);
                if (!(t == 0.0f)) sink->push(t);  // non-zero or NaN
            } else {
                eval_candidate<K, MODE, STATS, GRAD, false, false>(a, ce, ex, ey, a.txx, a.txy, rxx, rxy, lane_bad, acc, st, g);
            }
        }
        if (STATS) stat_add<14>(st, __builtin_amdgcn_s_memtime() - te0);
        qn = 0;
        __builtin_amdgcn_wave_barrier();
    };
    // Order 3, forward builds: the second walls that can follow the current first wall at all (not the same wall, not
    // mutually invisible with it bin for bin) as a bit mask over positions, one 64-lane test per chunk when the first wall
    // changes; the odometer then steps from set bit to set bit instead of visiting all N-1 second walls one by one (two
    // dependent scalar loads each).  Positions beyond 256 walls fall back to plain stepping.
    const bool use_amask = (K == 3) && Nc <= 256 && a.pair && a.pair_prefix_ok;
    unsigned long long amask[4] = {0ull, 0ull, 0ull, 0ull};
    int amask_for = -1;
    auto next_alive = [&](int from) -> int {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (from < (c + 1) * 64) {
                const int sh = from > c * 64 ? from - c * 64 : 0;
                const unsigned long long x = amask[c] & (~0ull << sh);
                if (x) return c * 64 + __builtin_ctzll(x);
            }
        }
        return Nc;
    };
    while (true) {
        bool skip_all = false;
        if (use_amask && amask_for != pos[0]) {
            const int w0 = cmem(a.cw)[pos[0]];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int p1 = c * 64 + lane;
                const bool ok = p1 < Nc && p1 != pos[0] && cmem(a.pair)[(size_t)w0 * a.N + cmem(a.cw)[p1 < Nc ? p1 : 0]] != ~0ull;
                amask[c] = __ballot(ok);
            }
            amask_for = pos[0];
            pos[1] = next_alive(0);
        }
        if (use_amask && pos[1] >= Nc) skip_all = true;  // nothing can follow this first wall
        if (!skip_all) {
        // images of the prefix
#pragma unroll
        for (int d = 0; d < K - 1; ++d) {
            cand[d] = cmem(a.cw)[pos[d]];
            const float4 r0 = ldc4(a.refl, 2 * cand[d]);
            image_of(r0, d == 0 ? a.txx : imgx[d > 0 ? d - 1 : 0], d == 0 ? a.txy : imgy[d > 0 ? d - 1 : 0], imgx[d], imgy[d]);
        }
        const float pIx = (K == 1) ? a.txx : imgx[K >= 2 ? K - 2 : 0];
        const float pIy = (K == 1) ? a.txy : imgy[K >= 2 ? K - 2 : 0];
        const int last_prefix_pos = (K == 1) ? -1 : pos[K >= 2 ? K - 2 : 0];
        // The whole first wall -- over the full parametric window in which on_objects is not exactly 0 -- is hidden from
        // the fixed end point: no candidate starting with it can be valid.  (A valid candidate has un != 0 in every step:
        // un == 0 leaves a zero-length segment, i.e. loss >= 1, which the host checks to be exactly invalid for the
        // current tol / alpha (shadow_prefix_ok); so its first point does lie on the wall's line within rounding.)
        // (The value+grad build skips it too: the reference's autodiff NaN artefacts, which do not care about validity, are
        // found by a pass of their own, d2d_nanscan.hpp.)
        bool prefix_dead = (K >= 2) && a.shadow && a.shadow_prefix_ok && (cmem(a.shadow)[cand[0]] == ~0ull);
        if (K >= 3 && prefix_dead) {
            // the FIRST wall is dead: so are all (N-1)^(K-2) prefixes that start with it -- leave the inner positions at
            // their end so that the odometer below moves straight on to the next first wall
#pragma unroll
            for (int d = 1; d < K - 1; ++d) pos[d] = Nc;
        }
        if (K >= 3 && a.pair && a.pair_prefix_ok) {
            // two consecutive walls of the prefix whose windows are mutually invisible bin for bin: whatever follows,
            // the segment between them is occluded (or one of its ends is off its wall)
#pragma unroll
            for (int d = 0; d + 1 < K - 1; ++d) prefix_dead = prefix_dead || (cmem(a.pair)[(size_t)cand[d] * a.N + cand[d + 1]] == ~0ull);
        }
        for (int chunk = 0; chunk < (prefix_dead ? 0 : n_chunks); ++chunk) {
            // ---- lanes = candidates: lane l <-> last wall = cw[chunk * 64 + l]
            const int lp = chunk * 64 + lane;
            bool alive = (lp < Nc) && (lp != last_prefix_pos);
            if (QUEUE) {
                const int wl = cmem(a.cw)[lp < Nc ? lp : 0];
                const float4 r0 = tab[2 * wl], r1 = tab[2 * wl + 1], fc = tab[2 * a.N + wl];
                const WallC wlast = make_wallc(r0, r1, fc, wl);
                float lx, ly;
                image_of(r0, pIx, pIy, lx, ly);
                if (alive) {
                    float smin, smax, M, E;
                    const bool ok = s_range(bx, by, lx, ly, wlast, smin, smax, M, E);
                    if (ok && (smax + M < a.on_lo || smin - M > a.on_hi)) alive = false;
                }
                const unsigned long long m1 = __ballot(alive);
                const int cnt = __builtin_popcountll(m1);
                if (STATS) stat_add<9>(st, 1);
                D2D_WORK(5);
                if (qn + cnt > 64) flush();
                if (alive) {
                    unsigned long long code = (unsigned long long)wl << (12 * (K - 1));
#pragma unroll
                    for (int d = 0; d < K - 1; ++d) code |= (unsigned long long)cand[d] << (12 * d);
                    cullq[qn + __builtin_popcountll(m1 & ((1ull << lane) - 1ull))] = code;
                }
                qn += cnt;
                continue;
            }
            {
                const int wl = cmem(a.cw)[lp < Nc ? lp : 0];
                WallC w[K];
                float Ix[K], Iy[K];
                const float4 r0 = tab[2 * wl], r1 = tab[2 * wl + 1], fc = tab[2 * a.N + wl];
                w[K - 1] = make_wallc(r0, r1, fc, wl);
                image_of(r0, pIx, pIy, Ix[K - 1], Iy[K - 1]);
#pragma unroll
                for (int d = 0; d < K - 1; ++d) {
                    const int wd = cand[d];
                    w[d] = make_wallc(ldc4(a.refl, 2 * wd), ldc4(a.refl, 2 * wd + 1), ldc4(a.flt, wd), wd);
                    Ix[d] = imgx[d];
                    Iy[d] = imgy[d];
                }
                unsigned long long sh0 = 0ull;
                if (a.shadow) sh0 = cmem(a.shadow)[(K == 1) ? wl : cand[0]];
                if (alive) {
                    // (order 1: the region's last-segment mask of the lane's wall, hidden_region_kernel)
                    const unsigned long long hk = (K == 1 && hidden_row && a.shadow) ? hidden_row[wl] : 0ull;
                    if (cull_candidate<K>(bx, by, w, Ix, Iy, a, sh0, a.on_lo, a.on_hi, hk, hidden_dperp)) alive = false;
                }
            }
            unsigned long long mask = __ballot(alive);
            if (STATS) stat_add<9>(st, K);
            D2D_WORK(5 * K);  // one culling level = 4 vertex evaluations
            const unsigned long long te0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
            // ---- lanes = RX cells: survivors in ascending order (= the reference's order)
            while (mask) {
                const int b = __builtin_ctzll(mask);
                mask &= mask - 1;
                cand[K - 1] = cmem(a.cw)[chunk * 64 + b];
                image_of(ldc4(a.refl, 2 * cand[K - 1]), pIx, pIy, imgx[K - 1], imgy[K - 1]);
                if (LIST) {
                    float t = 0.0f;
                    eval_candidate<K, MODE, STATS, GRAD, false, false>(a, cand, imgx, imgy, a.txx, a.txy, rxx, rxy, lane_bad, t, st, g);
                    if (!(t == 0.0f)) sink->push(t);  // non-zero or NaN
                } else {
                    eval_candidate<K, MODE, STATS, GRAD, false, false>(a, cand, imgx, imgy, a.txx, a.txy, rxx, rxy, lane_bad, acc, st, g);
                }
            }
            if (STATS) stat_add<14>(st, __builtin_amdgcn_s_memtime() - te0);  // exact evaluation of the survivors
        }
        }  // !skip_all
        // next prefix (lexicographic, no equal neighbours); static indexing keeps pos[] in registers
        if (K == 1) break;
        bool carry = true;
        int stop = -1;
#pragma unroll
        for (int d = K - 2; d >= 0; --d) {
            if (carry) {
                if (use_amask && d == 1) {
                    pos[d] = (pos[d] >= Nc) ? Nc : next_alive(pos[d] + 1);
                } else if (use_fmask && d == 0) {
                    pos[d] = next_first(pos[d] + 1);
                } else {
                    pos[d] += 1;
                    if (d > 0 && pos[d] == pos[d - 1]) pos[d] += 1;
                }
                if (pos[d] < (d == 0 ? p_end : Nc)) {
                    carry = false;
                    stop = d;
                }
            }
        }
        if (carry) break;  // the first position overflowed: all prefixes done
#pragma unroll
        for (int e = 1; e < K - 1; ++e)
            if (e > stop) pos[e] = (pos[e - 1] == 0) ? 1 : 0;
    }
    if (QUEUE) flush();
}

// First-wall positions [lo, hi) of part `part` of `parts`, balanced over the first walls the prefix skip does not kill.
__device__ __forceinline__ void first_wall_range(const SweepArgs& a, int part, int parts, int& lo, int& hi) {
    const int lane = threadIdx.x & 63;
    const int Nc = a.Nc;
    const bool use_dead = a.shadow && a.shadow_prefix_ok;
    const int n_chunks = (Nc + 63) >> 6;
    int A = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const int pp = c * 64 + lane;
        const bool alive = pp < Nc && !(use_dead && cmem(a.shadow)[cmem(a.cw)[pp]] == ~0ull);
        A += __builtin_popcountll(__ballot(alive));
    }
    auto boundary = [&](int r) -> int {
        if (r <= 0) return 0;
        if (r >= A) return Nc;
        for (int c = 0; c < n_chunks; ++c) {
            const int pp = c * 64 + lane;
            const bool alive = pp < Nc && !(use_dead && cmem(a.shadow)[cmem(a.cw)[pp]] == ~0ull);
            unsigned long long m = __ballot(alive);
            const int n = __builtin_popcountll(m);
            if (r < n) {
                for (; r > 0; --r) m &= m - 1;
                return c * 64 + __builtin_ctzll(m);
            }
            r -= n;
        }
        return Nc;
    };
    lo = boundary((int)(((long)A * part) / parts));
    hi = (part == parts - 1) ? Nc : boundary((int)(((long)A * (part + 1)) / parts));
}

#ifndef D2D_HIDDEN_PATCH
#define D2D_HIDDEN_PATCH 1  // A/B: 0 = the last-segment masks are consulted for the regions' lists (and order 1) only
#endif
// One batch of a candidate list, lanes = candidates: decodes the lane's entry, builds its image chain and runs the full
// tile-culling test against the box (bx, by).  Returns the ballot of the entries that cannot be dropped.
// TXG (TX grids): the chain is the fixed end point's through the walls in REVERSE order (sweep_order_culled_txg), with the
// wider bound of cull_candidate<K, true>; Ix / Iy are then that chain's images, of no use to the evaluation.
template <int K, bool GRAD, bool TXG = false>
__device__ __forceinline__ unsigned long long cull_batch(const SweepArgs& a, const float4* tab, const float (&bx)[4], const float (&by)[4],
                                                         unsigned long long code, bool have, float (&Ix)[K], float (&Iy)[K], float on_lo,
                                                         float on_hi, const unsigned long long* hidden_row = nullptr, float hidden_dperp = 0.0f) {
    bool alive = have;
    WallC w[K];
    float ix = a.txx, iy = a.txy;
    if constexpr (TXG) {
#pragma unroll
        for (int jx = 0; jx < K; ++jx) {
            const int wd = (int)((code >> (12 * (K - 1 - jx))) & 0xfffull);
            const float4 r0 = tab[2 * wd], r1 = tab[2 * wd + 1], fc = tab[2 * a.N + wd];
            w[jx] = make_wallc(r0, r1, fc, wd);
            image_of(r0, ix, iy, Ix[jx], Iy[jx]);
            ix = Ix[jx];
            iy = Iy[jx];
        }
        if (alive) {
            const unsigned long long sh0 = a.shadow ? cmem(a.shadow)[w[0].idx] : 0ull;
            const unsigned long long hk = (hidden_row && a.shadow) ? hidden_row[w[K - 1].idx] : 0ull;  // (the wall next to the cell)
            if (cull_candidate<K, true>(bx, by, w, Ix, Iy, a, sh0, on_lo, on_hi, hk, hidden_dperp)) alive = false;
        }
        return __ballot(alive);
    }
#pragma unroll
    for (int d = 0; d < K; ++d) {
        const int wd = (int)((code >> (12 * d)) & 0xfffull);
        const float4 r0 = tab[2 * wd], r1 = tab[2 * wd + 1], fc = tab[2 * a.N + wd];
        w[d] = make_wallc(r0, r1, fc, wd);
        image_of(r0, ix, iy, Ix[d], Iy[d]);
        ix = Ix[d];
        iy = Iy[d];
    }
    if (alive) {
        const unsigned long long sh0 = a.shadow ? cmem(a.shadow)[w[0].idx] : 0ull;
        const unsigned long long hk = (hidden_row && a.shadow) ? hidden_row[w[K - 1].idx] : 0ull;
        if (cull_candidate<K>(bx, by, w, Ix, Iy, a, sh0, on_lo, on_hi, hk, hidden_dperp)) alive = false;
    }
    return __ballot(alive);
}

// Order K >= 2 from the region's candidate list: 64 entries at a time, lanes = candidates: the full tile-culling test
// against the PATCH; the survivors are then evaluated exactly in list order (= the reference's order).  parts > 1: only
// the survivors whose rank lies in part `part` of `parts` (the list is culled twice then: once to count).
template <int K, int MODE, bool STATS, bool GRAD, bool LIST>
__device__ __forceinline__ void sweep_order_listed(const SweepArgs& a, const float4* tab, const float (&bx)[4],
                                                   const float (&by)[4], float rxx, float rxy, bool lane_bad, float& acc,
                                                   WaveStats& st, GradCtx* g, long region, int part, int parts, ListSink* sink) {
    static_assert(K >= 2, "lists exist for orders >= 2");
    const int lane = threadIdx.x & 63;
    const RegionLists* rl = a.rl;
    const auto* rlc = cmem(rl);
    const int n = cmem(rlc->leaf.cnt[K])[region];  // >= 0: patches of a region with a list that is not listed go to the enumerating kernel
    const int chunk0 = rlc->leaf.chunk0[K] + (int)region;
    const auto* pool = cmem(rlc->lp.pool);
    const auto* next = cmem(rlc->lp.next);
    // the region's last-segment masks once more, now with the bins this PATCH can reach (hidden_region_kernel)
    const unsigned long long* hidden_row = (D2D_HIDDEN_PATCH && rlc->leaf.hidden) ? rlc->leaf.hidden + (size_t)region * a.N : nullptr;
    const float hidden_dperp = hidden_row ? rlc->leaf.hidden_dperp : 0.0f;
    int r_lo = 0, r_hi = 0x7fffffff;
    if (parts > 1) {
        int T = 0, chunk = chunk0;
        for (int off = 0; off < n; off += 64) {
            const bool have = off + lane < n;
            const unsigned long long code = pool[(size_t)chunk * RL_CHUNK + (off & (RL_CHUNK - 1)) + (have ? lane : 0)];
            if ((off & (RL_CHUNK - 1)) == RL_CHUNK - 64 && off + 64 < n) chunk = next[chunk];
            float Ix[K], Iy[K];
            T += __builtin_popcountll(cull_batch<K, GRAD>(a, tab, bx, by, code, have, Ix, Iy, a.on_lo, a.on_hi, hidden_row, hidden_dperp));
            D2D_WORK(5 * K);
        }
        if (K == 2 && a.min_order <= 1) {
            // part 0 has swept orders 0 and 1 on its way here -- at cfg2 as much work as a quarter of a dear patch's order 2
            // (scripts/timeline.py: 36 us of part 0's 71, the other parts 33) -- and takes 1 / (4 parts) of the ranks only
            const int T0 = T / (4 * parts);
            r_lo = part == 0 ? 0 : T0 + (int)(((long)(T - T0) * (part - 1)) / (parts - 1));
            r_hi = part == 0 ? T0 : T0 + (int)(((long)(T - T0) * part) / (parts - 1));
        } else {
            r_lo = (int)(((long)T * part) / parts);
            r_hi = (int)(((long)T * (part + 1)) / parts);
        }
    }
    int ord = 0, chunk = chunk0;
    for (int off = 0; off < n && ord < r_hi; off += 64) {
        const bool have = off + lane < n;
        const unsigned long long code = pool[(size_t)chunk * RL_CHUNK + (off & (RL_CHUNK - 1)) + (have ? lane : 0)];
        if ((off & (RL_CHUNK - 1)) == RL_CHUNK - 64 && off + 64 < n) chunk = next[chunk];
        float Ix[K], Iy[K];
        float on_lo = a.on_lo, on_hi = a.on_hi;
        if (MODE == MODE_SIG && !LIST && !GRAD && parts == 1 && a.sig_mono) {
            // Sigmoid validity is never exactly zero near a wall, but every contribution is >= 0 and the sum only grows: a
            // candidate whose contribution is certainly below a quarter ulp of the sum it would be added to leaves that
            // sum unchanged (round to nearest), in every lane -- skipping it is exact.  valid <= sigmoid(alpha (s - 0))
            // <= exp(alpha s) for s < 0 (and the same beyond 1), fun <= f_max: the window in which on_objects matters
            // shrinks from 89 / alpha to |zc| / alpha, zc from the smallest sum in the wave (sig_zc).
            float zc = sig_zc_of(a.sig_l2f, acc);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) zc = fminf(zc, __shfl_xor(zc, off, 64));
            if (zc > -89.0f) {
                const float wdn = zc / a.alpha * 1.00001f - 1e-30f;  // (negative) s < wdn or s > 1 - wdn: negligible
                on_lo = fmaxf(on_lo, wdn);
                on_hi = fminf(on_hi, 1.0f - wdn);
            }
        }
        unsigned long long mask = cull_batch<K, GRAD>(a, tab, bx, by, code, have, Ix, Iy, on_lo, on_hi, hidden_row, hidden_dperp);
        if (STATS) stat_add<9>(st, K);
        D2D_WORK(5 * K);
        int budget = 64;
        if (parts > 1) {
            const int nb = __builtin_popcountll(mask);
            int skip = r_lo - ord;  // survivors of this batch that belong to earlier parts
            budget = r_hi - (ord > r_lo ? ord : r_lo);
            ord += nb;
            for (; skip > 0 && mask; --skip) mask &= mask - 1;
        }
        const unsigned long long te0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
        while (mask && budget > 0) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            --budget;
            const unsigned lo32 = (unsigned)__builtin_amdgcn_readlane((int)(code & 0xffffffffull), b);
            const unsigned hi32 = (K >= 3) ? (unsigned)__builtin_amdgcn_readlane((int)(code >> 32), b) : 0u;
            const unsigned long long cu = ((unsigned long long)hi32 << 32) | lo32;
            int ce[D2D_MAX_ORDER] = {-1, -1, -1, -1};
            float ex[D2D_MAX_ORDER], ey[D2D_MAX_ORDER];
#pragma unroll
            for (int d = 0; d < K; ++d) {
                ce[d] = (int)((cu >> (12 * d)) & 0xfffull);
                // the owning lane's image chain: the same operations on the same operands as the wave-uniform chain
                ex[d] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Ix[d]), b));
                ey[d] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Iy[d]), b));
            }
            if (LIST) {
                float t = 0.0f;
                eval_candidate<K, MODE, STATS, GRAD, false, false>(a, ce, ex, ey, a.txx, a.txy, rxx, rxy, lane_bad, t, st, g);
                if (!(t == 0.0f)) sink->push(t);  // non-zero or NaN
            } else {
                eval_candidate<K, MODE, STATS, GRAD, false, false>(a, ce, ex, ey, a.txx, a.txy, rxx, rxy, lane_bad, acc, st, g);
            }
        }
        if (STATS) stat_add<14>(st, __builtin_amdgcn_s_memtime() - te0);
    }
}

// Part `part` of `parts` of the order-K candidates: LISTED, the survivors of the region's list by rank; else the first
// walls of first_wall_range(part, parts), by enumeration.
template <int K, int MODE, bool STATS, bool GRAD, bool LIST, bool LISTED>
__device__ __forceinline__ void sweep_order_any(const SweepArgs& a, const float4* tab, const float (&bx)[4], const float (&by)[4],
                                                float rxx, float rxy, bool lane_bad, float& acc, WaveStats& st, GradCtx* g,
                                                long region, int part, int parts, ListSink* sink) {
    if constexpr (K >= 2 && LISTED) {
        sweep_order_listed<K, MODE, STATS, GRAD, LIST>(a, tab, bx, by, rxx, rxy, lane_bad, acc, st, g, region, part, parts, sink);
    } else {
        int lo = 0, hi = 0x7fffffff;
        if (parts > 1) first_wall_range(a, part, parts, lo, hi);
        sweep_order_culled<K, MODE, STATS, GRAD, LIST>(a, tab, bx, by, rxx, rxy, lane_bad, acc, st, g, lo, hi, sink);
    }
}

// The region (R x R patches) of patch (tcol, trow)
__device__ __forceinline__ long region_of(const SweepArgs& a, int tcol, int trow) {
    const int R = cmem(a.rl)->leaf.R;
    return (long)(trow / R) * cmem(a.rl)->leaf.regions_x + (tcol / R);
}

#ifndef D2D_HEAVY_PARTS
#define D2D_HEAVY_PARTS 4
#endif
// Hand-over between the parts of a cut patch without cache maintenance (fwd_patch): relies on gfx9 encodings and on the
// memory system of gfx942 / gfx950; every other target (and -DD2D_FENCE_FREE_HANDOVER=0) uses release / acquire.
#ifndef D2D_FENCE_FREE_HANDOVER
#if defined(__gfx942__) || defined(__gfx950__)
#define D2D_FENCE_FREE_HANDOVER 1
#else
#define D2D_FENCE_FREE_HANDOVER 0
#endif
#endif
// NaN scan of the value+grad sweeps (d2d_nanscan.hpp): waves per workgroup = patches per region (NAN_R x NAN_R), entries of the
// region's list in LDS, batches tested per round (so that the list cannot overflow)
#ifndef D2D_NAN_W
#define D2D_NAN_W 16  // A/B: 8 = regions of 4 x 2 patches (twice the registers per lane, two workgroups per CU)
#endif
constexpr int NAN_W = D2D_NAN_W;
constexpr int NAN_R = 4;            // patches per region along x; NAN_W / NAN_R along y
constexpr int NAN_RY = NAN_W / NAN_R;
#ifndef D2D_NAN_LCAP
#define D2D_NAN_LCAP 2048  // A/B: entries of a region's list per round (a round = LCAP / 64 batches between two barriers)
#endif
constexpr int NAN_LCAP = D2D_NAN_LCAP;
constexpr int NAN_RB = NAN_LCAP / 64;
constexpr int NAN_WQCAP = 2048;     // (patch, candidate) items of a region waiting for their probe (d2d_nanscan.hpp; beyond it a wave probes its own)
constexpr int HEAVY_PARTS = D2D_HEAVY_PARTS;  // the dearest patches of a launch are cut into this many parts (power_fwd_kernel)
constexpr int TILE_W = 8;  // a wave covers an 8 x 8 patch of RX cells: neighbouring cells share skips
constexpr int TILE_H = 8;

// MAXK = highest order compiled into this instantiation (the host picks the smallest that covers max_order:
// register allocation is the maximum over all compiled paths, and orders 3 / 4 need many more VGPRs).
#ifndef D2D_FWD_WAVES
#define D2D_FWD_WAVES 1  // minimum waves per SIMD asked of the register allocator (1 = unconstrained)
#endif
// GRADK: also run the hand-derived adjoint of every surviving candidate (value + gradient in one sweep).  Culled
// candidates contribute exactly 0 to the value and to every adjoint; what culling cannot reproduce are the
// reference's autodiff NaN artefacts of candidates it never evaluates (see DESIGN.md "NaN parity"): those are
// covered by the exhaustive power_vg_kernel (d2d_params.strict_nan).
// LISTED: the orders >= 2 come from the region candidate lists (a.rl); a patch that cannot use them is queued for the
// enumerating build of the same kernel (LISTED = false), which is launched right behind with a.fb_n set.
template <int MODE, bool STATS, int MAXK, bool GRADK, bool LISTED, bool SPARE = false>
__device__ __forceinline__ void fwd_patch(const SweepArgs& a, float4* tab, float* wl, const long b0_in, const bool from_queue) {
    const int lane = threadIdx.x & 63;
    const int tiles_x = (a.n + TILE_W - 1) / TILE_W;
    const bool scene = GRADK && a.partial != nullptr;
    float tbx_sum = 0.0f, tby_sum = 0.0f;  // scene VJP w.r.t. the fixed end point (wave sum)
    WaveStats st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st.c[i] = 0;
    st.shadow = -1;
    st.work = 0;
    // One 8 x 8 patch per wave, one wave per workgroup: measured equal or better than persistent waves walking several
    // patches (static striding or an atomic work queue) at 1024^2 .. 4096^2, and it keeps the VGPR count lower.
    // ... except the dearest patches of a launch with a work history (max_order == 2): a patch can take 4x the mean and a
    // 1024^2 launch is only ~4 mean patch latencies long, so they are its critical path even though they start first
    // (scripts/timeline.py: the last 22 % of the launch wait for fewer than 60 of 16 384 patches).  Each of them is cut into
    // four quarters of first walls, swept by four single-wave workgroups that leave their non-zero contributions as
    // ordered lists in global memory; the quarter that finishes last adds them up in candidate order (bit for bit the
    // reference's sum) and writes the cell.  A list cannot overflow: it holds as many entries as the quarter has candidates.
    const long n_items = (long)tiles_x * ((a.m + TILE_H - 1) / TILE_H) + (long)(HEAVY_PARTS - 1) * a.n_heavy;
    const bool exists = !SPARE || b0_in < n_items;  // (the last workgroup of a launch of several waves per workgroup may have spare waves)
    const long b0 = exists ? b0_in : 0;
    const bool quarter = !STATS && !GRADK && MAXK == 2 && !from_queue && (b0 < (long)HEAVY_PARTS * a.n_heavy);  // wave-uniform
    const long tile0 = quarter ? (b0 / HEAVY_PARTS) : (b0 - (long)(HEAVY_PARTS - 1) * a.n_heavy);
    const int part = quarter ? (int)(b0 % HEAVY_PARTS) : 0;
    const long tile = from_queue ? b0 : (a.sched ? (long)a.sched[tile0] : tile0);
#ifdef D2D_AB_TIMELINE
    const unsigned long long t_line0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz, common to all XCDs
    if (a.tl_ring && lane == 0 && !from_queue && b0_in == 0) a.tl_ring[2 * (a.tl_seq & 255)] = t_line0;  // (workgroup 0 starts first, or nearly so)
#endif
    const unsigned long long t_start = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
    const int tcol = (int)(tile % tiles_x), trow = (int)(tile / tiles_x);
    const long region = LISTED ? region_of(a, tcol, trow) : 0;
    const int col = tcol * TILE_W + (lane & (TILE_W - 1));
    const int row = trow * TILE_H + (lane / TILE_W);
    const bool in_range = (col < a.n) && (row < a.m);
    const int ccol = col < a.n ? col : a.n - 1;
    const int crow = row < a.m ? row : a.m - 1;
    const long idx = (long)crow * a.n + ccol;
    const float rxx = a.X[idx], rxy = a.Y[idx];
    if (scene) {
        for (int i = lane; i < 4 * a.N; i += 64) wl[i] = 0.0f;
        __syncthreads();
    }
    if (!exists) return;
    const bool lane_bad = !(fabsf(rxx) < 1e18f) || !(fabsf(rxy) < 1e18f) || !(fabsf(a.txx) < 1e18f) ||
                          !(fabsf(a.txy) < 1e18f);
    if (LISTED) {
        // not this kernel's patch: leave it (once) to the enumerating kernel
        if (cmem(cmem(a.rl)->flag)[region] != 0 || wave_any(lane_bad)) {
            if (part == 0 && lane == 0) a.fb_list[atomicAdd(a.fb_n, 1)] = (int)tile;
            return;
        }
    }
    float acc = 0.0f;  // scene.py:1893
    GradCtx g;
    g.grx = g.gry = g.tbx = g.tby = 0.0f;
    g.cot = in_range ? (a.cot ? a.cot[idx] : 1.0f) : 0.0f;
    g.wl = wl;
    g.scene = scene;
    g.ci = 0;
    g.cell = 0;
    // bounding box of the wave's cells (NaN / inf coordinates make every comparison fail: nothing is culled)
    float x0 = rxx, x1 = rxx, y0 = rxy, y1 = rxy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, off, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, off, 64));
        y0 = fminf(y0, __shfl_xor(y0, off, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
    }
    const bool box_ok = !wave_any(lane_bad);
    const float qn = __builtin_nanf("");
    const float bx[4] = {box_ok ? x0 : qn, x1, x1, x0};
    const float by[4] = {y0, y0, y1, y1};
    unsigned long long tq0 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
    if (STATS) stat_add<10>(st, tq0 - t_start);  // prologue of the patch
#ifdef D2D_AB_TIMELINE
    const unsigned t_lineB = (unsigned)(__builtin_amdgcn_s_memrealtime() & 0xffffull);
    unsigned t_lineC = t_lineB;
#endif
    if (quarter) {
        const long hq = tile0 * HEAVY_PARTS + part;
        ListSink sink;
        sink.col = a.heavy_list + hq * (long)a.heavy_cap * 64 + lane;
        sink.cnt = 0;
        sink.over = false;
        sink.cap = a.heavy_cap;
        sink.through = true;
        float dummy = 0.0f;
        if (part == 0) {
            if (a.min_order <= 0 && a.max_order >= 0) {
                float t = 0.0f;
                sweep_order<0, MODE, false, false>(a, a.txx, a.txy, rxx, rxy, lane_bad, t, st, nullptr);
                if (!(t == 0.0f)) sink.push(t);
            }
            if (a.min_order <= 1 && a.max_order >= 1) {
                const unsigned long long* hid = LISTED ? cmem(a.rl)->leaf.hidden : nullptr;
                sweep_order_culled<1, MODE, false, false, true>(a, tab, bx, by, rxx, rxy, lane_bad, dummy, st, nullptr, 0, 0x7fffffff, &sink, nullptr,
                                                                hid ? hid + (size_t)region * a.N : nullptr, hid ? cmem(a.rl)->leaf.hidden_dperp : 0.0f);
            }
        }
#ifdef D2D_AB_TIMELINE
        t_lineC = (unsigned)(__builtin_amdgcn_s_memrealtime() & 0xffffull);
#endif
        if (a.min_order <= 2 && a.max_order >= 2)
            sweep_order_any<2, MODE, false, false, true, LISTED>(a, tab, bx, by, rxx, rxy, lane_bad, dummy, st, nullptr, region, part, HEAVY_PARTS, &sink);
        // Hand-over to the part that finishes last, without cache maintenance: a release / acquire fence pair at agent
        // scope is buffer_wbl2 + buffer_inv on gfx950, and the invalidate drops every line of the XCD's L2, the tables of
        // all the other waves included (with 1024 cut patches every wave of the launch ran 3 x slower:
        // scripts/timeline.py).  Instead everything another workgroup will read is stored past the L2 (agent-scope
        // atomic stores: sc1 write-through), the wave waits until those stores are acknowledged (vmcnt(0)) before it
        // draws its number, and the finishing part reads with agent-scope loads (sc1), issued behind the branch on the
        // number it drew.
        int* const heavy_cnt = D2D_LATE_ARG(int*, heavy_cnt);
        int* const heavy_done = D2D_LATE_ARG(int*, heavy_done);
        const int n_heavy_l = D2D_LATE_ARG(int, n_heavy), heavy_cap_l = D2D_LATE_ARG(int, heavy_cap);
        __hip_atomic_store(&heavy_cnt[hq * 64 + lane], sink.over ? -1 : sink.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) __hip_atomic_store(&heavy_cnt[(long)n_heavy_l * HEAVY_PARTS * 64 + hq], (int)st.work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int old = 0;
#if D2D_FENCE_FREE_HANDOVER
        // gfx942 / gfx950 only: the s_waitcnt immediate below is the gfx9 encoding, stores count in vmcnt there, and an
        // sc1 store is acknowledged once it is visible at agent scope.  Any other target takes the portable branch.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (compiler ordering)
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): every store of the wave has been acknowledged
        if (lane == 0) old = __hip_atomic_fetch_add(&heavy_done[tile0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
        if (lane == 0) old = __hip_atomic_fetch_add(&heavy_done[tile0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
#ifdef D2D_AB_TIMELINE
        if (a.grad && lane == 0) {
            reinterpret_cast<unsigned*>(a.grad)[2 * b0] = (unsigned)(((t_line0 & 0xffffull) << 16) | (__builtin_amdgcn_s_memrealtime() & 0xffffull));
            reinterpret_cast<unsigned*>(a.grad)[2 * b0 + 1] = (t_lineB << 16) | t_lineC;
        }
#endif
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != HEAVY_PARTS - 1) return;  // another part will finish the patch
#if !D2D_FENCE_FREE_HANDOVER
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the finishing part only
#endif
        unsigned work = 0;
        for (int q = 0; q < HEAVY_PARTS; ++q) {
            const long hq2 = tile0 * HEAVY_PARTS + q;
            int n = __hip_atomic_load(&heavy_cnt[hq2 * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool bad = n < 0 || n > heavy_cap_l;  // cannot happen (heavy_cap covers every candidate); never silently wrong
            n = bad ? 0 : n;
            int nmax = n;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
            const int* col = reinterpret_cast<const int*>(D2D_LATE_ARG(float*, heavy_list)) + hq2 * (long)heavy_cap_l * 64 + lane;
            // eight loads in flight, then their additions in candidate order (scene.py:1909): the finishing part is the critical
            // path of a cut patch, and one load at a time is one L2 round trip per entry (an entry that does not exist loads
            // entry 0 instead and adds +0.0, which never changes acc: acc is never -0.0)
            for (int i0 = 0; i0 < nmax; i0 += 8) {
                int v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    v[k] = __hip_atomic_load(&col[(i0 + k < n ? i0 + k : 0) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = acc + (i0 + k < n ? __int_as_float(v[k]) : 0.0f);
            }
            if (bad) acc = __builtin_nanf("");
            work += (unsigned)__hip_atomic_load(&heavy_cnt[(long)n_heavy_l * HEAVY_PARTS * 64 + hq2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        st.work = work;
        if (lane == 0) __hip_atomic_store(&heavy_done[tile0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    } else {
    if (a.min_order <= 0 && a.max_order >= 0) sweep_order<0, MODE, STATS, GRADK>(a, a.txx, a.txy, rxx, rxy, lane_bad, acc, st, &g);
    unsigned long long tq1 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
    if (STATS) stat_add<11>(st, tq1 - tq0);      // order 0
    if (a.min_order <= 1 && a.max_order >= 1) {
        const unsigned long long* hid = LISTED ? cmem(a.rl)->leaf.hidden : nullptr;
        sweep_order_culled<1, MODE, STATS, GRADK>(a, tab, bx, by, rxx, rxy, lane_bad, acc, st, &g, 0, 0x7fffffff, nullptr, nullptr,
                                                  hid ? hid + (size_t)region * a.N : nullptr, hid ? cmem(a.rl)->leaf.hidden_dperp : 0.0f);
    }
    unsigned long long tq2 = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
    if (STATS) stat_add<12>(st, tq2 - tq1);      // order 1
#ifdef D2D_AB_TIMELINE
    t_lineC = (unsigned)(__builtin_amdgcn_s_memrealtime() & 0xffffull);
#endif
    if (a.min_order <= 2 && a.max_order >= 2) sweep_order_any<2, MODE, STATS, GRADK, false, LISTED>(a, tab, bx, by, rxx, rxy, lane_bad, acc, st, &g, region, 0, 1, nullptr);
    if (STATS) stat_add<13>(st, __builtin_amdgcn_s_memtime() - tq2);  // order 2
    if (MAXK >= 3 && a.min_order <= 3 && a.max_order >= 3) sweep_order_any<3, MODE, STATS, GRADK, false, LISTED>(a, tab, bx, by, rxx, rxy, lane_bad, acc, st, &g, region, 0, 1, nullptr);
    if (MAXK >= 4 && a.min_order <= 4 && a.max_order >= 4) sweep_order_any<4, MODE, STATS, GRADK, false, LISTED>(a, tab, bx, by, rxx, rxy, lane_bad, acc, st, &g, region, 0, 1, nullptr);
    }
    {
        float* const out = D2D_LATE_ARG(float*, out);
        const int out_mode = D2D_LATE_ARG(int, out_mode);
        if (in_range) {
            if (out_mode == D2D_OUT_ADD) {
                out[idx] = out[idx] + acc;
                if (GRADK) {
                    a.grad[2 * idx] = a.grad[2 * idx] + g.grx;
                    a.grad[2 * idx + 1] = a.grad[2 * idx + 1] + g.gry;
                }
            } else {
                out[idx] = acc;
                if (GRADK) {
                    a.grad[2 * idx] = g.grx;
                    a.grad[2 * idx + 1] = g.gry;
                }
            }
        }
    }
    if (scene) {
        tbx_sum += wave_sum(g.tbx);
        tby_sum += wave_sum(g.tby);
    }
    if (STATS && lane == 0 && a.wave_cycles) a.wave_cycles[tile] = __builtin_amdgcn_s_memtime() - t_start;
    if (!STATS) {
        unsigned* const cost_out = D2D_LATE_ARG(unsigned*, cost_out);
        if (cost_out && lane == 0) cost_out[tile] = st.work;
    }
#ifdef D2D_AB_TIMELINE  // diagnostic build (scripts/timeline.py): start / end stamps of every workgroup, beside the work history
    if (!STATS && !GRADK && a.grad && lane == 0)
    if (!quarter) {
        reinterpret_cast<unsigned*>(a.grad)[2 * b0] = (unsigned)(((t_line0 & 0xffffull) << 16) | (__builtin_amdgcn_s_memrealtime() & 0xffffull));
        reinterpret_cast<unsigned*>(a.grad)[2 * b0 + 1] = (t_lineB << 16) | t_lineC;
    }
#endif
    if (scene) {
        __syncthreads();
        // one row per patch: the row order of the fp64 reduction must not depend on the schedule
        float* dst = a.partial + tile * (4 * a.N + 2);
        for (int i = lane; i < 4 * a.N; i += 64) dst[i] = wl[i];
        if (lane == 0) {
            dst[4 * a.N] = tbx_sum;
            dst[4 * a.N + 1] = tby_sum;
        }
    }
    if (STATS && a.stats) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            unsigned v = st.c[i];  // (wave-uniform counts held per lane: a lane that sat out a divergent stretch counted less)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off, 64));
            if (lane == 0) atomicAdd(&a.stats[i], (unsigned long long)v);
        }
    }
}

// WPB: waves (= patches) per workgroup.  1: the register allocation that serves small scenes best; 4: the waves share
// the staged tables, which is what keeps big scenes (a 13 KB table at 200 walls) from running out of LDS at 3 waves per SIMD.
#ifdef D2D_NUM_SGPR  // A/B: cap the scalar registers (96 -> 7 waves per SIMD, 80 -> 8 by MI355X_MICROARCH.md's residency formula)
#ifdef D2D_NUM_VGPR
#define D2D_SGPR_ATTR __attribute__((amdgpu_num_sgpr(D2D_NUM_SGPR), amdgpu_num_vgpr(D2D_NUM_VGPR)))
#else
#define D2D_SGPR_ATTR __attribute__((amdgpu_num_sgpr(D2D_NUM_SGPR)))
#endif
#else
#define D2D_SGPR_ATTR
#endif
// Minimum waves per SIMD asked of the register allocator, A/B only.  The order-2 sweep from the region lists -- the benchmark's
// kernel -- fits 7 waves (71 VGPRs, no scratch, -DD2D_FWD_WAVES_L2=7) now that its cold arguments are read late (late_kernarg:
// 82 -> 58 parked scalars); unconstrained the allocator stops at 73, one register past the 72 that 7 waves allow.  Measured on
// the MI355X (round 3, scripts/ab_build.sh "w1:-DD2D_FWD_WAVES_L2=1" "w7:-DD2D_FWD_WAVES_L2=7", two runs each): sweep kernel
// 0.098 ms at 6 waves, 0.099 - 0.100 ms at 7 (90 parked scalars), step 0.110 - 0.113 vs 0.119 ms: residency is not what
// bounds the kernel (mean 4.9 waves per SIMD over the launch: it is never full for long).  Default: unconstrained.
#ifndef D2D_FWD_WAVES_L2
#define D2D_FWD_WAVES_L2 1
#endif
constexpr int fwd_min_waves(int mode, bool stats, int maxk, bool gradk, bool listed, int wpb) {
    if (wpb != 1) return 1;
    if (listed && maxk == 2 && !gradk && !stats && mode == MODE_HARD) return D2D_FWD_WAVES_L2;
    return D2D_FWD_WAVES;
}
template <int MODE, bool STATS, int MAXK, bool GRADK = false, bool LISTED = false, int WPB = 1>
__global__ void __launch_bounds__(64 * WPB, fwd_min_waves(MODE, STATS, MAXK, GRADK, LISTED, WPB)) D2D_SGPR_ATTR power_fwd_kernel(SweepArgs a) {
    const int lane = threadIdx.x & 63;
    // LDS copy of the per-wall tables for the lanes-as-candidates phase (lane-varying wall index), staged once per wave
    extern __shared__ float4 tab[];  // [2N] refl, [N] flt, then (GRADK) [N] float4 = the wave's scene-VJP partial sums
    float* wl = reinterpret_cast<float*>(tab + 3 * a.N);
    if (!LISTED && a.fb_n != nullptr) {
        // the patches the LISTED launch in front of this one left behind (usually none)
        const int n = *a.fb_n;
        if (n > 0) {
            for (int i = lane; i < 2 * a.N; i += 64) tab[i] = ldc4(a.refl, i);
            for (int i = lane; i < a.N; i += 64) tab[2 * a.N + i] = ldc4(a.flt, i);
            __syncthreads();
        }
        for (int i = blockIdx.x; i < n; i += gridDim.x) {
            fwd_patch<MODE, STATS, MAXK, GRADK, false>(a, tab, wl, (long)a.fb_list[i], true);
            __syncthreads();
        }
        return;
    }
    // one patch per wave; the waves of a workgroup (WPB) stage the tables together
    for (int i = threadIdx.x; i < 2 * a.N; i += 64 * WPB) tab[i] = ldc4(a.refl, i);
    for (int i = threadIdx.x; i < a.N; i += 64 * WPB) tab[2 * a.N + i] = ldc4(a.flt, i);
    __syncthreads();
    fwd_patch<MODE, STATS, MAXK, GRADK, LISTED, (WPB > 1)>(a, tab, wl, WPB == 1 ? (long)blockIdx.x : (long)blockIdx.x * WPB + (threadIdx.x >> 6), false);
}

// Forward sweep with every 8 x 8 patch shared by W waves (one workgroup).  Patches differ a lot in cost and the dearest
// ones sit on the critical path of a launch that only holds a few patches per SIMD (1024^2: 16), so the orders K >= 2
// are cut into W contiguous ranges of first-wall positions (balanced over the first walls that the shadow masks do not
// kill outright).  Wave 0 adds its range to acc directly; waves 1 .. W-1 record their non-zero contributions as ordered
// per-lane lists in LDS, which wave 0 then adds in range order: the same left-to-right fp32 sum as the reference's
// (scene.py:1893-1916), bit for bit.  A list that overflows is discarded and wave 0 redoes that range itself.
template <int K, int MODE, bool STATS, int W, bool LISTED>
__device__ __forceinline__ void split_order(const SweepArgs& a, const float4* tab, float* lists, int* meta,
                                            const float (&bx)[4], const float (&by)[4], float rxx, float rxy,
                                            bool lane_bad, float& acc, WaveStats& st, long region) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int* cnts = meta;                  // [(W - 1)][64]
    int* flags = meta + (W - 1) * 64;  // [(W - 1)] overflow
    // wave w takes part w of W of the first walls (first_wall_range / the region lists' slices)
    if (wv != 0) {
        ListSink sink;
        sink.col = lists + (size_t)(wv - 1) * SPLIT_LIST * 64 + lane;
        sink.cnt = 0;
        sink.over = false;
        float dummy = 0.0f;
        sweep_order_any<K, MODE, STATS, false, true, LISTED>(a, tab, bx, by, rxx, rxy, lane_bad, dummy, st, nullptr, region, wv, W, &sink);
        cnts[(wv - 1) * 64 + lane] = sink.cnt;
        const bool over = wave_any(sink.over);
        if (lane == 0) flags[wv - 1] = over ? 1 : 0;
    }
#pragma unroll 1
    for (int w = 0; w < W; ++w) {
        if (w == 1) __syncthreads();  // uniform: every wave runs this loop
        if (wv == 0) {
            if (w == 0 || flags[w - 1]) {
                sweep_order_any<K, MODE, STATS, false, false, LISTED>(a, tab, bx, by, rxx, rxy, lane_bad, acc, st, nullptr, region, w, W, nullptr);
            } else {
                const int n = cnts[(w - 1) * 64 + lane];
                int nmax = n;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) nmax = max(nmax, __shfl_xor(nmax, off, 64));
                const float* col = lists + (size_t)(w - 1) * SPLIT_LIST * 64 + lane;
                for (int i = 0; i < nmax; ++i)
                    if (i < n) acc = acc + col[i * 64];  // scene.py:1909, in candidate order
            }
        }
    }
    __syncthreads();  // lists and meta are reused by the next order
}

template <int MODE, bool STATS, int MAXK, int W, bool LISTED>
__device__ __forceinline__ void split_patch(const SweepArgs& a, const float4* tab, float* lists, int* meta, const long slot, const bool from_queue) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_x = (a.n + TILE_W - 1) / TILE_W;
    WaveStats st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st.c[i] = 0;
    st.shadow = -1;
    st.work = 0;
    const unsigned long long t_start = STATS ? __builtin_amdgcn_s_memtime() : 0ull;
    const int tile = from_queue ? (int)slot : (a.sched ? a.sched[slot] : (int)slot);
    const int tcol = tile % tiles_x, trow = tile / tiles_x;
    const long region = LISTED ? region_of(a, tcol, trow) : 0;
    const int col = tcol * TILE_W + (lane & (TILE_W - 1));
    const int row = trow * TILE_H + (lane / TILE_W);
    const bool in_range = (col < a.n) && (row < a.m);
    const int ccol = col < a.n ? col : a.n - 1;
    const int crow = row < a.m ? row : a.m - 1;
    const long idx = (long)crow * a.n + ccol;
    const float rxx = a.X[idx], rxy = a.Y[idx];
    const bool lane_bad = !(fabsf(rxx) < 1e18f) || !(fabsf(rxy) < 1e18f) || !(fabsf(a.txx) < 1e18f) ||
                          !(fabsf(a.txy) < 1e18f);
    if (LISTED) {
        // not this kernel's patch (every wave of the workgroup sees the same cells): leave it to the enumerating kernel
        if (cmem(cmem(a.rl)->flag)[region] != 0 || wave_any(lane_bad)) {
            if (threadIdx.x == 0) a.fb_list[atomicAdd(a.fb_n, 1)] = tile;
            return;
        }
    }
    float acc = 0.0f;  // scene.py:1893
    float x0 = rxx, x1 = rxx, y0 = rxy, y1 = rxy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, off, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, off, 64));
        y0 = fminf(y0, __shfl_xor(y0, off, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
    }
    const bool box_ok = !wave_any(lane_bad);
    const float qn = __builtin_nanf("");
    const float bx[4] = {box_ok ? x0 : qn, x1, x1, x0};
    const float by[4] = {y0, y0, y1, y1};
    const bool writer = wv == 0;
    if (writer) {
        if (a.min_order <= 0 && a.max_order >= 0) sweep_order<0, MODE, STATS, false>(a, a.txx, a.txy, rxx, rxy, lane_bad, acc, st, nullptr);
        if (a.min_order <= 1 && a.max_order >= 1) {
            const unsigned long long* hid = LISTED ? cmem(a.rl)->leaf.hidden : nullptr;
            sweep_order_culled<1, MODE, STATS, false>(a, tab, bx, by, rxx, rxy, lane_bad, acc, st, nullptr, 0, 0x7fffffff, nullptr, nullptr,
                                                      hid ? hid + (size_t)region * a.N : nullptr, hid ? cmem(a.rl)->leaf.hidden_dperp : 0.0f);
        }
    }
    if (a.min_order <= 2 && a.max_order >= 2) split_order<2, MODE, STATS, W, LISTED>(a, tab, lists, meta, bx, by, rxx, rxy, lane_bad, acc, st, region);
    if (MAXK >= 3 && a.min_order <= 3 && a.max_order >= 3) split_order<3, MODE, STATS, W, LISTED>(a, tab, lists, meta, bx, by, rxx, rxy, lane_bad, acc, st, region);
    if (MAXK >= 4 && a.min_order <= 4 && a.max_order >= 4) split_order<4, MODE, STATS, W, LISTED>(a, tab, lists, meta, bx, by, rxx, rxy, lane_bad, acc, st, region);
    if (writer && in_range) {
        if (a.out_mode == D2D_OUT_ADD) a.out[idx] = a.out[idx] + acc;
        else a.out[idx] = acc;
    }
    if (STATS && writer && lane == 0 && a.wave_cycles) a.wave_cycles[tile] = __builtin_amdgcn_s_memtime() - t_start;
    if (!STATS && a.cost_out) {  // workgroup-uniform: the work of the patch = the sum over its W waves
        __shared__ unsigned wave_work[W];
        __syncthreads();  // (a workgroup that walks the queue reuses it)
        if (lane == 0) wave_work[wv] = st.work;
        __syncthreads();
        if (wv == 0 && lane == 0) {
            unsigned sum = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) sum += wave_work[i];
            a.cost_out[tile] = sum;
        }
    }
    if (STATS && a.stats) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            unsigned v = st.c[i];  // (wave-uniform counts held per lane: a lane that sat out a divergent stretch counted less)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off, 64));
            if (lane == 0) atomicAdd(&a.stats[i], (unsigned long long)v);
        }
    }
}

template <int MODE, bool STATS, int MAXK, int W, bool LISTED = false>
__global__ void __launch_bounds__(64 * W) power_fwd_split_kernel(SweepArgs a) {
    extern __shared__ float4 tab[];  // [2N] refl, [N] flt, then the contribution lists and their bookkeeping
    for (int i = threadIdx.x; i < 2 * a.N; i += 64 * W) tab[i] = ldc4(a.refl, i);
    for (int i = threadIdx.x; i < a.N; i += 64 * W) tab[2 * a.N + i] = ldc4(a.flt, i);
    float* lists = reinterpret_cast<float*>(tab + 3 * a.N);
    int* meta = reinterpret_cast<int*>(lists + (size_t)(W - 1) * SPLIT_LIST * 64);
    __syncthreads();
    if (!LISTED && a.fb_n != nullptr) {
        const int n = *a.fb_n;  // the patches the LISTED launch in front of this one left behind (usually none)
        for (int i = blockIdx.x; i < n; i += gridDim.x) {
            split_patch<MODE, STATS, MAXK, W, false>(a, tab, lists, meta, (long)a.fb_list[i], true);
            __syncthreads();
        }
        return;
    }
    split_patch<MODE, STATS, MAXK, W, LISTED>(a, tab, lists, meta, (long)blockIdx.x, false);
}

// ---- small launches: every patch shared by W waves, candidate by candidate -----------------------------------------------
// A launch of a few hundred patches cannot fill 1024 SIMDs with one wave -- or four -- per patch, and it is as long as its
// dearest patch: 58 000 wave-instructions at 300 x 300 cells of cfg2's scene against a mean of 8 600 (d2d_debug_get_work), at
// the ~8 cycles per instruction of a wave that has its SIMD almost to itself.  Here the W waves of a workgroup share a patch
// at the finest grain that keeps the reference's sum order:
//   culling     the 64-candidate batches of an order -- the allowed walls (K = 1), the region's candidate list (K >= 2) --
//               are dealt to the waves round-robin; each wave runs the full culling test on its batches and leaves the
//               survivors' masks in LDS: every candidate is culled once per patch;
//   evaluation  in rounds of W x COOP_C survivors: the survivor of rank r0 + i W + w goes to wave w (INTERLEAVED: a visible
//               wall costs 20 x what a candidate that dies at on_objects costs, and neighbours in candidate order cost
//               alike), which stores its contribution -- zero or not -- in slot [w][i] of LDS;
//   summation   wave 0 adds the round's slots in rank order (adding an exact +0.0 never changes acc, which is never -0.0):
//               the reference's left-to-right fp32 sum (scene.py:1893-1916), bit for bit; nothing can overflow.
// Lists longer than COOP_MAXB batches are processed window by window.  LISTED launches only (the region lists exist);
// patches that cannot use their lists go to the queue of the enumerating kernel like everywhere else.
constexpr int COOP_C = 8;
constexpr int COOP_MAXB = 256;

template <int K, int MODE, int W>
__device__ __forceinline__ void coop_order(const SweepArgs& a, const float4* tab, float* slots, unsigned long long* bmask, int* bchunk,
                                           float* accpub, const float (&bx)[4], const float (&by)[4], float rxx, float rxy, bool lane_bad,
                                           float& acc, WaveStats& st, long region) {
    static_assert(K >= 1 && (W & (W - 1)) == 0, "orders >= 1, a power of two of waves");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // Sigmoid validity with fun >= 0: the cells' sums never shrink, so the sums wave 0 held after the last round are a floor
    // of the sums every later contribution is added to -- what is certainly below a quarter ulp of the floor is certainly
    // absorbed (sig_zc_of), in the culling and in the evaluation alike (power_fwd_kernel does this with the sum itself).
    const bool floor_on = (MODE == MODE_SIG) && a.sig_mono;
    if (floor_on) {
        if (wv == 0) accpub[lane] = acc;
        __syncthreads();
    }
    const auto* rlc = cmem(a.rl);
    const int n = (K == 1) ? a.Nc : cmem(rlc->leaf.cnt[K >= 2 ? K : 2])[region];
    const int nb = (n + 63) >> 6;
    const auto* pool = cmem(rlc->lp.pool);
    const auto* next = cmem(rlc->lp.next);
    int chunk = (K == 1) ? 0 : rlc->leaf.chunk0[K >= 2 ? K : 2] + (int)region;  // the chunk of the window's first batch
    for (int b0 = 0; b0 < nb; b0 += COOP_MAXB) {  // (workgroup-uniform)
        const int nbw = (nb - b0 < COOP_MAXB) ? nb - b0 : COOP_MAXB;
        // ---- culling: batch b0 + i of the window to wave i % W
        {
            float on_lo = a.on_lo, on_hi = a.on_hi;
            if (floor_on) {
                float zc = sig_zc_of(a.sig_l2f, accpub[lane]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) zc = fminf(zc, __shfl_xor(zc, o, 64));
                if (zc > -89.0f) {
                    const float wdn = zc / a.alpha * 1.00001f - 1e-30f;  // (negative) s < wdn or s > 1 - wdn: negligible
                    on_lo = fmaxf(on_lo, wdn);
                    on_hi = fminf(on_hi, 1.0f - wdn);
                }
            }
            int ch = chunk;
            for (int i = 0; i < nbw; ++i) {
                const int off = (b0 + i) << 6;
                if ((i & (W - 1)) == wv) {
                    unsigned long long m;
                    if constexpr (K == 1) {
                        const int lp = off + lane;
                        const int wl = cmem(a.cw)[lp < n ? lp : 0];
                        WallC w[1];
                        float Ix[1], Iy[1];
                        const float4 r0 = tab[2 * wl], r1 = tab[2 * wl + 1], fc = tab[2 * a.N + wl];
                        w[0] = make_wallc(r0, r1, fc, wl);
                        image_of(r0, a.txx, a.txy, Ix[0], Iy[0]);
                        const unsigned long long sh0 = a.shadow ? cmem(a.shadow)[wl] : 0ull;
                        const unsigned long long* hid = rlc->leaf.hidden;
                        const unsigned long long hk = (hid && a.shadow) ? hid[(size_t)region * a.N + wl] : 0ull;
                        m = __ballot(lp < n && !cull_candidate<1>(bx, by, w, Ix, Iy, a, sh0, on_lo, on_hi, hk, hid ? rlc->leaf.hidden_dperp : 0.0f));
                    } else {
                        const bool have = off + lane < n;
                        const unsigned long long code = pool[(size_t)ch * RL_CHUNK + (off & (RL_CHUNK - 1)) + (have ? lane : 0)];
                        float Ix[K], Iy[K];
                        const unsigned long long* hid = D2D_HIDDEN_PATCH ? rlc->leaf.hidden : nullptr;
                        m = cull_batch<K, false>(a, tab, bx, by, code, have, Ix, Iy, on_lo, on_hi, hid ? hid + (size_t)region * a.N : nullptr,
                                                 hid ? rlc->leaf.hidden_dperp : 0.0f);
                    }
                    if (lane == 0) {
                        bmask[i] = m;
                        bchunk[i] = ch;
                    }
                    D2D_WORK(5 * K);
                }
                if (K >= 2 && (off & (RL_CHUNK - 1)) == RL_CHUNK - 64 && off + 64 < n) ch = next[ch];
            }
            chunk = ch;  // (every wave walks the whole chain: the next window starts where this one ended)
        }
        __syncthreads();
        int T = 0;
        for (int i = 0; i < nbw; ++i) T += __builtin_popcountll(bmask[i]);
        T = __builtin_amdgcn_readfirstlane(T);
        // ---- evaluation in rounds, summation by wave 0
        for (int r0 = 0; r0 < T; r0 += W * COOP_C) {  // (workgroup-uniform)
            const int R = (T - r0 < W * COOP_C) ? T - r0 : W * COOP_C;
            const float acc_floor = floor_on ? accpub[lane] : -1.0f;
            int ord = 0;
            for (int i = 0; i < nbw && ord < r0 + R; ++i) {
                const unsigned long long mb = bmask[i];
                unsigned long long m = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(mb >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)(mb & 0xffffffffull));
                const int nbits = __builtin_popcountll(m);
                if (ord + nbits <= r0) {
                    ord += nbits;
                    continue;
                }
                const size_t base = (K == 1) ? (size_t)((b0 + i) << 6)
                                             : (size_t)__builtin_amdgcn_readfirstlane(bchunk[i]) * RL_CHUNK + (((b0 + i) << 6) & (RL_CHUNK - 1));
                while (m) {
                    const int bit = __builtin_ctzll(m);
                    m &= m - 1;
                    const int rel = ord - r0;
                    ++ord;
                    if (rel < 0) continue;
                    if (rel >= R) break;
                    if ((rel & (W - 1)) != wv) continue;
                    int ce[D2D_MAX_ORDER] = {-1, -1, -1, -1};
                    float ex[D2D_MAX_ORDER], ey[D2D_MAX_ORDER];
                    if constexpr (K == 1) {
                        ce[0] = cmem(a.cw)[base + bit];
                        image_of(ldc4(a.refl, 2 * ce[0]), a.txx, a.txy, ex[0], ey[0]);
                    } else {
                        const unsigned long long cu = pool[base + bit];  // wave-uniform: a scalar load
#pragma unroll
                        for (int d = 0; d < K; ++d) {
                            ce[d] = (int)((cu >> (12 * d)) & 0xfffull);
                            // the image chain: the same operations on the same operands as everywhere else (geometry.py:1086-1091)
                            image_of(ldc4(a.refl, 2 * ce[d]), d == 0 ? a.txx : ex[d > 0 ? d - 1 : 0], d == 0 ? a.txy : ey[d > 0 ? d - 1 : 0], ex[d], ey[d]);
                        }
                    }
                    float t = 0.0f;
                    eval_candidate<K, MODE, false, false, false, false>(a, ce, ex, ey, a.txx, a.txy, rxx, rxy, lane_bad, t, st, nullptr, acc_floor);
                    slots[((wv * COOP_C) + (rel / W)) * 64 + lane] = t;
                }
            }
            __syncthreads();
            if (wv == 0) {
                for (int rel = 0; rel < R; ++rel) acc = acc + slots[(((rel & (W - 1)) * COOP_C) + (rel / W)) * 64 + lane];  // scene.py:1909, in candidate order
                if (floor_on) accpub[lane] = acc;
            }
            __syncthreads();  // the slots are reused by the next round, the masks by the next window / order
        }
        __syncthreads();
    }
}

template <int MODE, int MAXK, int W>
__global__ void __launch_bounds__(64 * W) power_fwd_coop_kernel(SweepArgs a) {
    extern __shared__ float4 tab[];  // [2N] refl, [N] flt, then the rounds' slots [W][COOP_C][64]
    __shared__ unsigned long long bmask[COOP_MAXB];
    __shared__ int bchunk[COOP_MAXB];
    __shared__ float accpub[64];
    __shared__ unsigned wave_work[W];
    for (int i = threadIdx.x; i < 2 * a.N; i += 64 * W) tab[i] = ldc4(a.refl, i);
    for (int i = threadIdx.x; i < a.N; i += 64 * W) tab[2 * a.N + i] = ldc4(a.flt, i);
    float* slots = reinterpret_cast<float*>(tab + 3 * a.N);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tiles_x = (a.n + TILE_W - 1) / TILE_W;
    WaveStats st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st.c[i] = 0;
    st.shadow = -1;
    st.work = 0;
    const int tile = a.sched ? a.sched[blockIdx.x] : (int)blockIdx.x;
    const int tcol = tile % tiles_x, trow = tile / tiles_x;
    const long region = region_of(a, tcol, trow);
    const int col = tcol * TILE_W + (lane & (TILE_W - 1));
    const int row = trow * TILE_H + (lane / TILE_W);
    const bool in_range = (col < a.n) && (row < a.m);
    const int ccol = col < a.n ? col : a.n - 1;
    const int crow = row < a.m ? row : a.m - 1;
    const long idx = (long)crow * a.n + ccol;
    const float rxx = a.X[idx], rxy = a.Y[idx];
    const bool lane_bad = !(fabsf(rxx) < 1e18f) || !(fabsf(rxy) < 1e18f) || !(fabsf(a.txx) < 1e18f) || !(fabsf(a.txy) < 1e18f);
    // not this kernel's patch (every wave of the workgroup sees the same cells): leave it to the enumerating kernel
    if (cmem(cmem(a.rl)->flag)[region] != 0 || wave_any(lane_bad)) {
        if (threadIdx.x == 0) a.fb_list[atomicAdd(a.fb_n, 1)] = tile;
        return;
    }
    float acc = 0.0f;  // scene.py:1893
    float x0 = rxx, x1 = rxx, y0 = rxy, y1 = rxy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, off, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, off, 64));
        y0 = fminf(y0, __shfl_xor(y0, off, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
    }
    const float bx[4] = {x0, x1, x1, x0};
    const float by[4] = {y0, y0, y1, y1};
    if (wv == 0 && a.min_order <= 0 && a.max_order >= 0) sweep_order<0, MODE, false, false>(a, a.txx, a.txy, rxx, rxy, lane_bad, acc, st, nullptr);
    if (a.min_order <= 1 && a.max_order >= 1) coop_order<1, MODE, W>(a, tab, slots, bmask, bchunk, accpub, bx, by, rxx, rxy, lane_bad, acc, st, region);
    if (a.min_order <= 2 && a.max_order >= 2) coop_order<2, MODE, W>(a, tab, slots, bmask, bchunk, accpub, bx, by, rxx, rxy, lane_bad, acc, st, region);
    if (MAXK >= 3 && a.min_order <= 3 && a.max_order >= 3) coop_order<3, MODE, W>(a, tab, slots, bmask, bchunk, accpub, bx, by, rxx, rxy, lane_bad, acc, st, region);
    if (MAXK >= 4 && a.min_order <= 4 && a.max_order >= 4) coop_order<4, MODE, W>(a, tab, slots, bmask, bchunk, accpub, bx, by, rxx, rxy, lane_bad, acc, st, region);
    if (wv == 0 && in_range) {
        if (a.out_mode == D2D_OUT_ADD) a.out[idx] = a.out[idx] + acc;
        else a.out[idx] = acc;
    }
    if (a.cost_out) {  // the work of the patch = the sum over its waves
        if (lane == 0) wave_work[wv] = st.work;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned sum = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) sum += wave_work[i];
            a.cost_out[tile] = sum;
        }
    }
}

// Bounding box of the cells of a region, from the level's table (region_box_kernel: {x0, x1, y0, y1}, x0 = NaN when a
// cell is not comfortably finite); false when nothing may be culled for the region.
__device__ __forceinline__ bool region_box(const SweepArgs& a, const float4* __restrict__ box, long region, float (&bx)[4], float (&by)[4]) {
    const float4 b = box[region];
    bx[0] = b.x; bx[1] = b.y; bx[2] = b.y; bx[3] = b.x;
    by[0] = b.z; by[1] = b.z; by[2] = b.w; by[3] = b.w;
    return (b.x <= b.y) && (b.z <= b.w) && (fabsf(a.txx) < 1e18f) && (fabsf(a.txy) < 1e18f);
}

// Region candidate lists by enumeration: one wave per (region, slice of first walls) of level `lv`.  The wave runs the
// prefix odometer and the full tile-culling test of order K against the bounding box of the region's cells and writes
// the survivors, in candidate order, to the slice's list.  What holds for the box holds for everything inside it, so
// whoever only looks at the listed candidates skips nothing but exact zeros.  A list is marked "not listed" (count -1)
// when the pool runs out or when a cell of the region is not comfortably finite.
template <int K, bool GRAD, bool TXG = false>
__global__ void __launch_bounds__(64) region_list_kernel(SweepArgs a, RegionLevel lv, ListPool lp) {
    const int lane = threadIdx.x & 63;
    extern __shared__ float4 tab[];  // [2N] refl, [N] flt, the culling queue
    for (int i = lane; i < 2 * a.N; i += 64) tab[i] = ldc4(a.refl, i);
    for (int i = lane; i < a.N; i += 64) tab[2 * a.N + i] = ldc4(a.flt, i);
    __syncthreads();
    const long rs = blockIdx.x;
    const long region = rs / lv.S;
    const int s = (int)(rs % lv.S);
    float bx[4], by[4];
    const bool ok = region_box(a, lv.box, region, bx, by);
    EmitSink em;
    em.lp = lp;
    em.cur = lv.chunk0[K] + (int)rs;
    em.n = 0;
    em.over = !ok;
    if (ok) {
        int lo, hi;
        first_wall_range(a, s, lv.S, lo, hi);
        WaveStats st;
#pragma unroll
        for (int i = 0; i < 16; ++i) st.c[i] = 0;
        st.shadow = -1;
        st.work = 0;
        float dummy = 0.0f;
        if constexpr (TXG) sweep_order_culled_txg<K, MODE_HARD, GRAD, true>(a, tab, bx, by, 0.0f, 0.0f, false, dummy, st, nullptr, lo, hi, &em);
        else sweep_order_culled<K, MODE_HARD, false, GRAD, false, true>(a, tab, bx, by, 0.0f, 0.0f, false, dummy, st, nullptr, lo, hi, nullptr, &em);
        // instrumented launches: the culling levels evaluated here count as executed work too (5 work units per level)
        if (a.stats && lane == 0) atomicAdd(&a.stats[9], (unsigned long long)(st.work / 5));
    }
    if (lane == 0) lv.cnt[K][rs] = em.over ? -1 : em.n;
}

// Region candidate lists by refinement: one wave per region of level `lv` (one list per region), whose R divides the
// parent level's.  The wave gathers the parent region's lists (all slices, in order) through LDS into full batches and
// keeps what the tile culling cannot drop for its own, smaller box.  `flag`: raised when the list is not listed.
constexpr int RL_GATHER = 512;  // entries of the gather buffer (LDS)
template <int K, bool GRAD, bool TXG = false>
__global__ void __launch_bounds__(64) region_refine_kernel(SweepArgs a, RegionLevel lv, RegionLevel parent, ListPool lp, int* flag) {
    const int lane = threadIdx.x & 63;
    extern __shared__ float4 tab[];  // [2N] refl, [N] flt, then the gather buffer
    for (int i = lane; i < 2 * a.N; i += 64) tab[i] = ldc4(a.refl, i);
    for (int i = lane; i < a.N; i += 64) tab[2 * a.N + i] = ldc4(a.flt, i);
    unsigned long long* buf = reinterpret_cast<unsigned long long*>(tab + 3 * a.N + 1);
    __syncthreads();
    const long region = blockIdx.x;
    const int rx = (int)(region % lv.regions_x), ry = (int)(region / lv.regions_x);
    float bx[4], by[4];
    bool ok = region_box(a, lv.box, region, bx, by);
    const unsigned long long* hidden_row = lv.hidden ? lv.hidden + (size_t)region * a.N : nullptr;
    const int up = parent.R / lv.R;
    const long pslot0 = ((long)(ry / up) * parent.regions_x + (rx / up)) * parent.S;
    const int* pcnt = parent.cnt[K] + pslot0;
    for (int ps = 0; ps < parent.S; ++ps) ok = ok && (pcnt[ps] >= 0);
    EmitSink em;
    em.lp = lp;
    em.cur = lv.chunk0[K] + (int)region;
    em.n = 0;
    em.over = !ok;
    if (ok) {
        int total = 0;
        unsigned levels = 0;
        auto process = [&]() {
            __builtin_amdgcn_wave_barrier();
            for (int off = 0; off < total; off += 64) {
                levels += K;
                const bool have = off + lane < total;
                const unsigned long long code = buf[have ? off + lane : off];
                float Ix[K], Iy[K];
                const unsigned long long mask = cull_batch<K, GRAD, TXG>(a, tab, bx, by, code, have, Ix, Iy, a.on_lo, a.on_hi, hidden_row, lv.hidden_dperp);
                emit_batch(em, code, have && ((mask >> lane) & 1ull), mask);
            }
            total = 0;
            __builtin_amdgcn_wave_barrier();
        };
        // Small lists (every slice of the parent fits one batch and all of them fit the buffer -- small scenes): the slices'
        // lengths come with one vector load, their places in the buffer from a prefix sum, and the entries of four slices
        // are in flight at a time instead of one dependent round trip per slice.
        const int myn = (parent.S <= 64 && lane < parent.S) ? pcnt[lane] : 0;
        int incl = myn;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        const int sum_all = __builtin_amdgcn_readlane(incl, 63);
        const bool small = parent.S <= 64 && !wave_any(myn > 64) && sum_all <= RL_GATHER;
        if (small) {
            const int excl = incl - myn;
            for (int ps0 = 0; ps0 < parent.S; ps0 += 4) {
                unsigned long long v[4];
                int n4[4], o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ps = ps0 + j < parent.S ? ps0 + j : parent.S - 1;
                    n4[j] = ps0 + j < parent.S ? __builtin_amdgcn_readlane(myn, ps) : 0;
                    o4[j] = __builtin_amdgcn_readlane(excl, ps);
                    v[j] = lp.pool[(size_t)(parent.chunk0[K] + (int)(pslot0 + ps)) * RL_CHUNK + (lane < n4[j] ? lane : 0)];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (lane < n4[j]) buf[o4[j] + lane] = v[j];
            }
            total = sum_all;
        }
        for (int ps = 0; ps < (small ? 0 : parent.S); ++ps) {
            const int n = pcnt[ps];
            int chunk = parent.chunk0[K] + (int)(pslot0 + ps);
            for (int off = 0; off < n; off += 64) {
                if (total + 64 > RL_GATHER) process();
                const int m = min(64, n - off);
                if (lane < m) buf[total + lane] = lp.pool[(size_t)chunk * RL_CHUNK + (off & (RL_CHUNK - 1)) + lane];
                total += m;
                if ((off & (RL_CHUNK - 1)) == RL_CHUNK - 64 && off + 64 < n) chunk = lp.next[chunk];
            }
        }
        process();
        if (a.stats && lane == 0) atomicAdd(&a.stats[9], (unsigned long long)levels);
    }
    if (lane == 0) {
        lv.cnt[K][region] = em.over ? -1 : em.n;
        if (em.over) flag[region] = 1;
    }
}

// TX grids (scene.py:1489-1648): the cells are transmitters, (a.txx, a.txy) is the fixed receiver F.  The reference's op
// chain (images of the cell, backward scan from F) is what eval_candidate<TXG = true> runs; the culling only has to bound
// the same geometric interaction points, and a path is its own reverse: the candidate (w_0 .. w_{K-1}) is culled as the
// chain of F through (w_{K-1} .. w_0) towards the patch, with F's shadow masks on w_{K-1} (the segment w_{K-1} -> F).
// Prefix = (w_0 .. w_{K-2}) wave-uniform, lanes = last wall, survivors in ascending order: the reference's order.
// EMIT (K >= 2; region_list_kernel): first-wall positions [p_lo, p_hi) only, the box is a region's, and the survivors are
// appended to `emit` (in candidate order: prefix-major, last walls ascending) instead of being evaluated.
template <int K, int MODE, bool GRAD = false, bool EMIT = false>
__device__ __forceinline__ void sweep_order_culled_txg(const SweepArgs& a, const float4* tab, const float (&bx)[4],
                                                       const float (&by)[4], float cx, float cy, bool lane_bad, float& acc,
                                                       WaveStats& st, GradCtx* g = nullptr, int p_lo = 0, int p_hi = 0x7fffffff,
                                                       EmitSink* emit = nullptr, const unsigned long long* hidden_row = nullptr,
                                                       float hidden_dperp = 0.0f) {
    static_assert(!EMIT || K >= 2, "lists exist for orders >= 2");
    const int lane = threadIdx.x & 63;
    int cand[D2D_MAX_ORDER] = {-1, -1, -1, -1};
    float imgx[D2D_MAX_ORDER], imgy[D2D_MAX_ORDER];  // images of the lane's cell (per lane)
    const int Nc = a.Nc;
    int pos[D2D_MAX_ORDER] = {0, 0, 0, 0};
    const int n_chunks = (Nc + 63) >> 6;
    const int p_end = p_hi < Nc ? p_hi : Nc;
    if (K >= 2) pos[0] = p_lo;
#pragma unroll
    for (int d = 1; d < K - 1; ++d) pos[d] = (pos[d - 1] == 0) ? 1 : 0;
    if (K - 1 > 0 && (Nc < 2 || p_lo >= p_end)) return;
    if (Nc < 1) return;
    while (true) {
#pragma unroll
        for (int d = 0; d < K - 1; ++d) {
            cand[d] = cmem(a.cw)[pos[d]];
            if (!EMIT) image_of(ldc4(a.refl, 2 * cand[d]), d == 0 ? cx : imgx[d > 0 ? d - 1 : 0], d == 0 ? cy : imgy[d > 0 ? d - 1 : 0], imgx[d], imgy[d]);
        }
        const int last_prefix_pos = (K == 1) ? -1 : pos[K >= 2 ? K - 2 : 0];
        bool prefix_dead = false;
        if (K >= 3 && a.pair && a.pair_prefix_ok) {
#pragma unroll
            for (int d = 0; d + 1 < K - 1; ++d) prefix_dead = prefix_dead || (cmem(a.pair)[(size_t)cand[d] * a.N + cand[d + 1]] == ~0ull);
        }
        for (int chunk = 0; chunk < (prefix_dead ? 0 : n_chunks); ++chunk) {
            const int lp = chunk * 64 + lane;
            bool alive = (lp < Nc) && (lp != last_prefix_pos);
            {
                const int wl = cmem(a.cw)[lp < Nc ? lp : 0];
                WallC w[K];
                float Ix[K], Iy[K];
                const float4 r0 = tab[2 * wl], r1 = tab[2 * wl + 1], fc = tab[2 * a.N + wl];
                w[0] = make_wallc(r0, r1, fc, wl);
                image_of(r0, a.txx, a.txy, Ix[0], Iy[0]);
#pragma unroll
                for (int j = 1; j < K; ++j) {
                    const int wd = cand[K - 1 - j];
                    const float4 q0 = ldc4(a.refl, 2 * wd);
                    w[j] = make_wallc(q0, ldc4(a.refl, 2 * wd + 1), ldc4(a.flt, wd), wd);
                    image_of(q0, Ix[j - 1], Iy[j - 1], Ix[j], Iy[j]);
                }
                const unsigned long long sh0 = a.shadow ? cmem(a.shadow)[wl] : 0ull;
                // (order 1: the wall is also the one next to the cell -- the region's masks of the segment cell -> wall)
                const unsigned long long hk = (K == 1 && hidden_row && a.shadow) ? hidden_row[wl] : 0ull;
                if (alive && cull_candidate<K, true>(bx, by, w, Ix, Iy, a, sh0, a.on_lo, a.on_hi, hk, hidden_dperp)) alive = false;
            }
            unsigned long long mask = __ballot(alive);
            D2D_WORK(5 * K);
            if constexpr (EMIT) {
                unsigned long long code = (unsigned long long)cmem(a.cw)[lp < Nc ? lp : 0] << (12 * (K - 1));
#pragma unroll
                for (int d = 0; d < K - 1; ++d) code |= (unsigned long long)cand[d] << (12 * d);
                emit_batch(*emit, code, alive, mask);
            } else
            while (mask) {
                const int b = __builtin_ctzll(mask);
                mask &= mask - 1;
                cand[K - 1] = cmem(a.cw)[chunk * 64 + b];
                image_of(ldc4(a.refl, 2 * cand[K - 1]), K == 1 ? cx : imgx[K >= 2 ? K - 2 : 0], K == 1 ? cy : imgy[K >= 2 ? K - 2 : 0], imgx[K - 1], imgy[K - 1]);
                eval_candidate<K, MODE, false, GRAD, false, true>(a, cand, imgx, imgy, cx, cy, a.txx, a.txy, lane_bad, acc, st, g);
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
                if (pos[d] < (d == 0 ? p_end : Nc)) {
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

// Order K >= 2 of a TX-grid patch from its leaf region's candidate list (built by region_list_kernel / region_refine_kernel
// <K, GRAD, TXG = true> with the reversed chain): 64 entries per batch, lanes = candidates, the full culling test against
// the patch, survivors evaluated exactly in list order with the images of every lane's own cell.
template <int K, int MODE, bool GRAD>
__device__ __forceinline__ void sweep_order_listed_txg(const SweepArgs& a, const float4* tab, const float (&bx)[4], const float (&by)[4],
                                                       float cx, float cy, bool lane_bad, float& acc, WaveStats& st, GradCtx* g, long region) {
    static_assert(K >= 2, "lists exist for orders >= 2");
    const int lane = threadIdx.x & 63;
    const auto* rlc = cmem(a.rl);
    const int n = cmem(rlc->leaf.cnt[K])[region];
    int chunk = rlc->leaf.chunk0[K] + (int)region;
    const auto* pool = cmem(rlc->lp.pool);
    const auto* next = cmem(rlc->lp.next);
    for (int off = 0; off < n; off += 64) {
        const bool have = off + lane < n;
        const unsigned long long code = pool[(size_t)chunk * RL_CHUNK + (off & (RL_CHUNK - 1)) + (have ? lane : 0)];
        if ((off & (RL_CHUNK - 1)) == RL_CHUNK - 64 && off + 64 < n) chunk = next[chunk];
        float Ix[K], Iy[K];
        const unsigned long long* hid = D2D_HIDDEN_PATCH ? rlc->leaf.hidden : nullptr;
        unsigned long long mask = cull_batch<K, GRAD, true>(a, tab, bx, by, code, have, Ix, Iy, a.on_lo, a.on_hi, hid ? hid + (size_t)region * a.N : nullptr,
                                                            hid ? rlc->leaf.hidden_dperp : 0.0f);
        D2D_WORK(5 * K);
        while (mask) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            const unsigned lo32 = (unsigned)__builtin_amdgcn_readlane((int)(code & 0xffffffffull), b);
            const unsigned hi32 = (K >= 3) ? (unsigned)__builtin_amdgcn_readlane((int)(code >> 32), b) : 0u;
            const unsigned long long cu = ((unsigned long long)hi32 << 32) | lo32;
            int ce[D2D_MAX_ORDER] = {-1, -1, -1, -1};
            float ex[D2D_MAX_ORDER], ey[D2D_MAX_ORDER];  // images of the lane's cell (per lane)
#pragma unroll
            for (int d = 0; d < K; ++d) {
                ce[d] = (int)((cu >> (12 * d)) & 0xfffull);
                image_of(ldc4(a.refl, 2 * ce[d]), d == 0 ? cx : ex[d > 0 ? d - 1 : 0], d == 0 ? cy : ey[d > 0 ? d - 1 : 0], ex[d], ey[d]);
            }
            eval_candidate<K, MODE, false, GRAD, false, true>(a, ce, ex, ey, cx, cy, a.txx, a.txy, lane_bad, acc, st, g);
        }
    }
}

// GRADK: value + gradient (per cell w.r.t. the transmitter = the cell, scene.py:1617-1620; scene VJP w.r.t. the fixed
// receiver and the walls) of the candidates the culling cannot drop.  The reference's autodiff NaN artefacts are
// reproduced for those candidates only (the culling reasons about the reversed chain, whose poles are not the exact
// chain's): d2d_params.strict_nan selects the exhaustive power_vg_kernel.
// One TX-grid patch.  LISTED: the orders >= 2 come from the region candidate lists; a patch that cannot use them is queued
// for the enumerating build (LISTED = false), launched right behind with a.fb_n set.
template <int MODE, int MAXK, bool GRADK, bool LISTED>
__device__ __forceinline__ void txg_patch(const SweepArgs& a, const float4* tab, float* wl, const long b0, const bool from_queue) {
    const int lane = threadIdx.x & 63;
    const int tiles_x = (a.n + TILE_W - 1) / TILE_W;
    const bool scene = GRADK && a.partial != nullptr;
    if (scene) {
        for (int i = lane; i < 4 * a.N; i += 64) wl[i] = 0.0f;
        __syncthreads();
    }
    WaveStats st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st.c[i] = 0;
    st.shadow = -1;
    st.work = 0;
    const long tile = from_queue ? b0 : (a.sched ? (long)a.sched[b0] : b0);
    const int tcol = (int)(tile % tiles_x), trow = (int)(tile / tiles_x);
    const int col = tcol * TILE_W + (lane & (TILE_W - 1));
    const int row = trow * TILE_H + (lane / TILE_W);
    const bool in_range = (col < a.n) && (row < a.m);
    const int ccol = col < a.n ? col : a.n - 1;
    const int crow = row < a.m ? row : a.m - 1;
    const long idx = (long)crow * a.n + ccol;
    const float cx = a.X[idx], cy = a.Y[idx];
    const bool lane_bad = !(fabsf(cx) < 1e18f) || !(fabsf(cy) < 1e18f) || !(fabsf(a.txx) < 1e18f) || !(fabsf(a.txy) < 1e18f);
    const long region = LISTED ? region_of(a, tcol, trow) : 0;
    if (LISTED) {
        // not this kernel's patch: leave it to the enumerating kernel
        if (cmem(cmem(a.rl)->flag)[region] != 0 || wave_any(lane_bad)) {
            if (lane == 0) a.fb_list[atomicAdd(a.fb_n, 1)] = (int)tile;
            return;
        }
    }
    float acc = 0.0f;  // scene.py:1593
    GradCtx g;
    g.grx = g.gry = g.tbx = g.tby = 0.0f;
    g.cot = in_range ? (a.cot ? a.cot[idx] : 1.0f) : 0.0f;  // clamped duplicate lanes contribute nothing
    g.wl = wl;
    g.scene = scene;
    g.ci = 0;
    g.cell = 0;
    float x0 = cx, x1 = cx, y0 = cy, y1 = cy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, off, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, off, 64));
        y0 = fminf(y0, __shfl_xor(y0, off, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
    }
    const bool box_ok = !wave_any(lane_bad);
    const float qn = __builtin_nanf("");
    const float bx[4] = {box_ok ? x0 : qn, x1, x1, x0};
    const float by[4] = {y0, y0, y1, y1};
    if (a.min_order <= 0 && a.max_order >= 0) sweep_order<0, MODE, false, GRADK, true>(a, cx, cy, a.txx, a.txy, lane_bad, acc, st, &g);
    if (a.min_order <= 1 && a.max_order >= 1) {
        const unsigned long long* hid = LISTED ? cmem(a.rl)->leaf.hidden : nullptr;
        sweep_order_culled_txg<1, MODE, GRADK>(a, tab, bx, by, cx, cy, lane_bad, acc, st, &g, 0, 0x7fffffff, nullptr,
                                               hid ? hid + (size_t)region * a.N : nullptr, hid ? cmem(a.rl)->leaf.hidden_dperp : 0.0f);
    }
    if constexpr (LISTED) {
        if (a.min_order <= 2 && a.max_order >= 2) sweep_order_listed_txg<2, MODE, GRADK>(a, tab, bx, by, cx, cy, lane_bad, acc, st, &g, region);
        if (MAXK >= 3 && a.min_order <= 3 && a.max_order >= 3) sweep_order_listed_txg<3, MODE, GRADK>(a, tab, bx, by, cx, cy, lane_bad, acc, st, &g, region);
        if (MAXK >= 4 && a.min_order <= 4 && a.max_order >= 4) sweep_order_listed_txg<4, MODE, GRADK>(a, tab, bx, by, cx, cy, lane_bad, acc, st, &g, region);
    } else {
        if (a.min_order <= 2 && a.max_order >= 2) sweep_order_culled_txg<2, MODE, GRADK>(a, tab, bx, by, cx, cy, lane_bad, acc, st, &g);
        if (MAXK >= 3 && a.min_order <= 3 && a.max_order >= 3) sweep_order_culled_txg<3, MODE, GRADK>(a, tab, bx, by, cx, cy, lane_bad, acc, st, &g);
        if (MAXK >= 4 && a.min_order <= 4 && a.max_order >= 4) sweep_order_culled_txg<4, MODE, GRADK>(a, tab, bx, by, cx, cy, lane_bad, acc, st, &g);
    }
    if (in_range) {
        if (a.out_mode == D2D_OUT_ADD) {
            a.out[idx] = a.out[idx] + acc;
            if (GRADK) {
                a.grad[2 * idx] = a.grad[2 * idx] + g.grx;
                a.grad[2 * idx + 1] = a.grad[2 * idx + 1] + g.gry;
            }
        } else {
            a.out[idx] = acc;
            if (GRADK) {
                a.grad[2 * idx] = g.grx;
                a.grad[2 * idx + 1] = g.gry;
            }
        }
    }
    if (a.cost_out && lane == 0) a.cost_out[tile] = st.work;
    if (scene) {
        const float sx = wave_sum(g.tbx), sy = wave_sum(g.tby);
        __syncthreads();
        float* dst = a.partial + tile * (4 * a.N + 2);  // one row per patch, whatever the schedule
        for (int i = lane; i < 4 * a.N; i += 64) dst[i] = wl[i];
        if (lane == 0) {
            dst[4 * a.N] = sx;
            dst[4 * a.N + 1] = sy;
        }
    }
}

template <int MODE, int MAXK, bool GRADK = false, bool LISTED = false>
__global__ void __launch_bounds__(64) power_fwd_txg_kernel(SweepArgs a) {
    const int lane = threadIdx.x & 63;
    extern __shared__ float4 tab[];  // [2N] refl, [N] flt, then (GRADK) [N] float4 = the wave's scene-VJP partial sums
    float* wl = reinterpret_cast<float*>(tab + 3 * a.N);
    if (!LISTED && a.fb_n != nullptr) {
        const int n = *a.fb_n;  // the patches the LISTED launch in front of this one left behind (usually none)
        if (n > 0) {
            for (int i = lane; i < 2 * a.N; i += 64) tab[i] = ldc4(a.refl, i);
            for (int i = lane; i < a.N; i += 64) tab[2 * a.N + i] = ldc4(a.flt, i);
            __syncthreads();
        }
        for (int i = blockIdx.x; i < n; i += gridDim.x) {
            txg_patch<MODE, MAXK, GRADK, false>(a, tab, wl, (long)a.fb_list[i], true);
            __syncthreads();
        }
        return;
    }
    for (int i = lane; i < 2 * a.N; i += 64) tab[i] = ldc4(a.refl, i);
    for (int i = lane; i < a.N; i += 64) tab[2 * a.N + i] = ldc4(a.flt, i);
    __syncthreads();
    txg_patch<MODE, MAXK, GRADK, LISTED>(a, tab, wl, (long)blockIdx.x, false);
}

// Patch schedule.  The hardware starts workgroups in blockIdx order, and a dear patch that starts late is the tail of the
// launch (1024^2, 50 walls: patches cost up to 3.6x the mean; starting the dear ones first is worth 20-30 %).  Cost
// proxy of a patch: the number of order-1 candidates the tile culling cannot drop (first walls that are both reachable
// from the patch and not in the fixed end point's shadow) -- every one of them opens a prefix of higher-order candidates.
// patch_cost_kernel: one wave per patch -> key in [0, 63] + histogram;  patch_order_kernel: counting sort, dearest
// first.  Ties are placed in atomic order: the schedule may differ from run to run, the results cannot.
constexpr int SCHED_KEYS = 256;  // = blockDim of the two sort passes
#ifndef D2D_SCHED_PER_THREAD
#define D2D_SCHED_PER_THREAD 4
#endif
constexpr int SCHED_PER_THREAD = D2D_SCHED_PER_THREAD;  // patches per thread in the two counting-sort passes (16: 4 % slower steps at 1024^2 -- too few blocks)
#ifdef D2D_AUX_KERNELS  // non-template kernels: defined once, in d2d.hip
__global__ void __launch_bounds__(256) patch_cost_kernel(SweepArgs a, unsigned char* __restrict__ key) {
    const int lane = threadIdx.x & 63;
    const int tiles_x = (a.n + TILE_W - 1) / TILE_W;
    const int tiles_y = (a.m + TILE_H - 1) / TILE_H;
    const long n_tiles = (long)tiles_x * tiles_y;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;  // whole wave
    const int tcol = (int)(tile % tiles_x), trow = (int)(tile / tiles_x);
    const int col = tcol * TILE_W + (lane & (TILE_W - 1));
    const int row = trow * TILE_H + (lane / TILE_W);
    const int ccol = col < a.n ? col : a.n - 1;
    const int crow = row < a.m ? row : a.m - 1;
    const long idx = (long)crow * a.n + ccol;
    const float rxx = a.X[idx], rxy = a.Y[idx];
    float x0 = rxx, x1 = rxx, y0 = rxy, y1 = rxy;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, off, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, off, 64));
        y0 = fminf(y0, __shfl_xor(y0, off, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
    }
    const float bx[4] = {x0, x1, x1, x0};
    const float by[4] = {y0, y0, y1, y1};
    int alive_n = 0;
    for (int c0 = 0; c0 < a.Nc; c0 += 64) {
        const int lp = c0 + lane;
        const int wl = cmem(a.cw)[lp < a.Nc ? lp : 0];
        WallC w[1];
        float Ix[1], Iy[1];
        const float4 r0 = ldc4(a.refl, 2 * wl), r1 = ldc4(a.refl, 2 * wl + 1), fc = ldc4(a.flt, wl);
        w[0] = make_wallc(r0, r1, fc, wl);
        image_of(r0, a.txx, a.txy, Ix[0], Iy[0]);
        const unsigned long long sh0 = a.shadow ? cmem(a.shadow)[wl] : 0ull;
        const bool alive = lp < a.Nc && !cull_candidate<1>(bx, by, w, Ix, Iy, a, sh0, a.on_lo, a.on_hi);
        alive_n += __builtin_popcountll(__ballot(alive));
    }
    if (lane == 0) key[tile] = (unsigned char)(((long)alive_n * (SCHED_KEYS - 1)) / (a.Nc > 0 ? a.Nc : 1));
}

// Cost key of a patch from the work it took the last time this context swept the same grid (counted by the sweep kernels
// themselves in units of ~25 wave-instructions: deterministic, unlike elapsed time): 8 buckets per octave, 9 % resolution.
__device__ __forceinline__ unsigned char key_from_cost(unsigned cost) {
    const int k = (int)(8.0f * __log2f((float)(cost | 1u))) - 16;  // 2^2 .. 2^34 units -> 0 .. 255
    return (unsigned char)(k < 0 ? 0 : (k > SCHED_KEYS - 1 ? SCHED_KEYS - 1 : k));
}

// pass 1: hist[k] = number of patches with key k; with `cost` (work history) or `lists` (the lengths of the region
// candidate lists the launch has just built: a patch costs about what its region's lists hold) the keys are derived here
__global__ void __launch_bounds__(256) patch_hist_kernel(unsigned char* __restrict__ key, const unsigned* __restrict__ cost,
                                                         int* __restrict__ hist, long n_tiles, const RegionLists* __restrict__ lists,
                                                         int tiles_x, int k_lo, int k_hi) {
    __shared__ int cnt[SCHED_KEYS];
    if (threadIdx.x < SCHED_KEYS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * (256 * SCHED_PER_THREAD);
    for (int i = 0; i < SCHED_PER_THREAD; ++i) {
        const long t = base + (long)i * 256 + threadIdx.x;
        if (t < n_tiles) {
            unsigned char k;
            if (lists) {
                const int R = lists->leaf.R;
                const long region = (long)((int)(t / tiles_x) / R) * lists->leaf.regions_x + ((int)(t % tiles_x) / R);
                unsigned len = 4;
                for (int o = k_lo; o <= k_hi; ++o) {
                    const int c = lists->leaf.cnt[o][region];
                    len += c > 0 ? (unsigned)c : 0u;
                }
                k = key_from_cost(len << 4);
                key[t] = k;
            } else if (cost) {
                k = key_from_cost(cost[t]);
                key[t] = k;
            } else {
                k = key[t];
            }
            atomicAdd(&cnt[k], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < SCHED_KEYS && cnt[threadIdx.x]) atomicAdd(&hist[threadIdx.x], cnt[threadIdx.x]);
}

// pass 2: scatter, dearest key first
__global__ void __launch_bounds__(256) patch_order_kernel(const unsigned char* __restrict__ key, const int* __restrict__ hist,
                                                          int* __restrict__ cursor, int* __restrict__ sched, long n_tiles) {
    __shared__ int cnt[SCHED_KEYS], start[SCHED_KEYS];
    if (threadIdx.x < SCHED_KEYS) cnt[threadIdx.x] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * (256 * SCHED_PER_THREAD);
    int local[SCHED_PER_THREAD];
#pragma unroll
    for (int i = 0; i < SCHED_PER_THREAD; ++i) {
        const long t = base + (long)i * 256 + threadIdx.x;
        local[i] = (t < n_tiles) ? atomicAdd(&cnt[key[t]], 1) : 0;
    }
    __syncthreads();
    if (threadIdx.x < SCHED_KEYS) {
        int s = 0;
        for (int k = SCHED_KEYS - 1; k > (int)threadIdx.x; --k) s += hist[k];
        start[threadIdx.x] = s + (cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], cnt[threadIdx.x]) : 0);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SCHED_PER_THREAD; ++i) {
        const long t = base + (long)i * 256 + threadIdx.x;
        if (t < n_tiles) sched[start[key[t]] + local[i]] = (int)t;
    }
}

// The two passes above in ONE workgroup (launches of up to SORT1_MAX patches: 1024^2 has 16 384): keys, histogram, offsets and
// scatter all in LDS, nothing to zero beforehand and nothing for a second kernel to wait for.  Each wave counts into its own 256
// bins, so a wave's atomics only contend with themselves; within a key the patches are placed wave by wave, inside a wave in
// atomic order: the schedule may differ from run to run, the results cannot.
constexpr int SORT1_THREADS = 1024;
constexpr int SORT1_BATCH = 16;  // loads in flight per thread
constexpr long SORT1_MAX = 1 << 16;
// Round 4: the patches' work counters / keys are fetched sixteen at a time per thread (the plain loop was one dependent round
// trip to memory per patch).  (Four waves instead of sixteen -- a 1024-thread workgroup needs sixteen free wave slots on ONE CU,
// beside a sweep that fills the chip -- were measured too: 48 us against 20, the kernel is as fast as it has waves.)
__global__ void __launch_bounds__(SORT1_THREADS) patch_sort_kernel(unsigned char* __restrict__ key, const unsigned* __restrict__ cost,
                                                                   int* __restrict__ sched, long n_tiles, const RegionLists* __restrict__ lists,
                                                                   int tiles_x, int k_lo, int k_hi) {
    constexpr int W = SORT1_THREADS / 64;
    __shared__ int cnt[W][SCHED_KEYS];  // per wave: patches per key, then the wave's cursor within the key
    __shared__ int tot[SCHED_KEYS];
    const int wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < W * SCHED_KEYS; i += SORT1_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    // pass 1: keys (work history, list lengths, or given) and per-wave histograms
    for (long t0 = 0; t0 < n_tiles; t0 += (long)SORT1_THREADS * SORT1_BATCH) {
        unsigned v[SORT1_BATCH];
#pragma unroll
        for (int i = 0; i < SORT1_BATCH; ++i) {
            const long t = t0 + (long)i * SORT1_THREADS + threadIdx.x;
            v[i] = 0u;
            if (t < n_tiles) {
                if (lists) {
                    const int R = lists->leaf.R;
                    const long region = (long)((int)(t / tiles_x) / R) * lists->leaf.regions_x + ((int)(t % tiles_x) / R);
                    unsigned len = 4;
                    for (int o = k_lo; o <= k_hi; ++o) {
                        const int c = lists->leaf.cnt[o][region];
                        len += c > 0 ? (unsigned)c : 0u;
                    }
                    v[i] = len << 4;
                } else if (cost) {
                    v[i] = cost[t];
                } else {
                    v[i] = key[t];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SORT1_BATCH; ++i) {
            const long t = t0 + (long)i * SORT1_THREADS + threadIdx.x;
            if (t < n_tiles) {
                unsigned char k;
                if (lists || cost) {
                    k = key_from_cost(v[i]);
                    key[t] = k;
                } else {
                    k = (unsigned char)v[i];
                }
                atomicAdd(&cnt[wv][k], 1);
            }
        }
    }
    __syncthreads();
    // offsets: key descending (dearest first), within a key wave ascending
    if (threadIdx.x < SCHED_KEYS) {
        int s = 0;
        for (int w = 0; w < W; ++w) s += cnt[w][threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
    if (threadIdx.x < SCHED_KEYS) {
        int s = 0;
        for (int k = SCHED_KEYS - 1; k > (int)threadIdx.x; --k) s += tot[k];
        for (int w = 0; w < W; ++w) {
            const int c = cnt[w][threadIdx.x];
            cnt[w][threadIdx.x] = s;  // from a count to the wave's first slot
            s += c;
        }
    }
    __syncthreads();
    // pass 2: scatter (same thread -> same patches -> same wave as in pass 1)
    for (long t0 = 0; t0 < n_tiles; t0 += (long)SORT1_THREADS * SORT1_BATCH) {
        unsigned char k[SORT1_BATCH];
#pragma unroll
        for (int i = 0; i < SORT1_BATCH; ++i) {
            const long t = t0 + (long)i * SORT1_THREADS + threadIdx.x;
            k[i] = t < n_tiles ? key[t] : (unsigned char)0;
        }
#pragma unroll
        for (int i = 0; i < SORT1_BATCH; ++i) {
            const long t = t0 + (long)i * SORT1_THREADS + threadIdx.x;
            if (t < n_tiles) sched[atomicAdd(&cnt[wv][k[i]], 1)] = (int)t;
        }
    }
}

// Bounding boxes of the cells of every region of R x R patches (one 256-thread workgroup per region): {x0, x1, y0, y1},
// x0 = NaN when some cell is not comfortably finite.  Depends on the grid only: rebuilt when the grid or R changes.
__global__ void __launch_bounds__(256) region_box_kernel(const float* __restrict__ X, const float* __restrict__ Y, int m, int n, int R,
                                                         int regions_x, float4* __restrict__ box) {
    const int region = blockIdx.x;
    const int rx = region % regions_x, ry = region / regions_x;
    const int c0 = rx * R * TILE_W, r0 = ry * R * TILE_H;
    const int c1 = min(c0 + R * TILE_W, n), r1 = min(r0 + R * TILE_H, m);
    const int w = c1 - c0, cells = w * (r1 - r0);
    const float inf = __builtin_inff();
    float x0 = inf, x1 = -inf, y0 = inf, y1 = -inf;
    bool bad = false;
    for (int i = threadIdx.x; i < cells; i += 256) {
        const long idx = (long)(r0 + i / w) * n + (c0 + i % w);
        const float x = X[idx], y = Y[idx];
        bad = bad || !(fabsf(x) < 1e18f) || !(fabsf(y) < 1e18f);
        x0 = fminf(x0, x);
        x1 = fmaxf(x1, x);
        y0 = fminf(y0, y);
        y1 = fmaxf(y1, y);
    }
    __shared__ float sm[4][4];
    __shared__ int sbad[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, off, 64));
        x1 = fmaxf(x1, __shfl_xor(x1, off, 64));
        y0 = fminf(y0, __shfl_xor(y0, off, 64));
        y1 = fmaxf(y1, __shfl_xor(y1, off, 64));
    }
    const bool wbad = __any(bad);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sm[wv][0] = x0; sm[wv][1] = x1; sm[wv][2] = y0; sm[wv][3] = y1;
        sbad[wv] = wbad ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) {
            x0 = fminf(x0, sm[i][0]); x1 = fmaxf(x1, sm[i][1]); y0 = fminf(y0, sm[i][2]); y1 = fmaxf(y1, sm[i][3]);
        }
        const bool anybad = sbad[0] | sbad[1] | sbad[2] | sbad[3];
        box[region] = make_float4(anybad ? __builtin_nanf("") : x0, x1, y0, y1);
    }
}

#ifdef D2D_AB_TIMELINE
// (the launch's end: a one-thread kernel behind the sweep on its stream -- 21 000 atomics on one word took 340 us)
__global__ void tl_end_kernel(unsigned long long* __restrict__ ring, int seq) { ring[2 * (seq & 255) + 1] = __builtin_amdgcn_s_memrealtime(); }
#endif
// The lists' descriptor as the sweep kernels read it (a.rl), written in stream order from a by-value argument.
__global__ void write_region_lists_kernel(RegionLists* __restrict__ dst, RegionLists v) { *dst = v; }

// What a launch needs zeroed (shadow masks, sort counters, list bookkeeping): a kernel of its own rather than
// hipMemsetAsync, which the runtime does not let run ahead on the side stream.
__global__ void __launch_bounds__(256) zero_words_kernel(unsigned long long* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0ull;
}

// Self-test of the bare division chain against the compiler's generic expansion (bit equality expected).
__global__ void selftest_div_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ q_fast,
                                    float* __restrict__ q_ref, float* __restrict__ q_hostr, const float* __restrict__ ry, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q_fast[i] = div_with_rcp(x[i], y[i], rcp_refined(y[i]));
    q_ref[i] = x[i] / y[i];
    q_hostr[i] = div_with_rcp(x[i], y[i], ry[i]);  // with a host-computed correctly rounded reciprocal
}

// expf_libm on the device, for comparison with the host C library's expf (tests/test_gpu_selftest.py)
__global__ void selftest_expf_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = expf_libm(x[i]);
}

// Shadow coverage of every wall as seen from the fixed end point `e` (the transmitter of an RX-grid sweep): one thread per
// (wall w, blocker j) pair rasterises, into 64 bins of w's parametric range, where the segment e -> p is CERTAINLY
// reported as intersecting j by the exact path (hard: hit; approx: the four activations exactly saturated), for every
// p within `dperp` of the bin.  t_a, t_b are linear-fractional in p, so on a thin quad around a bin that does not meet
// the pole (fd keeps its sign) their ranges are spanned by the 4 vertices.  Margins: 8 eps per product sum for the
// rounding of either evaluation chain.  Only bins certified at all four vertices are set (atomicOr).
// (one (wall w, blocker j) pair, lane = bin b: is bin b of w certainly hidden from (ex, ey) by j?)
__device__ __forceinline__ bool shadow_pair_bin(const float4* __restrict__ occl, const float4* __restrict__ refl,
                                                const unsigned char* __restrict__ kind, int w, int j, int b, float ex, float ey, float win_lo,
                                                float win_hi, float dperp, float dom_lo, float dom_w) {
    if (w == j) return false;                  // segment 0 ignores the wall it ends on (geometry.py:881-890)
    if (kind[j] == D2D_VERTEX) return false;   // vertices never occlude (geometry.py:407-414)
    {
        const float4 t1 = refl[2 * w + 1];
        if (t1.x * t1.x + t1.y * t1.y == 0.0f) return false;  // a zero-length wall has no line for its point to lie on
    }
    const float eps = 1.1920929e-07f;
    const float4 r0 = refl[2 * w], r1 = refl[2 * w + 1];
    const float4 o = occl[j];  // P1, A
    const float Cx = o.x - ex, Cy = o.y - ey;                 // C = P1 - P3, P3 = e
    const float fb = o.z * Cy - o.w * Cx;
    const float errB = 8.0f * eps * (fabsf(o.z * Cy) + fabsf(o.w * Cx));
    const float tlen = fabsf(r1.x) + fabsf(r1.y);
    const float pad = dperp * (r1.w > 0.0f ? 1.0f / r1.w : 0.0f);  // dperp expressed in parametric units of w
    bool ok = true;
    int sgn = 0;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const float sg = dom_lo + ((v & 1) ? (float)(b + 1) * dom_w + pad : (float)b * dom_w - pad);
        const float off = (v & 2) ? dperp : -dperp;
        const float qx = r0.x + sg * r1.x + off * r0.z, qy = r0.y + sg * r1.y + off * r0.w;  // P4 = q
        const float Bx = ex - qx, By = ey - qy;
        const float fa = By * Cx - Bx * Cy;
        const float fd = o.w * Bx - o.z * By;
        const float errA = 8.0f * eps * (fabsf(By * Cx) + fabsf(Bx * Cy));
        const float errD = 8.0f * eps * (fabsf(o.w * Bx) + fabsf(o.z * By)) + 4.0f * eps * tlen * (fabsf(o.z) + fabsf(o.w));
        const float ad = fabsf(fd);
        if (!(ad > 8.0f * errD)) { ok = false; break; }
        const int sv = fd > 0.0f ? 1 : -1;
        if (sgn == 0) sgn = sv;
        if (sv != sgn) { ok = false; break; }
        // (reciprocals instead of four correctly rounded divisions per vertex: v_rcp_f32 is good to 1 ulp, a quotient formed
        // with it to 3 -- |t| <= 1.01 wherever the test can still pass --, and the bounds are rounded up: all inside + 8 eps;
        // this kernel runs once per launch beside the previous sweep, one dependent chain per wave: 15 -> 8 us)
        const float rfd = __builtin_amdgcn_rcpf(fd), rin = __builtin_amdgcn_rcpf(ad - errD) * 1.000001f;
        const float ta = fa * rfd, tb = fb * rfd;
        const float ea = (errA + 2.0f * errD) * rin + 8.0f * eps, eb = (errB + 2.0f * errD) * rin + 8.0f * eps;
        if (!(ta - ea >= win_lo && ta + ea <= win_hi && tb - eb >= win_lo && tb + eb <= win_hi)) { ok = false; break; }
    }
    return ok;
}

__global__ void shadow_tx_kernel(const float4* __restrict__ occl, const float4* __restrict__ refl,
                                 const unsigned char* __restrict__ kind, int N, float ex, float ey, float win_lo, float win_hi,
                                 float dperp, float dom_lo, float dom_w, unsigned long long* __restrict__ shadow) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (int)(gid & 63);       // one lane per bin: a wave = one (w, j) pair, its ballot = the pair's 64 bits
    const long idx = gid >> 6;
    if (idx >= (long)N * N) return;
    const int w = (int)(idx / N), j = (int)(idx % N);
    const unsigned long long bits = __ballot(shadow_pair_bin(occl, refl, kind, w, j, b, ex, ey, win_lo, win_hi, dperp, dom_lo, dom_w));
    if (bits && b == 0) atomicOr(&shadow[w], bits);
}

// The same masks by ONE kernel that needs nothing zeroed beforehand and zeroes what the rest of the launch's preparation
// wants zeroed (sort counters of big launches, the lists' bookkeeping: `zero[0 .. n_zero)`): workgroup w < N owns wall w -- its
// four waves take every fourth blocker each, OR their ballots in registers, combine through LDS and STORE the mask --, the
// workgroups behind them clear 256 words each.  N + a few workgroups instead of N x N single-wave ones and a memset kernel in
// front: beside a sweep that fills the chip, what a small kernel costs is its wave count and the kernel boundaries.
__global__ void __launch_bounds__(256) shadow_fill_kernel(const float4* __restrict__ occl, const float4* __restrict__ refl,
                                                          const unsigned char* __restrict__ kind, int N, float ex, float ey, float win_lo,
                                                          float win_hi, float dperp, float dom_lo, float dom_w, int want_masks,
                                                          unsigned long long* __restrict__ shadow, unsigned long long* __restrict__ zero,
                                                          long n_zero) {
    if ((int)blockIdx.x >= N) {
        const long i = (long)(blockIdx.x - N) * 256 + threadIdx.x;
        if (i < n_zero) zero[i] = 0ull;
        return;
    }
    __shared__ unsigned long long part[4];
    const int w = blockIdx.x, wv = threadIdx.x >> 6, b = threadIdx.x & 63;
    unsigned long long bits = 0ull;
    if (want_masks)
        for (int j = wv; j < N; j += 4) bits |= __ballot(shadow_pair_bin(occl, refl, kind, w, j, b, ex, ey, win_lo, win_hi, dperp, dom_lo, dom_w));
    if (b == 0) part[wv] = bits;
    __syncthreads();
    if (threadIdx.x == 0) shadow[w] = part[0] | part[1] | part[2] | part[3];
}

// Wall-to-wall occlusion masks (scene-only: no end point involved, so they are built once per scene / mode).  One wave
// per (earlier wall we, later wall wl, blocker j); lane = (bin of we) + 8 * (bin of wl).  A bit is set when the segment
// p -> q is CERTAINLY reported as intersecting j by the exact path for every p within dperp of we's bin and every q
// within dperp of wl's bin: t_a, t_b and the denominator are (bi)linear-fractional in (p, q), monotone along straight
// lines in either argument while fd keeps its sign, so their ranges over the two thin quads are spanned by the 4 x 4
// vertex pairs.  Same margins as shadow_tx_kernel.
__global__ void __launch_bounds__(256) pair_shadow_kernel(const float4* __restrict__ occl, const float4* __restrict__ refl,
                                                          const unsigned char* __restrict__ kind, int N, float win_lo, float win_hi,
                                                          float dperp, float dom_lo, float dom_w8, unsigned long long* __restrict__ pair) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(gid & 63);
    const long idx = gid >> 6;
    if (idx >= (long)N * N * N) return;
    const int j = (int)(idx % N), wl = (int)((idx / N) % N), we = (int)(idx / ((long)N * N));
    if (we == wl || j == we || j == wl) return;  // a segment ignores the two walls it joins (geometry.py:881-904)
    if (kind[j] == D2D_VERTEX) return;
    const float4 a0 = refl[2 * we], a1 = refl[2 * we + 1], b0 = refl[2 * wl], b1 = refl[2 * wl + 1];
    if (a1.x * a1.x + a1.y * a1.y == 0.0f || b1.x * b1.x + b1.y * b1.y == 0.0f) return;
    const float eps = 1.1920929e-07f;
    const float4 o = occl[j];  // P1, A
    const int be = lane & 7, bl = lane >> 3;
    const float pad_e = dperp * (a1.w > 0.0f ? 1.0f / a1.w : 0.0f), pad_l = dperp * (b1.w > 0.0f ? 1.0f / b1.w : 0.0f);
    const float tlen = fabsf(a1.x) + fabsf(a1.y) + fabsf(b1.x) + fabsf(b1.y);
    bool ok = true;
    int sgn = 0;
    for (int vp = 0; vp < 4 && ok; ++vp) {
        const float sp = dom_lo + ((vp & 1) ? (float)(be + 1) * dom_w8 + pad_e : (float)be * dom_w8 - pad_e);
        const float op = (vp & 2) ? dperp : -dperp;
        const float px = a0.x + sp * a1.x + op * a0.z, py = a0.y + sp * a1.y + op * a0.w;  // P3 = p
        const float Cx = o.x - px, Cy = o.y - py;
        const float fb = o.z * Cy - o.w * Cx;
        const float errB = 8.0f * eps * (fabsf(o.z * Cy) + fabsf(o.w * Cx)) + 4.0f * eps * tlen * (fabsf(o.z) + fabsf(o.w));
        for (int vq = 0; vq < 4; ++vq) {
            const float sq = dom_lo + ((vq & 1) ? (float)(bl + 1) * dom_w8 + pad_l : (float)bl * dom_w8 - pad_l);
            const float oq = (vq & 2) ? dperp : -dperp;
            const float qx = b0.x + sq * b1.x + oq * b0.z, qy = b0.y + sq * b1.y + oq * b0.w;  // P4 = q
            const float Bx = px - qx, By = py - qy;
            const float fa = By * Cx - Bx * Cy;
            const float fd = o.w * Bx - o.z * By;
            const float errA = 8.0f * eps * (fabsf(By * Cx) + fabsf(Bx * Cy)) + 4.0f * eps * tlen * (fabsf(Cx) + fabsf(Cy) + fabsf(Bx) + fabsf(By));
            const float errD = 8.0f * eps * (fabsf(o.w * Bx) + fabsf(o.z * By)) + 4.0f * eps * tlen * (fabsf(o.z) + fabsf(o.w));
            const float ad = fabsf(fd);
            if (!(ad > 8.0f * errD)) { ok = false; break; }
            const int sv = fd > 0.0f ? 1 : -1;
            if (sgn == 0) sgn = sv;
            if (sv != sgn) { ok = false; break; }
            const float ta = fa / fd, tb = fb / fd;
            const float ea = (errA + 2.0f * errD) / (ad - errD) + 4.0f * eps, eb = (errB + 2.0f * errD) / (ad - errD) + 4.0f * eps;
            if (!(ta - ea >= win_lo && ta + ea <= win_hi && tb - eb >= win_lo && tb + eb <= win_hi)) { ok = false; break; }
        }
    }
    const unsigned long long bits = __ballot(ok);
    if (bits && lane == 0) atomicOr(&pair[(size_t)we * N + wl], bits);
}

// Last-segment masks (scene, grid and validity mode only; built once and kept): one wave per (leaf region r, wall w),
// lane = bin b of w.  Bit b is set when, for some blocker j, the segment p -> q is CERTAINLY reported as intersecting j by
// the exact path for every p within dperp of w's bin b (P3, the last interaction point) and every q of the region's
// bounding box (P4, the cell): the same bilinear-fractional argument and margins as pair_shadow_kernel, the second quad
// being the box (its corners are exact inputs; the margins derived for a rounded point only widen the bound).  swap (TX
// grids): the segment runs from the cell to the path's FIRST wall, so the box is P3 and the wall's bin P4.
__global__ void __launch_bounds__(64) hidden_region_kernel(const float4* __restrict__ occl, const float4* __restrict__ refl,
                                                           const unsigned char* __restrict__ kind, int N, const float4* __restrict__ box,
                                                           float win_lo, float win_hi, float dperp, float dom_lo, float dom_w,
                                                           unsigned long long* __restrict__ hidden, int swap) {
    const int b = threadIdx.x & 63;
    const long rw = blockIdx.x;
    const long r = rw / N;
    const int w = (int)(rw % N);
    const float4 bb = box[r];  // {x0, x1, y0, y1}; x0 = NaN: a cell is not comfortably finite
    const float4 a0 = refl[2 * w], a1 = refl[2 * w + 1];
    unsigned long long bits = 0ull;
    const bool usable = (bb.x <= bb.y) && (bb.z <= bb.w) && (a1.x * a1.x + a1.y * a1.y != 0.0f);
    if (usable) {
        const float eps = 1.1920929e-07f;
        const float pad = dperp * (a1.w > 0.0f ? 1.0f / a1.w : 0.0f);
        const float tlen = fabsf(a1.x) + fabsf(a1.y);
        float pxv[4], pyv[4];
#pragma unroll
        for (int vp = 0; vp < 4; ++vp) {
            const float sp = dom_lo + ((vp & 1) ? (float)(b + 1) * dom_w + pad : (float)b * dom_w - pad);
            const float op = (vp & 2) ? dperp : -dperp;
            pxv[vp] = a0.x + sp * a1.x + op * a0.z;
            pyv[vp] = a0.y + sp * a1.y + op * a0.w;
        }
        for (int j = 0; j < N; ++j) {
            if (j == w || kind[j] == D2D_VERTEX) continue;  // the segment ignores the wall it starts on; vertices never occlude
            const float4 o = occl[j];  // P1, A
            bool ok = true;
            int sgn = 0;
            for (int vp = 0; vp < 4 && ok; ++vp) {
                // RX grids: P3 = p on the wall (the last interaction point), P4 = q in the box (the cell); TX grids (swap): the
                // cell is the earlier point of the segment, P3 = q, P4 = p
                const float px = swap ? ((vp & 1) ? bb.y : bb.x) : pxv[vp], py = swap ? ((vp & 2) ? bb.w : bb.z) : pyv[vp];
                const float Cx = o.x - px, Cy = o.y - py;
                const float fb = o.z * Cy - o.w * Cx;
                const float errB = 8.0f * eps * (fabsf(o.z * Cy) + fabsf(o.w * Cx)) + 4.0f * eps * tlen * (fabsf(o.z) + fabsf(o.w));
                for (int vq = 0; vq < 4; ++vq) {
                    const float qx = swap ? pxv[vq] : ((vq & 1) ? bb.y : bb.x), qy = swap ? pyv[vq] : ((vq & 2) ? bb.w : bb.z);
                    const float Bx = px - qx, By = py - qy;
                    const float fa = By * Cx - Bx * Cy;
                    const float fd = o.w * Bx - o.z * By;
                    const float errA = 8.0f * eps * (fabsf(By * Cx) + fabsf(Bx * Cy)) + 4.0f * eps * tlen * (fabsf(Cx) + fabsf(Cy) + fabsf(Bx) + fabsf(By));
                    const float errD = 8.0f * eps * (fabsf(o.w * Bx) + fabsf(o.z * By)) + 4.0f * eps * tlen * (fabsf(o.z) + fabsf(o.w));
                    const float ad = fabsf(fd);
                    if (!(ad > 8.0f * errD)) { ok = false; break; }
                    const int sv = fd > 0.0f ? 1 : -1;
                    if (sgn == 0) sgn = sv;
                    if (sv != sgn) { ok = false; break; }
                    const float ta = fa / fd, tb = fb / fd;
                    const float ea = (errA + 2.0f * errD) / (ad - errD) + 4.0f * eps, eb = (errB + 2.0f * errD) / (ad - errD) + 4.0f * eps;
                    if (!(ta - ea >= win_lo && ta + ea <= win_hi && tb - eb >= win_lo && tb + eb <= win_hi)) { ok = false; break; }
                }
            }
            bits |= __ballot(ok);
            if (bits == ~0ull) break;
        }
    }
    if (b == 0) hidden[rw] = bits;
}

#endif  // D2D_AUX_KERNELS

// Value + gradient sweep: same forward arithmetic as power_fwd_kernel (bit-identical values), plus the
// hand-derived adjoint of every contributing candidate.  One wave per block; the wave's partial sums of
// the scene-parameter VJP live in LDS and are written to `partial` (reduced in fixed order afterwards,
// so results are reproducible run to run).
// GRADK = 0: values only (the exhaustive TX-grid value sweep behind the "txg_exhaustive" option).
template <int MODE, bool TXG, bool GRADK>
__global__ void __launch_bounds__(64) power_vg_kernel(SweepArgs a) {
    extern __shared__ float wl[];  // [4 N]
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
    const float gx_ = a.X[idx], gy_ = a.Y[idx];
    // (a.txx, a.txy) is the FIXED end point: the transmitter for an RX grid, the receiver for a TX grid
    const float txx = TXG ? gx_ : a.txx, txy = TXG ? gy_ : a.txy;
    const float rxx = TXG ? a.txx : gx_, rxy = TXG ? a.txy : gy_;
    const bool lane_bad = !(fabsf(gx_) < 1e18f) || !(fabsf(gy_) < 1e18f) || !(fabsf(a.txx) < 1e18f) ||
                          !(fabsf(a.txy) < 1e18f);
    const bool scene = GRADK && a.partial != nullptr;
    if (scene) {
        for (int i = lane; i < 4 * a.N; i += 64) wl[i] = 0.0f;
        __syncthreads();
    }
    GradCtx g;
    g.grx = g.gry = g.tbx = g.tby = 0.0f;
    g.cot = in_range ? (a.cot ? a.cot[idx] : 1.0f) : 0.0f;  // clamped duplicate lanes contribute nothing
    g.wl = wl;
    g.scene = scene;
    g.ci = 0;
    g.cell = idx;
    float acc = 0.0f;
    WaveStats st;
    st.shadow = -1;
    st.work = 0;
    if (a.min_order <= 0 && a.max_order >= 0) sweep_order<0, MODE, false, GRADK, TXG>(a, txx, txy, rxx, rxy, lane_bad, acc, st, &g);
    if (a.min_order <= 1 && a.max_order >= 1) sweep_order<1, MODE, false, GRADK, TXG>(a, txx, txy, rxx, rxy, lane_bad, acc, st, &g);
    if (a.min_order <= 2 && a.max_order >= 2) sweep_order<2, MODE, false, GRADK, TXG>(a, txx, txy, rxx, rxy, lane_bad, acc, st, &g);
    if (a.min_order <= 3 && a.max_order >= 3) sweep_order<3, MODE, false, GRADK, TXG>(a, txx, txy, rxx, rxy, lane_bad, acc, st, &g);
    if (a.min_order <= 4 && a.max_order >= 4) sweep_order<4, MODE, false, GRADK, TXG>(a, txx, txy, rxx, rxy, lane_bad, acc, st, &g);
    if (in_range) {
        if (a.out_mode == D2D_OUT_ADD) {
            a.out[idx] = a.out[idx] + acc;
            if (GRADK) {
                a.grad[2 * idx] = a.grad[2 * idx] + g.grx;
                a.grad[2 * idx + 1] = a.grad[2 * idx + 1] + g.gry;
            }
        } else {
            a.out[idx] = acc;
            if (GRADK) {
                a.grad[2 * idx] = g.grx;
                a.grad[2 * idx + 1] = g.gry;
            }
        }
    }
    if (scene) {
        float sx = wave_sum(g.tbx), sy = wave_sum(g.tby);
        __syncthreads();
        float* dst = a.partial + (long)blockIdx.x * (4 * a.N + 2);
        for (int i = lane; i < 4 * a.N; i += 64) dst[i] = wl[i];
        if (lane == 0) {
            dst[4 * a.N] = sx;
            dst[4 * a.N + 1] = sy;
        }
    }
}

#ifdef D2D_AUX_KERNELS  // non-template kernels: defined once, in d2d.hip
// Fixed-order reduction of the per-wave partials: out[e] (+)= sum_w partial[w][e], accumulated in fp64.
__global__ void __launch_bounds__(256) vjp_reduce_kernel(const float* __restrict__ partial, long n_waves, int n_elem,
                                                         double* __restrict__ out, int accumulate) {
    __shared__ double sm[256];
    const int e = blockIdx.x;
    double s = 0.0;
    for (long w = threadIdx.x; w < n_waves; w += 256) s += (double)partial[w * n_elem + e];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) sm[threadIdx.x] += sm[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[e] = (accumulate ? out[e] : 0.0) + sm[0];
}

#endif  // D2D_AUX_KERNELS

}  // namespace d2d

// =====================================================================================
// Generic (any object kind, any solver) literal evaluation: used by the path tracing kernel
// ("emit paths") and by the optimiser-based sweeps (MinPath / FermatPath).  Small problem sizes:
// written literally (every activation evaluated, NaN-propagating min/max), no skipping.
// =====================================================================================
namespace d2d {

// NaN-propagating min / max (jnp.minimum / jnp.maximum)
__device__ __forceinline__ float minp(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a < b ? a : b); }
__device__ __forceinline__ float maxp(float a, float b) { return (a != a || b != b) ? __builtin_nanf("") : (a > b ? a : b); }

struct ObjTables {
    const float4* __restrict__ occl;         // [N] {P1x, P1y, Ax, Ay}
    const float4* __restrict__ refl;         // [2N] {ox, oy, nx, ny}, {tx, ty, sq, |t|}
    const unsigned char* __restrict__ kind;  // [N] D2D_WALL / D2D_RIS / D2D_VERTEX
    const float2* __restrict__ sincos;       // [N] {sin(phi), cos(phi)} (RIS)
    const float4* __restrict__ xys;          // [N] {origin.x, origin.y, dest.x, dest.y}: the raw end points (gradient sweeps)
    int N;
};

struct Truth {
    int mode;
    float alpha;
    __device__ float act(float x) const {
        float z = alpha * x;
        if (mode == MODE_HSIG) return minp(maxp(z + 3.0f, 0.0f), 6.0f) / 6.0f;
        return 1.0f / (1.0f + expf_libm(-z));
    }
    __device__ float t_and(float a, float b) const { return mode ? minp(a, b) : ((a != 0.0f && b != 0.0f) ? 1.0f : 0.0f); }
    __device__ float t_or(float a, float b) const { return mode ? maxp(a, b) : ((a != 0.0f || b != 0.0f) ? 1.0f : 0.0f); }
    __device__ float t_not(float a) const { return mode ? (1.0f - a) : (a != 0.0f ? 0.0f : 1.0f); }
    __device__ float ge(float x, float y) const { return mode ? act(x - y) : (x >= y ? 1.0f : 0.0f); }
    __device__ float le(float x, float y) const { return mode ? act(y - x) : (x <= y ? 1.0f : 0.0f); }
    __device__ float lt(float x, float y) const { return mode ? act(y - x) : (x < y ? 1.0f : 0.0f); }
};

constexpr int NP = D2D_MAX_ORDER + 2;

// sum_k obj_k.evaluate_cartesian(P[k:k+3]): Wall geometry.py:641-650, RIS :698-711, Vertex :416-419
__device__ __forceinline__ float interaction_loss(const ObjTables& T, int k, const int (&cd)[D2D_MAX_ORDER], const float (&px)[NP],
                                                  const float (&py)[NP]) {
    float loss = 0.0f;
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) {
        if (i < k) {
            const int kd = T.kind[cd[i]];
            const float4 r0 = T.refl[2 * cd[i]];
            float ev = 0.0f;
            if (kd == D2D_WALL) {
                float ix_, iy_, rx_, ry_;
                normalize2(px[i + 1] - px[i], py[i + 1] - py[i], ix_, iy_);
                normalize2(px[i + 2] - px[i + 1], py[i + 2] - py[i + 1], rx_, ry_);
                float din = ix_ * r0.z + iy_ * r0.w;
                float s2 = 2.0f * din;
                float ex = rx_ - (ix_ - s2 * r0.z);
                float ey = ry_ - (iy_ - s2 * r0.w);
                ev = ex * ex + ey * ey;
            } else if (kd == D2D_RIS) {
                float rx_, ry_;
                normalize2(px[i + 2] - px[i + 1], py[i + 2] - py[i + 1], rx_, ry_);
                float mx = -rx_, my = -ry_;
                float sin_a = mx * r0.w - my * r0.z;
                float cos_a = mx * r0.z + my * r0.w;
                const float2 sc = T.sincos[cd[i]];
                float ds = sin_a - sc.x, dc = cos_a - sc.y;
                ev = ds * ds + dc * dc;
            }
            loss = loss + ev;
        }
    }
    return loss;
}

// path_length, geometry.py:176-203
__device__ __forceinline__ float literal_length(int k, const float (&px)[NP], const float (&py)[NP]) {
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i <= D2D_MAX_ORDER; ++i) {
        if (i <= k) {
            float vx = (px[i + 1] - px[i]) + D2D_EPS;
            float vy = (py[i + 1] - py[i]) + D2D_EPS;
            r = r + sqrtf(vx * vx + vy * vy);
        }
    }
    return r;
}

// on_objects / intersects_with_objects / is_valid, geometry.py:821-963, for any mix of object kinds
__device__ __forceinline__ void literal_validity(const ObjTables& T, const Truth& L, int k, const int (&cd)[D2D_MAX_ORDER],
                                                 const float (&px)[NP], const float (&py)[NP], float loss, float tol,
                                                 float seg_lo, float seg_hi, float& on, float& hit, float& valid) {
    on = 1.0f;
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) {
        if (i < k) {
            float cval;
            if (T.kind[cd[i]] == D2D_VERTEX) {
                cval = 1.0f;  // geometry.py:397-403
            } else {
                const float4 r0 = T.refl[2 * cd[i]];
                const float4 r1 = T.refl[2 * cd[i] + 1];
                float dx = px[i + 1] - r0.x, dy = py[i + 1] - r0.y;
                float s = (r1.x * dx + r1.y * dy) / r1.z;
                cval = L.t_and(L.ge(s, 0.0f), L.le(s, 1.0f));
            }
            on = L.t_and(on, cval);
        }
    }
    hit = 0.0f;
#pragma unroll
    for (int i = 0; i <= D2D_MAX_ORDER; ++i) {
        if (i <= k) {
            const int ig0 = (i == 0) ? -1 : cd[i - 1];
            const int ig1 = (i == k) ? -1 : cd[i < D2D_MAX_ORDER ? i : 0];
            const float bx = px[i] - px[i + 1], by = py[i] - py[i + 1];
            for (int j = 0; j < T.N; ++j) {
                if (j == ig0 || j == ig1) continue;
                if (T.kind[j] == D2D_VERTEX) continue;  // geometry.py:407-414: false_value; or() leaves hit unchanged
                const float4 w = T.occl[j];
                float Cx = w.x - px[i], Cy = w.y - py[i];
                float fa = by * Cx - bx * Cy;
                float fb = w.z * Cy - w.w * Cx;
                float fd = w.w * bx - w.z * by;
                bool dz = (fd == 0.0f);
                float dd = dz ? 1.0f : fd;
                float ta = dz ? __builtin_inff() : fa / dd;
                float tb = dz ? __builtin_inff() : fb / dd;
                float h = L.t_and(L.t_and(L.ge(ta, seg_lo), L.le(ta, seg_hi)), L.t_and(L.ge(tb, seg_lo), L.le(tb, seg_hi)));
                hit = L.t_or(hit, h);
            }
        }
    }
    float ok = L.lt(loss, tol);
    valid = L.t_and(L.t_and(on, L.t_not(hit)), ok);
    if (valid != valid) valid = 0.0f;  // jnp.nan_to_num
}

// ---- image method with runtime order (geometry.py:1013-1114) ----------------------------------------
__device__ __forceinline__ void image_solve(const ObjTables& T, int k, const int (&cd)[D2D_MAX_ORDER], float txx, float txy,
                                            float rxx, float rxy, float (&px)[NP], float (&py)[NP]) {
#pragma unroll
    for (int i = 0; i < NP; ++i) px[i] = py[i] = __builtin_nanf("");
    px[0] = txx;
    py[0] = txy;
    float imx[D2D_MAX_ORDER], imy[D2D_MAX_ORDER];
    float ix = txx, iy = txy;
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) {
        if (i < k) {
            float ox, oy;
            image_of(T.refl[2 * cd[i]], ix, iy, ox, oy);
            ix = ox;
            iy = oy;
            imx[i] = ix;
            imy[i] = iy;
        }
    }
    float ptx = rxx, pty = rxy;
#pragma unroll
    for (int i = D2D_MAX_ORDER - 1; i >= 0; --i) {
        if (i < k) {
            const float4 r0 = T.refl[2 * cd[i]];
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
}

// ---- MinPath / FermatPath: Adam on the parametric coordinates ------------------------------------------
// geometry.py:1117-1288, optimize.py:44-97 with optax.adam(0.1): mu = b1 mu + (1-b1) g; nu = b2 nu + (1-b2) g^2;
// x += -lr * (mu / (1 - b1^t)) / (sqrt(nu / (1 - b2^t)) + eps).  The gradient of the objective w.r.t. theta is
// derived by hand (reverse mode through parametric_to_cartesian and evaluate_cartesian / path_length).
struct AdamCfg {
    int solver;  // D2D_SOLVER_MINPATH or D2D_SOLVER_FERMAT
    int steps;
    int many;    // number of random starts; the start whose recorded loss is smallest wins (optimize.py:136-182)
    const float* __restrict__ bc1;  // [steps] 1 - b1^t
    const float* __restrict__ bc2;  // [steps] 1 - b2^t
    float lr, b1, b2, eps;
    float omb1, omb2;  // 1 - b1, 1 - b2 evaluated in double precision and rounded (optax: Python floats, weakly typed)
};

__device__ __forceinline__ void theta_to_points(const ObjTables& T, int k, const int (&cd)[D2D_MAX_ORDER],
                                                const float (&theta)[D2D_MAX_ORDER], float txx, float txy, float rxx, float rxy,
                                                float (&px)[NP], float (&py)[NP]) {
    // parametric_to_cartesian, geometry.py:988-1010 / 581-587 / 381-385
#pragma unroll
    for (int i = 0; i < NP; ++i) px[i] = py[i] = __builtin_nanf("");
    px[0] = txx;
    py[0] = txy;
    int j = 0;
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) {
        if (i < k) {
            const float4 r0 = T.refl[2 * cd[i]];
            const float4 r1 = T.refl[2 * cd[i] + 1];
            float x = r0.x, y = r0.y;  // Vertex: both rows hold the point, t = 0
            if (T.kind[cd[i]] != D2D_VERTEX) {
                float th = 0.0f;
#pragma unroll
                for (int q = 0; q < D2D_MAX_ORDER; ++q)
                    if (q == j) th = theta[q];
                x = r0.x + th * r1.x;
                y = r0.y + th * r1.y;
                ++j;
            }
#pragma unroll
            for (int q = 0; q < NP; ++q)
                if (q == i + 1) {
                    px[q] = x;
                    py[q] = y;
                }
        }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q)
        if (q == k + 1) {
            px[q] = rxx;
            py[q] = rxy;
        }
}

// objective and its gradient w.r.t. the interaction points (pbx, pby), then w.r.t. theta
__device__ __forceinline__ float objective_grad(const ObjTables& T, int solver, int k, const int (&cd)[D2D_MAX_ORDER],
                                                const float (&px)[NP], const float (&py)[NP], float (&gth)[D2D_MAX_ORDER]) {
    float pbx[NP], pby[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) pbx[i] = pby[i] = 0.0f;
    float loss = 0.0f;
    if (solver == D2D_SOLVER_FERMAT) {
#pragma unroll
        for (int i = 0; i <= D2D_MAX_ORDER; ++i) {
            if (i <= k) {
                float wx = (px[i + 1] - px[i]) + D2D_EPS, wy = (py[i + 1] - py[i]) + D2D_EPS;
                float len = sqrtf(wx * wx + wy * wy);
                loss = loss + len;
                float gx = wx / len, gy = wy / len;
                pbx[i + 1] += gx; pby[i + 1] += gy;
                pbx[i] -= gx; pby[i] -= gy;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < D2D_MAX_ORDER; ++i) {
            if (i < k) {
                const int kd = T.kind[cd[i]];
                const float4 r0 = T.refl[2 * cd[i]];
                float ev = 0.0f;
                if (kd == D2D_WALL) {
                    float v1x = px[i + 1] - px[i], v1y = py[i + 1] - py[i];
                    float v2x = px[i + 2] - px[i + 1], v2y = py[i + 2] - py[i + 1];
                    float ix_, iy_, rx_, ry_;
                    normalize2(v1x, v1y, ix_, iy_);
                    normalize2(v2x, v2y, rx_, ry_);
                    float din = ix_ * r0.z + iy_ * r0.w;
                    float s2 = 2.0f * din;
                    float ex = rx_ - (ix_ - s2 * r0.z);
                    float ey = ry_ - (iy_ - s2 * r0.w);
                    ev = ex * ex + ey * ey;
                    float ebx = 2.0f * ex, eby = 2.0f * ey;
                    float dinb = 2.0f * (ebx * r0.z + eby * r0.w);
                    float ibx = -ebx + dinb * r0.z, iby = -eby + dinb * r0.w;
                    float a1x, a1y, a2x, a2y;
                    normalize2_bwd(v1x, v1y, ibx, iby, a1x, a1y);
                    normalize2_bwd(v2x, v2y, ebx, eby, a2x, a2y);
                    pbx[i + 1] += a1x - a2x; pby[i + 1] += a1y - a2y;
                    pbx[i] -= a1x; pby[i] -= a1y;
                    pbx[i + 2] += a2x; pby[i + 2] += a2y;
                } else if (kd == D2D_RIS) {
                    float v2x = px[i + 2] - px[i + 1], v2y = py[i + 2] - py[i + 1];
                    float rx_, ry_;
                    normalize2(v2x, v2y, rx_, ry_);
                    float mx = -rx_, my = -ry_;
                    float sin_a = mx * r0.w - my * r0.z;
                    float cos_a = mx * r0.z + my * r0.w;
                    const float2 sc = T.sincos[cd[i]];
                    float ds = sin_a - sc.x, dc = cos_a - sc.y;
                    ev = ds * ds + dc * dc;
                    float sb = 2.0f * ds, cb = 2.0f * dc;
                    float mbx = sb * r0.w + cb * r0.z, mby = -sb * r0.z + cb * r0.w;
                    float a2x, a2y;
                    normalize2_bwd(v2x, v2y, -mbx, -mby, a2x, a2y);
                    pbx[i + 2] += a2x; pby[i + 2] += a2y;
                    pbx[i + 1] -= a2x; pby[i + 1] -= a2y;
                }
                loss = loss + ev;
            }
        }
    }
    // d/d theta_j = t_j . pbar_j
    int j = 0;
#pragma unroll
    for (int q = 0; q < D2D_MAX_ORDER; ++q) gth[q] = 0.0f;
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) {
        if (i < k && T.kind[cd[i]] != D2D_VERTEX) {
            const float4 r1 = T.refl[2 * cd[i] + 1];
            float gval = r1.x * pbx[i + 1] + r1.y * pby[i + 1];
#pragma unroll
            for (int q = 0; q < D2D_MAX_ORDER; ++q)
                if (q == j) gth[q] = gval;
            ++j;
        }
    }
    return loss;
}

// One Adam run from theta0: final theta in `th`, returns the objective recorded at the last step (before the last update).
__device__ __forceinline__ float opt_run(const ObjTables& T, const AdamCfg& A, int k, const int (&cd)[D2D_MAX_ORDER],
                                         const float* __restrict__ theta0, float txx, float txy, float rxx, float rxy,
                                         float (&th)[D2D_MAX_ORDER]) {
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
                mu[q] = A.b1 * mu[q] + A.omb1 * g[q];
                nu[q] = A.b2 * nu[q] + A.omb2 * (g[q] * g[q]);
                float mh = mu[q] / c1, nh = nu[q] / c2;
                th[q] = th[q] + (-A.lr) * (mh / (sqrtf(nh) + A.eps));
            }
        }
    }
    return last;
}

// Returns the path points and the loss the reference attaches to the path. theta0: [many][D2D_MAX_ORDER].
__device__ __forceinline__ float opt_solve(const ObjTables& T, const AdamCfg& A, int k, const int (&cd)[D2D_MAX_ORDER],
                                           const float* __restrict__ theta0, float txx, float txy, float rxx, float rxy,
                                           float (&px)[NP], float (&py)[NP]) {
    float best[D2D_MAX_ORDER], th[D2D_MAX_ORDER];
    float best_loss = opt_run(T, A, k, cd, theta0, txx, txy, rxx, rxy, best);
    for (int m = 1; m < A.many; ++m) {  // jnp.argmin: the first minimum wins; NaN losses win like in jnp.argmin
        float l = opt_run(T, A, k, cd, theta0 + m * D2D_MAX_ORDER, txx, txy, rxx, rxy, th);
        const bool better = (l < best_loss) || (l != l && best_loss == best_loss);
        best_loss = better ? l : best_loss;
#pragma unroll
        for (int q = 0; q < D2D_MAX_ORDER; ++q) best[q] = better ? th[q] : best[q];
    }
    theta_to_points(T, k, cd, best, txx, txy, rxx, rxy, px, py);
    if (A.solver == D2D_SOLVER_FERMAT) return interaction_loss(T, k, cd, px, py);  // geometry.py:1204
    return best_loss;                                                               // geometry.py:1284-1288
}

struct TraceArgs {
    ObjTables T;
    AdamCfg A;
    int solver;
    const int* __restrict__ cand;   // [C][D2D_MAX_ORDER]
    const int* __restrict__ order;  // [C]
    const float* __restrict__ theta0;  // [C][D2D_MAX_ORDER] initial guesses of the optimiser solvers, or null
    int C;
    const float* __restrict__ tx;  // [P][2]
    const float* __restrict__ rx;  // [P][2]
    int P;
    const float* __restrict__ xys_in;   // [P][C][NP][2] or null: validate these paths instead of solving
    const float* __restrict__ loss_in;  // [P][C] or null
    float* __restrict__ xys;    // [P][C][NP][2]
    float* __restrict__ loss;   // [P][C]
    float* __restrict__ valid;  // [P][C]   final is_valid
    float* __restrict__ on;     // [P][C]   on_objects            (may be null)
    float* __restrict__ hit;    // [P][C]   intersects_with_objects (may be null)
    float* __restrict__ length; // [P][C]   path_length           (may be null)
    int mode;  // MODE_*
    float alpha, tol, seg_lo, seg_hi;
};

#ifdef D2D_AUX_KERNELS  // non-template kernels: defined once, in d2d.hip
// One thread per (tx/rx pair, candidate).  Serves Scene.all_paths / all_valid_paths / accumulate_over_paths
// (scene.py:1156-1334), {Image,Min,Fermat}Path.from_tx_objects_rx and Path.is_valid / on_objects /
// intersects_with_objects (geometry.py:821-963) of the host mirror.
__global__ void __launch_bounds__(64) trace_kernel(TraceArgs a) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long)a.P * a.C) return;
    const int p = (int)(tid / a.C), c = (int)(tid % a.C);
    const int k = a.order[c];
    int cd[D2D_MAX_ORDER];
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) cd[i] = a.cand[c * D2D_MAX_ORDER + i];
    float px[NP], py[NP];
    const float txx = a.tx[2 * p], txy = a.tx[2 * p + 1], rxx = a.rx[2 * p], rxy = a.rx[2 * p + 1];
    const Truth L{a.mode, a.alpha};
    float loss = 0.0f;
    if (a.xys_in) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            px[i] = a.xys_in[(tid * NP + i) * 2];
            py[i] = a.xys_in[(tid * NP + i) * 2 + 1];
        }
        loss = a.loss_in ? a.loss_in[tid] : 0.0f;
    } else if (a.solver == D2D_SOLVER_IMAGE) {
        image_solve(a.T, k, cd, txx, txy, rxx, rxy, px, py);
        loss = interaction_loss(a.T, k, cd, px, py);
    } else if (k == 0) {
        image_solve(a.T, 0, cd, txx, txy, rxx, rxy, px, py);  // geometry.py:1178-1180 / 1268-1270
    } else {
        loss = opt_solve(a.T, a.A, k, cd, a.theta0 + (long)c * a.A.many * D2D_MAX_ORDER, txx, txy, rxx, rxy, px, py);
    }
    float on, hit, valid;
    literal_validity(a.T, L, k, cd, px, py, loss, a.tol, a.seg_lo, a.seg_hi, on, hit, valid);
    const float r = literal_length(k, px, py);
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

#endif  // D2D_AUX_KERNELS

// Grid sweep for the optimiser-based solvers: one RX cell per lane, candidates walked in the reference's order
// (scene.py:1892-1918); theta0 is per candidate and shared by every cell, as in the reference (scene.py:1887-1890).
struct OptSweepArgs {
    ObjTables T;
    AdamCfg A;
    const int* __restrict__ cand;      // [C][D2D_MAX_ORDER]   (explicit list: these sweeps have few candidates)
    const int* __restrict__ order;     // [C]
    const float* __restrict__ theta0;  // [C][D2D_MAX_ORDER]
    int C;
    const float* __restrict__ X;
    const float* __restrict__ Y;
    float* __restrict__ out;
    long cells;
    float txx, txy;  // the fixed end point (transmitter for an RX grid, receiver for a TX grid)
    int grid_is_tx;
    int mode;
    float alpha, tol, seg_lo, seg_hi;
    float fnum[D2D_MAX_ORDER + 1];
    float h2;
    int fun_id;
    int out_mode;
    const float* __restrict__ cust_f;   // [C][cells]                          D2D_FUN_CUSTOM (the reverse-mode value+grad sweep only)
    const float* __restrict__ cust_pb;  // [C][cells][D2D_MAX_ORDER + 2][2]
};

// valid * fun of one (cell, candidate): solve, validate, evaluate (scene.py:1892-1918 with an optimiser-based path class)
__device__ __forceinline__ float opt_contribution(const OptSweepArgs& a, int c, float txx, float txy, float rxx, float rxy) {
    const Truth L{a.mode, a.alpha};
    const int k = a.order[c];
    int cd[D2D_MAX_ORDER];
#pragma unroll
    for (int i = 0; i < D2D_MAX_ORDER; ++i) cd[i] = a.cand[c * D2D_MAX_ORDER + i];
    const float* th0 = a.theta0 + (long)c * a.A.many * D2D_MAX_ORDER;
    float px[NP], py[NP];
    float loss = 0.0f;
    if (k == 0) image_solve(a.T, 0, cd, txx, txy, rxx, rxy, px, py);
    else loss = opt_solve(a.T, a.A, k, cd, th0, txx, txy, rxx, rxy, px, py);
    float on, hit, valid;
    literal_validity(a.T, L, k, cd, px, py, loss, a.tol, a.seg_lo, a.seg_hi, on, hit, valid);
    const float r = literal_length(k, px, py);
    float f;
    if (a.fun_id == D2D_FUN_RECEIVED_POWER) {
        float num = a.fnum[0];
#pragma unroll
        for (int q = 1; q <= D2D_MAX_ORDER; ++q)
            if (q == k) num = a.fnum[q];
        f = num / (a.h2 + r * r);
    } else if (a.fun_id == D2D_FUN_LENGTH_SQUARED) f = r * r;
    else if (a.fun_id == D2D_FUN_LENGTH) f = r;
    else f = 1.0f;
    return valid * f;
}

#ifdef D2D_AUX_KERNELS
// One cell per lane, the candidates one after the other (any number of candidates).
__global__ void __launch_bounds__(64) power_opt_kernel(OptSweepArgs a) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.cells) return;
    const float gx_ = a.X[idx], gy_ = a.Y[idx];
    const float txx = a.grid_is_tx ? gx_ : a.txx, txy = a.grid_is_tx ? gy_ : a.txy;
    const float rxx = a.grid_is_tx ? a.txx : gx_, rxy = a.grid_is_tx ? a.txy : gy_;
    float acc = 0.0f;
    for (int c = 0; c < a.C; ++c) acc = acc + opt_contribution(a, c, txx, txy, rxx, rxy);
    if (a.out_mode == D2D_OUT_ADD) a.out[idx] = a.out[idx] + acc;
    else a.out[idx] = acc;
}

// The same with the candidates spread over blockIdx.y: every (cell, candidate) is `steps` SEQUENTIAL Adam iterations, so
// a 300^2 grid with 7 candidates only fills the chip when the candidates run side by side (1.4 -> 10 waves per SIMD).
// Contributions go to contrib[c][cell]; opt_reduce_kernel adds them in candidate order: the same fp32 sum.
__global__ void __launch_bounds__(64) power_opt_cand_kernel(OptSweepArgs a, float* __restrict__ contrib) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.cells) return;
    const int c = blockIdx.y;
    const float gx_ = a.X[idx], gy_ = a.Y[idx];
    const float txx = a.grid_is_tx ? gx_ : a.txx, txy = a.grid_is_tx ? gy_ : a.txy;
    const float rxx = a.grid_is_tx ? a.txx : gx_, rxy = a.grid_is_tx ? a.txy : gy_;
    contrib[(long)c * a.cells + idx] = opt_contribution(a, c, txx, txy, rxx, rxy);
}

__global__ void __launch_bounds__(256) opt_reduce_kernel(const float* __restrict__ contrib, int C, long cells, float* __restrict__ out,
                                                         int out_mode) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= cells) return;
    float acc = 0.0f;  // scene.py:1893
    for (int c = 0; c < C; ++c) acc = acc + contrib[(long)c * cells + idx];
    if (out_mode == D2D_OUT_ADD) out[idx] = out[idx] + acc;
    else out[idx] = acc;
}

#endif  // D2D_AUX_KERNELS

}  // namespace d2d
