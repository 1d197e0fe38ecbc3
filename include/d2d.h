/*
 * d2d.h -- C ABI of libd2d.so, the MI355X (gfx950) drop-in for DiffeRT2d's hot path.
 *
 * DiffeRT2d (v0.4.0) is pure Python on JAX: it has no FFI, its extension points are Python
 * protocols.  This header is the boundary a maintainer would bind with ctypes/cffi from
 * differt2d/scene.py (see INTEGRATION.md for the stub).  Each entry point cites the
 * reference interface it replaces (paths relative to the DiffeRT2d checkout).
 *
 * Conventions
 *   - every function returns D2D_OK (0) or a negative d2d_status; nothing throws across the
 *     ABI; d2d_last_error() returns a thread-local, human-readable message for the last failure;
 *   - the caller owns all host buffers (C-contiguous fp32 / int32 / uint8); the library owns
 *     every device buffer inside d2d_ctx; one ctx = one GPU + one HIP stream; calls on one ctx
 *     are serialised by the caller, different ctxs may be driven from different threads;
 *   - all floating point is IEEE fp32 with one rounding per operation (no fma contraction,
 *     correctly rounded divide/sqrt), operations in the order the reference writes them;
 *   - there is NO CPU fallback: without a usable gfx950 device d2d_create fails.
 */
#ifndef D2D_H
#define D2D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D2D_MAX_ORDER 4 /* highest interaction order a sweep accepts */
#define D2D_ABI_VERSION 10

typedef enum d2d_status {
    D2D_OK = 0,
    D2D_ERR_INVALID = -1,     /* bad argument (NULL, negative size, order > D2D_MAX_ORDER, alpha <= 0 ...) */
    D2D_ERR_HIP = -2,         /* a HIP runtime call failed */
    D2D_ERR_NO_DEVICE = -3,   /* no gfx950 device visible */
    D2D_ERR_UNSUPPORTED = -4, /* feature outside the native closed set (e.g. ImagePath with RIS objects) */
    D2D_ERR_STATE = -5,       /* call order violated (e.g. sweep before d2d_set_scene / d2d_set_grid) */
    D2D_ERR_COMM = -6         /* RCCL failure */
} d2d_status;

/* Object kinds, differt2d/geometry.py:542-721 (Wall), :683-721 (RIS), :352-431 (Vertex). */
enum { D2D_WALL = 0, D2D_RIS = 1, D2D_VERTEX = 2 };

/* Path solver = `path_cls`, differt2d/scene.py:1814 (ImagePath geometry.py:1013-1114,
 * MinPath :1207-1288, FermatPath :1117-1204). */
enum { D2D_SOLVER_IMAGE = 0, D2D_SOLVER_MINPATH = 1, D2D_SOLVER_FERMAT = 2 };

/* Activation = `function` kwarg of differt2d/logic.py:258-312. */
enum { D2D_ACT_HARD_SIGMOID = 0, D2D_ACT_SIGMOID = 1 };

/* Natively fused `fun` (PathFun, differt2d/scene.py:51):
 *   RECEIVED_POWER  differt2d/utils.py:17-54        r_coef**n / (height^2 + length^2)
 *   LENGTH_SQUARED  tests/test_scene.py:444-445     path.length() ** 2
 *   LENGTH          differt2d/geometry.py:811-819   path.length()
 *   ONE             1.0 (the map then counts valid paths -- "intersection counts") */
enum {
    D2D_FUN_RECEIVED_POWER = 0,
    D2D_FUN_LENGTH_SQUARED = 1,
    D2D_FUN_LENGTH = 2,
    D2D_FUN_ONE = 3,
    D2D_FUN_CUSTOM = 4 /* values and derivatives supplied per (candidate, cell): d2d_set_path_fun_values; value+grad launches only */
};

/* Which end of the paths the grid cells are. */
enum { D2D_GRID_RX = 0, D2D_GRID_TX = 1 };

/* How a sweep combines with what the output map already holds. */
enum {
    D2D_OUT_OVERWRITE = 0, /* Z  = facc                                              */
    D2D_OUT_ADD = 1        /* Z  = Z + facc   (reduce_all over transmitters, differt2d/scene.py:1939-1952) */
};

/* Everything Scene.accumulate_on_receivers_grid_over_paths threads down to Path.is_valid and
 * `fun` (differt2d/scene.py:1803-1826; kwargs -> differt2d/geometry.py:910-919, logic.py:258-267). */
typedef struct d2d_params {
    int32_t min_order;  /* differt2d/scene.py:1817 */
    int32_t max_order;  /* differt2d/scene.py:1818 */
    int32_t approx;     /* 0 = jnp.logical_*, 1 = min/max/activation (differt2d/logic.py:315-537) */
    int32_t act;        /* D2D_ACT_* */
    float alpha;        /* differt2d/defaults.py:3 (100.0) */
    float tol;          /* differt2d/geometry.py:915 (1e-2): loss tolerance of is_valid */
    float patch;        /* differt2d/geometry.py:916 / defaults.py:7 (0.0) */
    float seg_tol;      /* differt2d/geometry.py:89 (0.005); not reachable from the reference's sweep kwargs */
    int32_t fun_id;     /* D2D_FUN_* */
    float r_coef;       /* differt2d/defaults.py:12 (0.5) */
    float height;       /* differt2d/defaults.py:15 (0.1) */
    int32_t solver;     /* D2D_SOLVER_* */
    int32_t steps;      /* differt2d/optimize.py:50 (100): Adam steps of MinPath / FermatPath */
    int32_t out_mode;   /* D2D_OUT_* */
    int32_t grid_role;  /* D2D_GRID_RX: the grid cells are receivers and `tx` is the transmitter
                           (accumulate_on_receivers_grid_over_paths, differt2d/scene.py:1803-1953);
                           D2D_GRID_TX: the grid cells are transmitters and the `tx` argument of the launch is the
                           fixed RECEIVER (accumulate_on_transmitters_grid_over_paths, differt2d/scene.py:1489-1648);
                           the per-cell gradient is then taken w.r.t. the transmitter (scene.py:1617-1620).
                           TX grids run the culled kernels only while a path with a zero-length segment (a step of the backward
                           scan with un == 0, loss >= 0.999) is EXACTLY invalid under `tol` / `alpha` / the activation: hard
                           tol <= 0.5; hard_sigmoid alpha (tol - 0.999) + 3 <= 0; sigmoid alpha (tol - 0.999) <= -89.5 (e.g.
                           NOT sigmoid with alpha = 10).  Otherwise every (cell, candidate) is evaluated -- correct, tens of
                           times slower, and counted by d2d_debug_txg_fallbacks */
    int32_t strict_nan; /* value+grad sweeps only.  The reference's reverse-mode autodiff returns NaN for a cell whenever the
                           backward scan of ANY candidate -- valid or not -- hits un == 0 (differt2d/geometry.py:1105) or, in
                           the approx modes, a zero-length segment (normalize, :227-228).  0 (default): candidates that tile
                           culling proves invalid are not evaluated; a separate NaN scan (conservative test per 8 x 8 patch and
                           candidate, then the backward scan itself for the few survivors) finds every such cell and poisons
                           the gradient map and the scene VJP exactly as the exhaustive evaluation does.  1: every candidate of
                           every cell is evaluated in full (about 40x slower; the cross-check the tests hold 0 against) */
    int32_t many;       /* differt2d/optimize.py:142: random starts per candidate of MinPath / FermatPath, best recorded loss
                           wins (0 or 1 = one start; the path classes default to 1, differt2d/geometry.py:1198, 1282) */
    int32_t reserved[1];
} d2d_params;

typedef struct d2d_ctx d2d_ctx;

/* ---- library / device ------------------------------------------------------------------ */

int d2d_abi_version(void);
const char* d2d_last_error(void);
/* Number of visible HIP devices (0 is not an error). */
int d2d_device_count(int* count);
/* Fills name (<= cap bytes), number of CUs and total global memory of a device. */
int d2d_device_info(int device, char* name, int cap, int* cus, int64_t* mem_bytes);

/* One context per GPU. Fails with D2D_ERR_NO_DEVICE when `device` does not exist. */
int d2d_create(int device, d2d_ctx** ctx);
void d2d_destroy(d2d_ctx* ctx);
int d2d_synchronize(d2d_ctx* ctx);

/* ---- scene (replaces Scene.objects / Scene.from_walls_array, differt2d/scene.py:191, 413-426) -- */

/* objects: xys[N][2][2] (origin, dest; a Vertex stores its point in both rows), kind[N]
 * (D2D_WALL / D2D_RIS / D2D_VERTEX, NULL = all walls), phi[N] (RIS angle, NULL = pi/4).
 * Size: the image-method sweeps keep per-object tables in LDS -- up to ~1 300 objects in 64 KB with several workgroups per CU,
 * up to ~2 400 with one workgroup per CU (156 of gfx950's 160 KB); a sweep over a bigger scene returns D2D_ERR_UNSUPPORTED.
 * Scenes above 4 095 objects sweep without region lists (12-bit object indices in the list entries). */
int d2d_set_scene(d2d_ctx* ctx, const float* xys, const uint8_t* kind, const float* phi, int32_t n_objects);
/* (A scene that is resident already -- bit-identical arguments -- is recognised: nothing is uploaded and the scene-only
 * masks and the schedule's work history stay valid; the candidate mask is reset to "all" as always.) */

/* filter_objects of Scene.all_path_candidates (differt2d/scene.py:1089-1134): allowed[i] != 0
 * means object i may appear in a path candidate; NULL = all. Filtered objects still occlude. */
int d2d_set_candidate_mask(d2d_ctx* ctx, const uint8_t* allowed);

/* Context-free candidate enumeration (pure host integer code, replaces the Rust
 * differt_core.rt.CompleteGraph / DiGraph.all_paths calls at differt2d/scene.py:153-175): for each order
 * k ascending, every tuple of k allowed object indices with no two equal neighbours, lexicographic
 * (recorded order: docs/source/notebooks/cost20120_helsinki_model.ipynb cell 20). cand is row-major
 * [count][D2D_MAX_ORDER], -1 padded; either output pointer may be NULL. */
int d2d_count_candidates(int32_t n_objects, const uint8_t* allowed, int32_t min_order, int32_t max_order, int64_t* count);
int d2d_enumerate_candidates(int32_t n_objects, const uint8_t* allowed, int32_t min_order, int32_t max_order, int32_t* cand,
                             int32_t* order, int64_t capacity);

/* Number of path candidates sum_k |{tuples of length k, no equal neighbours}| for the current
 * scene and mask (differt2d/scene.py:122-175; differt-core 0.0.31 CompleteGraph.all_paths). */
int d2d_num_candidates(d2d_ctx* ctx, int32_t min_order, int32_t max_order, int64_t* count);

/* Writes the candidate list itself (row-major [count][D2D_MAX_ORDER], -1 padded) and each
 * candidate's order, in the reference's enumeration order. Either pointer may be NULL. */
int d2d_list_candidates(d2d_ctx* ctx, int32_t min_order, int32_t max_order, int32_t* cand, int32_t* order,
                        int64_t capacity);

/* ---- grid sweep (replaces Scene.accumulate_on_receivers_grid_over_paths, differt2d/scene.py:1803-1953,
 *      for one transmitter at a time; X, Y as produced by Plottable.grid, differt2d/abc.py:57-81) -------- */

/* Uploads the receiver grid (row-major [m][n], any coordinates) and (re)allocates the resident
 * output maps; the value map is zeroed.  A grid that is resident already (same m, n and bit patterns, compared byte for
 * byte with a host copy the context keeps: 8 bytes of host memory per cell) is not uploaded again and keeps everything
 * keyed to it (the regions' bounding boxes, the work history of the patch schedule): the reference's callers pass X, Y with
 * every call (differt2d/scene.py:1803-1826). */
int d2d_set_grid(d2d_ctx* ctx, const float* X, const float* Y, int32_t m, int32_t n);
/* The same with a caller-supplied version token instead of the content hash: version != 0 asserts that equal versions
 * mean equal contents (an immutable array the caller has passed before, like the reference's JAX arrays); 0 = hash.  A strided
 * sample of ~500 cells of both arrays is compared beside the token (an array whose read-only flag was flipped, written through
 * and flipped back has the same token: a change that misses every sampled cell is not seen). */
int d2d_set_grid_versioned(d2d_ctx* ctx, const float* X, const float* Y, int32_t m, int32_t n, uint64_t version);
/* Diagnostic: how many d2d_set_grid / d2d_set_grid_versioned calls found their grid resident already. */
int d2d_debug_grid_reuses(d2d_ctx* ctx, int64_t* count);
/* Diagnostic: the launch shape of the last RX-grid value sweep -- waves_per_patch = 1 (one wave per patch, the dearest
 * patches cut in parts), 4 (patches shared by 4 waves prefix by prefix) or, with candidates = 1, the 4 / 8 / 16 waves of
 * the kernel that shares a patch candidate by candidate (options coop_waves, coop_max_tiles); 0 before any sweep. */
int d2d_debug_sweep_shape(d2d_ctx* ctx, int32_t* waves_per_patch, int32_t* candidates);
/* Diagnostic: how many TX-grid sweeps of this context ran the EXHAUSTIVE kernel although culling was asked for (no
 * strict_nan, no "txg_exhaustive" option), because a degenerate path is not exactly invalid under their parameters (see
 * d2d_params.grid_role) -- so that a bench or a profile does not misattribute that time. */
int d2d_debug_txg_fallbacks(d2d_ctx* ctx, int64_t* count);
/* Diagnostic: how often the last-segment masks of the leaf regions ("hidden_masks" option) were built, and whether the
 * context holds valid ones now. */
int d2d_debug_hidden_masks(d2d_ctx* ctx, int64_t* builds, int32_t* valid);

/* Initial guesses of the optimiser-based solvers for the NEXT sweeps: theta0[n_candidates * many][D2D_MAX_ORDER]
 * (n_rows = n_candidates * max(1, params->many)), candidates in enumeration order, the `many` starts of one candidate
 * consecutive (unused entries ignored); every RX cell starts from the same guesses, as in the reference
 * (differt2d/scene.py:1887-1890, optimize.py:132, 174-178). */
int d2d_set_theta0(d2d_ctx* ctx, const float* theta0, int64_t n_rows);

/* The optimiser of the MinPath / FermatPath solvers for the NEXT sweeps and traces (differt2d/optimize.py:44-51: any
 * optax.GradientTransformation; default optax.adam(learning_rate=0.1), optimize.py:83).  Native: Adam with any
 * hyper-parameters (optax.adam(learning_rate, b1, b2, eps), eps_root = 0) -- they are Python floats on the reference's
 * side, hence doubles here; the gradients through the solver (reverse mode and forward tangents) follow them.  Any other
 * kind: D2D_ERR_UNSUPPORTED. */
#define D2D_OPT_ADAM 0
int d2d_set_optimizer(d2d_ctx* ctx, int32_t kind, double learning_rate, double b1, double b2, double eps);

/* Launches the fused forward sweep for transmitter tx[2] on the ctx stream (asynchronous).
 * Inputs and outputs stay resident in HBM. params->solver selects ImagePath (fused image-method kernel) or
 * MinPath / FermatPath (per-cell Adam loop of params->steps iterations, hand-derived gradient). */
int d2d_power_map_launch(d2d_ctx* ctx, const d2d_params* params, const float* tx);

/* ---- value + gradient (replaces grad=True / value_and_grad=True of the sweep, differt2d/scene.py:1920-1923,
 *      and the user-side jax.value_and_grad over scene parameters, examples/plot_power_optimize.py:78-93) ---- */

/* Cotangent of the value map for the scene-parameter VJP: cot[m*n], or NULL for all ones (the gradient of
 * sum(Z)). Reset by d2d_set_grid. */
int d2d_set_cotangent(d2d_ctx* ctx, const float* cot);

/* Gradient of a sweep whose path function `fun` is an arbitrary host callable (the reference differentiates any JAX
 * callable, differt2d/scene.py:1892-1923: d/d cell of sum_c valid_c * fun_c).  The host traces the paths
 * (d2d_trace_paths), evaluates fun and its derivative on them and hands both over:
 *   f[n_candidates][m*n]                          fun of (candidate, cell),
 *   xys_bar[n_candidates][m*n][D2D_MAX_ORDER+2][2] d fun / d path points (rows 0 .. order+1 of a candidate are read; a
 *                                                 derivative w.r.t. tx.xy / rx.xy as arguments of fun is added to rows 0 / order+1),
 * candidates in the sweep's order (orders min_order..max_order, lexicographic over the allowed objects -- the order of
 * d2d_trace_paths' candidate list built by the reference's all_path_candidates).  A following d2d_power_map_vg_launch with
 * params->fun_id == D2D_FUN_CUSTOM (image solver: every candidate of every cell is evaluated, as under strict_nan;
 * MinPath / FermatPath: the reverse pass over the stored trajectory, option "opt_grad_mode" 0, with the theta0 rows the
 * paths were traced with) then
 * chains them through the hand-derived adjoint of the validity and of the path method: d2d_get_map returns
 * sum_c valid_c * f_c and d2d_get_grad_rx its derivative w.r.t. the cell, d2d_get_scene_vjp the pull-back to the fixed
 * end point and the wall end points THROUGH THE PATHS (fun's own dependence on the objects is the caller's).
 * Host arrays, copied before the call returns; NULL / 0 drops them; dropped by d2d_set_grid, by a different scene and by
 * a different candidate mask.  D2D_ERR_STATE from the launch
 * when n_candidates is not the number of candidates the sweep walks. */
int d2d_set_path_fun_values(d2d_ctx* ctx, const float* f, const float* xys_bar, int64_t n_candidates);

/* Value+grad sweep.  ImagePath: fused, hand-derived reverse mode.  MinPath / FermatPath (differt2d/geometry.py:1117-1288):
 * hand-derived reverse mode through the reference's lax.scan of Adam steps (differt2d/optimize.py:83-97): the solver writes its
 * trajectory to HBM and the steps are walked backwards (d2d_optrev.hpp; 16 bytes per step and unknown, grids whose
 * trajectories exceed "opt_traj_mb" are swept in chunks of cells), for Wall, RIS (incl. phi, geometry.py:698-711) and Vertex
 * objects; "opt_grad_mode" = 1 selects the older forward-tangent kernel (d2d_optgrad.hpp), kept as an independent
 * cross-check.  Writes the value map exactly as
 * d2d_power_map_launch does, the per-cell gradient d Z[i,j] / d rx[i,j] (resident, [m][n][2]) and, when
 * want_scene_vjp != 0, the VJP of the map w.r.t. the transmitter position and every object end point,
 * contracted with the cotangent. out_mode D2D_OUT_ADD accumulates all of them (reduce_all). Asynchronous. */
int d2d_power_map_vg_launch(d2d_ctx* ctx, const d2d_params* params, const float* tx, int32_t want_scene_vjp);

/* Synchronises and copies the per-cell gradient map to out[m*n*2] (last axis d/dx, d/dy). */
int d2d_get_grad_rx(d2d_ctx* ctx, float* out);

/* Synchronises and copies the scene-parameter VJP: tx_bar[2] (the fixed end point), xys_bar[N][2][2] (object end points;
 * may be NULL) and phi_bar[N] (RIS angles, differt2d/geometry.py:683-721; may be NULL; identically 0 after an ImagePath
 * sweep, whose candidates hold Wall objects only). */
int d2d_get_scene_vjp(d2d_ctx* ctx, float* tx_bar, float* xys_bar, float* phi_bar);

/* Same sweep through the instrumented build of the kernel (same results, not for timing): fills
 * stats[D2D_NUM_STATS] with executed-work counters summed over waves (one count = one 64-lane wave):
 *   [0] candidates evaluated (interaction points + on_objects)   [1] ... that reached the loss stage
 *   [2] ... that reached the occlusion loop                      [3] ... with a non-zero validity (valid * fun evaluated)
 *   [4] segment/wall tests evaluated                             [5] tests that took the exact-divide path
 *   [6] sum of k over [0]    [7] sum of k over the candidates whose loss was evaluated exactly    [8] sum of (k+1) over [3]
 *   [9] tile-culling levels evaluated (one count = 64 candidates x 4 vertex evaluations)
 *   [10..14] shader-clock ticks summed over waves: patch prologue, order 0, order 1, order 2, exact evaluation of survivors
 * bench.py prices these with SURVEY.md section 8(d)'s per-unit FLOP figures. */
#define D2D_NUM_STATS 16
int d2d_power_map_stats(d2d_ctx* ctx, const d2d_params* params, const float* tx, uint64_t* stats);

/* Diagnostic (instrumented build): shader-clock ticks each 64-lane wave (= one 8 x 8 patch of cells, row-major over
 * patches) spent in the sweep -- the load-balance picture behind the roofline numbers. */
int d2d_power_map_wave_cycles(d2d_ctx* ctx, const d2d_params* params, const float* tx, uint64_t* cycles, int64_t capacity,
                              int64_t* n_waves);

/* Launch-shape tuning of a context; never changes a result bit (tests sweep these to cover every kernel variant).
 *   "split_max_tiles": launches of at most this many 8 x 8 patches share every patch between 4 waves (0 = never; default -1:
 *                  5120 with hard validity, never with hard_sigmoid -- those share candidate by candidate, "coop_waves", or
 *                  run one wave per patch; never with sigmoid validity unless "split_sigmoid" is non-zero)
 *   "coop_waves": the smallest launches share every patch between this many waves CANDIDATE BY CANDIDATE, wave 0 adding the
 *                  contributions in the reference's order (4, 8 or 16; 0 = never; default -1: by the launch's size and
 *                  validity mode -- 16 waves up to 256 patches and 8 up to 640 (hard) / 2304 (hard_sigmoid), 16 / 8 / 4 up
 *                  to 640 / 1600 / 4096 patches with sigmoid validity, and never more than "split_max_tiles");
 *                  "coop_max_tiles": replaces the upper limit (-1)
 *   "hidden_masks": non-zero (default) = forward RX-grid sweeps with region lists keep, per leaf region and wall, a 64-bin mask
 *                  of where the wall is certainly hidden from the WHOLE region (scene, grid and validity mode only: built
 *                  by the second launch in a row that would use it, kept until the scene, the grid or the mode changes);
 *                  candidates whose last interaction point can only lie in hidden bins leave the region's list, and order-1
 *                  candidates the patch's culling (same results); "hidden_min_tiles": only launches of at least this many
 *                  patches use them (default 400: smaller ones are latency-bound and lose more to the extra load than they gain)
 *   "prep_fused": non-zero (default) = the per-launch preparation runs as 4 kernels (masks of both kinds in one, the
 *                  schedule's histogram + sort in one); zero = the 7 separate kernels of round 2 (same results)
 *   "sched_min_tiles": launches of at least this many patches start their dearest patches first (default 2048)
 *   "heavy_split": with a work history and max_order == 2, this many of the dearest patches of a launch that is too big to
 *                  share every patch are cut in four parts swept by separate workgroups (0 = none, at most a quarter of the
 *                  patches; default -1: 3 patches in 32 when the launch holds fewer than 4 patches per wave slot of the
 *                  chip, else 64)
 *   "cost_history": non-zero (default) = a launch that sweeps the same grid as the previous one orders its patches by the
 *                   work each took then (counted by the kernels); zero = always by the geometric proxy
 *   "pair_masks": zero = do not build / use the wall-to-wall occlusion masks (A/B and tests; same results)
 *   "nan_scan": the pass behind a culled value+grad sweep that finds the reference's autodiff NaN cells (d2d_params.strict_nan):
 *                  1 (default) = one workgroup of 16 waves per region of 4 x 4 patches, 2 = one wave per patch (same flags),
 *                  0 = off (round 3's behaviour: NaN only inside the candidates the sweep evaluates; A/B);
 *                  "nan_scan_stats": non-zero = count its work (d2d_debug_nan_scan; slows the scan down)
 *                  "nan_scan_async": non-zero (default) = the scan runs BESIDE the sweep on a stream of its own and leaves flags
 *                  that a small kernel applies once both are through; zero = behind the sweep on the sweep's stream (same results)
 *                  "nan_scan_prio": priority of that stream, 0 (default) = lowest, 1 = highest (A/B: the highest is slower)
 *                  "nan_scan_wqcap" / "nan_scan_rb": entries of a region's probe queue / batches per round of its list that the
 *                  two-level scan USES (0 = default: all it has; tests set them small so that a full queue and a full list are the
 *                  rule instead of a rare event -- same flags whatever the values).  Any of these two, or "nan_scan_stats", selects the
 *                  region kernel's debug instance (run-time sizes, counters through LDS); the product instance has neither
 *   "comm_prio": priority of the stream the RCCL collectives run on beside the next sweep: -1 lowest, 0 (default) normal, 1 highest;
 *                  set it BEFORE the first collective (D2D_ERR_STATE afterwards)
 *   "sig_narrow_filter": sigmoid validity, forward sweeps: 1 (default) = the divide-free filter of the occlusion tests drops what is
 *                  certainly below z = -17.5 (1 - sigmoid(z) is exactly 1.0f there), 0 = what is certainly below -89 (same results)
 *   "opt_parallel": zero = MinPath / FermatPath sweeps walk the candidates one after the other in every lane (same results)
 *   "opt_grad_mode": gradients through the MinPath / FermatPath solvers: 0 (default) reverse mode over the stored trajectory,
 *                   1 forward tangents carried through the loop (same derivative; NaN only where a local partial derivative
 *                   is infinite vs. also where an accumulated tangent overflowed); "opt_traj_mb": device memory the
 *                   trajectory store of the reverse mode may take (default 16384)
 *   "txg_exhaustive": non-zero = TX-grid value sweeps use the exhaustive kernel instead of the culled one (same results)
 *   "region_lists": zero = no region candidate lists: every patch enumerates the prefixes of its candidates itself
 *                   (default non-zero: culled RX-grid launches of max_order >= 2 build, per launch, for every region of
 *                   patches the list of candidates that the tile culling cannot drop for the region's bounding box, and
 *                   the patches only test and evaluate those; same results)
 *   "region_size" / "region_size_top": leaf regions are region_size x region_size patches (default 4, 1..64); their
 *                   lists are refined from those of regions of region_size_top patches a side (default 16, rounded down to
 *                   a multiple of region_size), which are built by enumeration
 *   "region_slices": lists per top region = slices of first walls = waves that enumerate it (default 0: a quarter of
 *                   the allowed objects); "region_budget_mb": device memory of ALL list pools (default 24576: an upper bound; the pipeline keeps one pool per
 *                   rotating set, three in all, so a pool may grow to a third of it); a list
 *                   that does not fit is marked as not listed and the patches of its region enumerate (same results)
 *   "sched_key_mode": schedule keys from 0 (default) the work history if there is one, else the lengths of the region
 *                   lists, else the geometric proxy; 1 never the history; 2 never the lists
 *   "pipeline": non-zero (default) = everything a launch rebuilds (shadow masks, region lists, the schedule's sort) lives
 *                   in three rotating sets and is built on side streams beside the previous launches' sweeps; the work
 *                   history a schedule is sorted by is then three launches old instead of one; zero = on the
 *                   launch's own stream, in front of its sweep (same results)
 *   "unpiped_max_tiles": launches of orders <= 1 over at most this many 8 x 8 patches (default 256) prepare on the sweep's own
 *                   stream whatever "pipeline" says: a small call is launch latency, and the side stream's fork and join cost it
 *                   8 us of 45 (same results); 0 = never
 *   "side_stream": zero = the schedule's sort is never moved to a stream of its own; "fwd_waves": patches per workgroup
 *                   of the sweep with region lists (0 default: 4 when the per-wall LDS table is big, else 1)
 *   "region_budget_mb" also bounds the growth of the list pool: it starts at 256 MB and is quadrupled (up to the budget,
 *                   a third of it per pool) when a launch's lists did not fit (read back without waiting); when the device cannot
 *                   provide the pool, the launch enumerates instead (same results) and later launches ask for a quarter
 *   "time_kernel": non-zero = bracket the sweep kernel of every launch with HIP events (see d2d_last_kernel_ms)
 * Also read once at d2d_create from the environment: D2D_SPLIT_MAX_TILES, D2D_SCHED_MIN_TILES. No reference counterpart
 * (XLA picks its own launch shapes). Returns D2D_ERR_INVALID for an unknown name. */
int d2d_set_option(d2d_ctx* ctx, const char* name, int64_t value);

/* Diagnostics of the patch schedule (the order in which the culled kernels start their 8 x 8 patches; it changes
 * speed, never results): d2d_debug_get_schedule copies the order built by the last scheduled launch and its cost keys
 * (n = number of patches); d2d_debug_set_schedule makes later launches of exactly n patches use `order` (a permutation
 * of 0..n-1) instead; n = 0 removes the override. */
int d2d_debug_set_schedule(d2d_ctx* ctx, const int32_t* order, int64_t n);
int d2d_debug_get_schedule(d2d_ctx* ctx, int32_t* order, uint8_t* key, int64_t n);
/* Region candidate lists of the last launch that built any (all zeros otherwise): out[0] pool chunks handed out,
 * [1] pool chunks available, [2] patches left to the enumerating kernel, [3] leaf regions with a list that is not listed,
 * [4] / [5] / [6] entries of the leaf lists of order 2 / 3 / 4, [7] leaf regions.  Waits for the stream. */
int d2d_debug_region_stats(d2d_ctx* ctx, int64_t* out /* [8] */);
/* Counters of the last value+grad launch's NaN scan (option "nan_scan_stats" = 1; zeros otherwise): out[0] (patch, candidate)
 * pairs whose backward scan was probed cell by cell, [1] cells flagged NaN, [2] patches with a flagged cell; two-level scan only:
 * [3] probes a wave made itself because its region's queue was full, [4] queue items refused because they named no patch of the
 * region or no object of the scene (an internal error: always 0), [5] rounds of the regions' lists.  Waits for the stream. */
int d2d_debug_nan_scan(d2d_ctx* ctx, int64_t* out /* [6] */);

/* The work history behind the schedule: what each of the n patches took in the last culled sweep (units of ~25
 * wave-instructions, counted by the kernels). */
int d2d_debug_get_work(d2d_ctx* ctx, uint32_t* work, int64_t n);

/* Duration of the sweep kernel proper of the last launch on this context -- without the preparation kernels in front
 * of it (shadow masks, patch schedule) and the VJP reduction behind it -- from HIP events recorded on the context's
 * stream; needs the "time_kernel" option. Waits for that kernel. This is the figure bench.py's roofline uses. */
int d2d_last_kernel_ms(d2d_ctx* ctx, float* ms);

/* Diagnostic: evaluates x[i] / y[i] on the GPU three ways -- q_fast: the kernels' bare fma chain on a refined
 * v_rcp; q_ref: the compiler's generic correctly rounded expansion; q_hostr: the bare chain on a host-computed
 * reciprocal. For operands in [2^-62, 2^62] (or x == 0) all three must be bit-identical. */
int d2d_selftest_div(d2d_ctx* ctx, const float* x, const float* y, int64_t n, float* q_fast, float* q_ref, float* q_hostr);

/* Diagnostic: the device's expf as the sigmoid activation uses it (the algorithm of glibc's expf, in double, rounded once:
 * d2d_kernels.hpp expf_libm) on x[n] -> y[n]; must equal the host C library's expf bit for bit. */
int d2d_selftest_expf(d2d_ctx* ctx, const float* x, int64_t n, float* y);

/* Synchronises and copies the resident value map to out[m*n]. */
int d2d_get_map(d2d_ctx* ctx, float* out);

/* Convenience: d2d_set_grid + d2d_power_map_launch + d2d_get_map. */
int d2d_power_map(d2d_ctx* ctx, const d2d_params* params, const float* tx, const float* X, const float* Y,
                  int32_t m, int32_t n, float* out);

/* ---- individual paths (replaces Scene.all_paths / all_valid_paths / accumulate_over_paths,
 *      differt2d/scene.py:1156-1334; ImagePath.from_tx_objects_rx, differt2d/geometry.py:1013-1114;
 *      Path.on_objects / intersects_with_objects / is_valid / length, differt2d/geometry.py:811-963) ---- */

/* For each of the P (tx, rx) pairs and each of the C candidates (cand[C][D2D_MAX_ORDER] object
 * indices, order[C] their lengths) solves the path (or, when xys_in != NULL, takes the given points
 * xys_in[P][C][D2D_MAX_ORDER+2][2] and losses loss_in[P][C] (NULL = 0)) and evaluates it against the
 * current scene. Outputs (row-major, [P][C] leading): xys[..][D2D_MAX_ORDER+2][2] (unused rows NaN),
 * loss, valid (is_valid after nan_to_num; 0/1 in hard mode), and optionally on (on_objects),
 * hit (intersects_with_objects) and length (path_length). theta0[C * max(1, many)][D2D_MAX_ORDER] = initial parametric
 * coordinates of the optimiser-based solvers (MinPath geometry.py:1207-1288, FermatPath :1117-1204; the reference
 * draws them from a per-candidate PRNG key shared by all pairs, scene.py:1887-1890), theta0_rows = the number of rows the
 * caller provides (checked against C * max(1, many)); NULL / 0 for ImagePath. Synchronous. */
int d2d_trace_paths(d2d_ctx* ctx, const d2d_params* params, const float* tx, const float* rx, int32_t P,
                    const int32_t* cand, const int32_t* order, int32_t C, const float* theta0, int64_t theta0_rows,
                    const float* xys_in, const float* loss_in, float* xys, float* loss, float* valid, float* on, float* hit,
                    float* length);

/* ---- multi-GPU (one process per GPU; the reference has no multi-device code: its only batching is jax.vmap
 *      over the grid, differt2d/scene.py:1927-1932; RX rows are sharded over ranks and maps are assembled with one
 *      RCCL all-gather on the ctx stream; the scene VJP with one all-reduce) ------------------------------------- */

#define D2D_COMM_ID_BYTES 128
/* Rank 0 creates the id (ncclGetUniqueId) and ships it to the other ranks out of band. */
int d2d_comm_unique_id(uint8_t* id /* [D2D_COMM_ID_BYTES] */);
/* Collective: ncclCommInitRank on the ctx's device. */
int d2d_comm_init(d2d_ctx* ctx, const uint8_t* id, int32_t rank, int32_t world);
int d2d_comm_destroy(d2d_ctx* ctx);
/* Ranks of the communicator as RCCL itself reports them (ncclCommCount): what bench.py prints as `rccl_ranks`. */
int d2d_comm_count(d2d_ctx* ctx, int32_t* ranks);
/* Collective, asynchronous: all-gathers this rank's resident map (what = 0: value map, m*n floats; what = 1:
 * grad_rx map, m*n*2 floats; every rank must hold the same m, n) into a resident buffer [world][...]. The map is
 * first copied aside on the ctx stream and the all-gather runs on a second stream behind that copy, so the NEXT sweep
 * on this ctx overlaps with it; d2d_synchronize, d2d_timer_end, d2d_comm_get_gathered and d2d_comm_allreduce_host wait
 * for the collectives in flight. */
int d2d_comm_allgather_map(d2d_ctx* ctx, int32_t what);
/* Collective, asynchronous: the same maps gathered to ONE rank -- what the reference's single-process caller receives
 * (one assembled m x n (x 2) array, differt2d/scene.py:1927-1953): ncclSend from every other rank, world - 1 ncclRecv on
 * `root` (xGMI is point to point: 7 direct links into the root, 1/world of the all-gather's bytes per GPU).  Same staging
 * copy, second stream and overlap as the all-gather; only `root` may call d2d_comm_get_gathered afterwards. */
int d2d_comm_gather_map(d2d_ctx* ctx, int32_t what, int32_t root);
/* Synchronises and copies the gathered map (what = 0 / 1 as above; the two are kept in separate buffers, so a step may
 * gather both) to out[world * per_rank]; capacity (in floats) must be exactly that. */
int d2d_comm_get_gathered(d2d_ctx* ctx, int32_t what, float* out, int64_t capacity);
/* Collective, asynchronous: sums the resident scene VJP (fp64, 4N+2 values, + N for phi) over ranks in place, on the
 * communication stream behind the reduction that produced it; d2d_get_scene_vjp waits for it. */
int d2d_comm_allreduce_vjp(d2d_ctx* ctx);

/* Collective, synchronous: all-reduces n host doubles over ranks through the GPUs (op 0 = sum, 1 = max).
 * A sum of one value is the barrier; a max is how bench.py takes the slowest rank's time. */
int d2d_comm_allreduce_host(d2d_ctx* ctx, double* values, int32_t n, int32_t op);

/* ---- timing on the ctx stream (HIP events) -------------------------------------------------- */

int d2d_timer_begin(d2d_ctx* ctx);
int d2d_timer_end(d2d_ctx* ctx, float* elapsed_ms); /* synchronises on the end event */

#ifdef __cplusplus
}
#endif
#endif /* D2D_H */
