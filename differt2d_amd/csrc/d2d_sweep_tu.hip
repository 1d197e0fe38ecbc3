// One translation unit per (kernel family, validity mode): instantiates the sweep kernels of that pair and defines the
// per-mode launcher that d2d_launch.cpp's dispatchers call.  Compiled several times by the Makefile with
//   -DD2D_TU_FAMILY={0 fwd, 1 fwd_grad, 2 fwd_split, 3 txg, 4 vg, 5 region lists (mode 0 only), 9 fwd_coop,
//   6 fwd / 7 fwd_grad / 8 fwd_split with the orders >= 2 taken from the region lists (LISTED), 10 NaN scan (mode 0 only)}  -DD2D_TU_MODE={0 hard, 1 hard_sigmoid, 2 sigmoid}
#include "d2d_launch.hpp"
#if D2D_TU_FAMILY == 10
#include "d2d_nanscan.hpp"
#endif

#ifndef D2D_TU_FAMILY
#error "compile with -DD2D_TU_FAMILY=<0..4> -DD2D_TU_MODE=<0..2>"
#endif

namespace d2d {

constexpr int TU_MODE = D2D_TU_MODE;

#if D2D_TU_FAMILY == 0
template <int MODE> hipError_t launch_fwd_m(bool stats, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a);
template <>
hipError_t launch_fwd_m<TU_MODE>(bool stats, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    const dim3 block(64);
    if (stats) {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, true, 2>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, true, 3>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, true, 4>), grid, block, lds, s, a);
    } else {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 2>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 3>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 4>), grid, block, lds, s, a);
    }
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 1
template <int MODE> hipError_t launch_fwd_grad_m(int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a);
template <>
hipError_t launch_fwd_grad_m<TU_MODE>(int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    const dim3 block(64);
    if (max_order <= 2) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 2, true>), grid, block, lds, s, a);
    else if (max_order == 3) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 3, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 4, true>), grid, block, lds, s, a);
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 2
template <int MODE> hipError_t launch_fwd_split_m(bool stats, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a);
template <>
hipError_t launch_fwd_split_m<TU_MODE>(bool stats, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    const dim3 block(64 * SPLIT_W);
    if (stats) {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, true, 2, SPLIT_W>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, true, 3, SPLIT_W>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, true, 4, SPLIT_W>), grid, block, lds, s, a);
    } else {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, false, 2, SPLIT_W>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, false, 3, SPLIT_W>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, false, 4, SPLIT_W>), grid, block, lds, s, a);
    }
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 3
template <int MODE> hipError_t launch_txg_m(bool listed, bool grad, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a);
template <>
hipError_t launch_txg_m<TU_MODE>(bool listed, bool grad, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    const dim3 block(64);
    if (listed) {
        if (grad) {
            if (max_order <= 2) hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 2, true, true>), grid, block, lds, s, a);
            else if (max_order == 3) hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 3, true, true>), grid, block, lds, s, a);
            else hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 4, true, true>), grid, block, lds, s, a);
        } else {
            if (max_order <= 2) hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 2, false, true>), grid, block, lds, s, a);
            else if (max_order == 3) hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 3, false, true>), grid, block, lds, s, a);
            else hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 4, false, true>), grid, block, lds, s, a);
        }
    } else if (grad) {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 2, true>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 3, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 4, true>), grid, block, lds, s, a);
    } else {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 2>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 3>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_txg_kernel<TU_MODE, 4>), grid, block, lds, s, a);
    }
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 4
template <int MODE> hipError_t launch_vg_m(bool txg, bool grad, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a);
template <>
hipError_t launch_vg_m<TU_MODE>(bool txg, bool grad, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    const dim3 block(64);
    if (txg && grad) hipLaunchKernelGGL((power_vg_kernel<TU_MODE, true, true>), grid, block, lds, s, a);
    else if (txg) hipLaunchKernelGGL((power_vg_kernel<TU_MODE, true, false>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((power_vg_kernel<TU_MODE, false, true>), grid, block, lds, s, a);  // (RX grid, values only: launch_fwd)
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 6
template <int MODE> hipError_t launch_fwd_listed_m(bool stats, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a);
template <>
hipError_t launch_fwd_listed_m<TU_MODE>(bool stats, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    const bool wide = grid.y == 4;  // (grid.y carries the waves per workgroup: 1 or 4)
    const dim3 block(wide ? 256 : 64);
    grid.y = 1;
    if (stats) {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, true, 2, false, true>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, true, 3, false, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, true, 4, false, true>), grid, block, lds, s, a);
    } else if (wide) {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 2, false, true, 4>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 3, false, true, 4>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 4, false, true, 4>), grid, block, lds, s, a);
    } else {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 2, false, true>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 3, false, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 4, false, true>), grid, block, lds, s, a);
    }
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 7
template <int MODE> hipError_t launch_fwd_grad_listed_m(int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a);
template <>
hipError_t launch_fwd_grad_listed_m<TU_MODE>(int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    const dim3 block(64);
    if (max_order <= 2) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 2, true, true>), grid, block, lds, s, a);
    else if (max_order == 3) hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 3, true, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((power_fwd_kernel<TU_MODE, false, 4, true, true>), grid, block, lds, s, a);
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 8
template <int MODE> hipError_t launch_fwd_split_listed_m(bool stats, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a);
template <>
hipError_t launch_fwd_split_listed_m<TU_MODE>(bool stats, int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    const dim3 block(64 * SPLIT_W);
    if (stats) {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, true, 2, SPLIT_W, true>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, true, 3, SPLIT_W, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, true, 4, SPLIT_W, true>), grid, block, lds, s, a);
    } else {
        if (max_order <= 2) hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, false, 2, SPLIT_W, true>), grid, block, lds, s, a);
        else if (max_order == 3) hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, false, 3, SPLIT_W, true>), grid, block, lds, s, a);
        else hipLaunchKernelGGL((power_fwd_split_kernel<TU_MODE, false, 4, SPLIT_W, true>), grid, block, lds, s, a);
    }
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 9
template <int MODE> hipError_t launch_fwd_coop_m(int max_order, int W, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a);
template <int W>
static void launch_fwd_coop_w(int max_order, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    const dim3 block(64 * W);
    if (max_order <= 2) hipLaunchKernelGGL((power_fwd_coop_kernel<TU_MODE, 2, W>), grid, block, lds, s, a);
    else if (max_order == 3) hipLaunchKernelGGL((power_fwd_coop_kernel<TU_MODE, 3, W>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((power_fwd_coop_kernel<TU_MODE, 4, W>), grid, block, lds, s, a);
}
template <>
hipError_t launch_fwd_coop_m<TU_MODE>(int max_order, int W, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a) {
    if (W == 16) launch_fwd_coop_w<16>(max_order, grid, lds, s, a);
    else if (W == 8) launch_fwd_coop_w<8>(max_order, grid, lds, s, a);
    else launch_fwd_coop_w<4>(max_order, grid, lds, s, a);
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 5
// region_list_kernel / region_refine_kernel <K, GRAD>: independent of the validity mode (compiled once, -DD2D_TU_MODE=0)
hipError_t launch_region_lists(int K, bool grad, bool txg, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a, const RegionLevel& lv,
                               const ListPool& lp) {
    const dim3 block(64);
#define D2D_RL(KK, G, T) hipLaunchKernelGGL((region_list_kernel<KK, G, T>), grid, block, lds, s, a, lv, lp)
#define D2D_RL_K(G, T)            \
    do {                          \
        if (K == 2) D2D_RL(2, G, T);      \
        else if (K == 3) D2D_RL(3, G, T); \
        else D2D_RL(4, G, T);             \
    } while (0)
    (void)grad;  // (the value+grad sweeps read the forward sweeps' lists: their NaN positions come from d2d_nanscan.hpp)
    if (txg) D2D_RL_K(false, true);
    else D2D_RL_K(false, false);
#undef D2D_RL_K
#undef D2D_RL
    return hipGetLastError();
}
hipError_t launch_region_refine(int K, bool grad, bool txg, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a, const RegionLevel& lv,
                                const RegionLevel& parent, const ListPool& lp, int* flag) {
    const dim3 block(64);
#define D2D_RR(KK, G, T) hipLaunchKernelGGL((region_refine_kernel<KK, G, T>), grid, block, lds, s, a, lv, parent, lp, flag)
#define D2D_RR_K(G, T)            \
    do {                          \
        if (K == 2) D2D_RR(2, G, T);      \
        else if (K == 3) D2D_RR(3, G, T); \
        else D2D_RR(4, G, T);             \
    } while (0)
    (void)grad;
    if (txg) D2D_RR_K(false, true);
    else D2D_RR_K(false, false);
#undef D2D_RR_K
#undef D2D_RR
    return hipGetLastError();
}
#elif D2D_TU_FAMILY == 10
// nan_scan_kernel<APPROX, TXG, MAXK>: depends on hard / approx only (compiled once, -DD2D_TU_MODE=0)
hipError_t launch_nan_scan(bool approx, bool txg, int max_order, bool regions, bool dbg, dim3 grid, size_t lds, hipStream_t s, const SweepArgs& a,
                           unsigned long long* stats) {
    const dim3 block(regions ? 64 * NAN_W : 64);
#define D2D_NS(KERNEL, ...)                                                                                   \
    do {                                                                                                      \
        if (max_order <= 2) hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 2>), grid, block, lds, s, a, stats);      \
        else if (max_order == 3) hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 3>), grid, block, lds, s, a, stats); \
        else hipLaunchKernelGGL((KERNEL<__VA_ARGS__, 4>), grid, block, lds, s, a, stats);                     \
    } while (0)
    // (the region kernel's DBG instance: run-time buffer sizes and counters -- tests; the product launches DBG = false)
#define D2D_NSR(A, T)                                                                                                      \
    do {                                                                                                                   \
        if (dbg) {                                                                                                         \
            if (max_order <= 2) hipLaunchKernelGGL((nan_scan_region_kernel<A, T, 2, true>), grid, block, lds, s, a, stats); \
            else if (max_order == 3) hipLaunchKernelGGL((nan_scan_region_kernel<A, T, 3, true>), grid, block, lds, s, a, stats); \
            else hipLaunchKernelGGL((nan_scan_region_kernel<A, T, 4, true>), grid, block, lds, s, a, stats);               \
        } else {                                                                                                           \
            D2D_NS(nan_scan_region_kernel, A, T);                                                                          \
        }                                                                                                                  \
    } while (0)
    if (regions) {
        if (approx) {
            if (txg) D2D_NSR(true, true);
            else D2D_NSR(true, false);
        } else {
            if (txg) D2D_NSR(false, true);
            else D2D_NSR(false, false);
        }
    } else {
        if (approx) {
            if (txg) D2D_NS(nan_scan_kernel, true, true);
            else D2D_NS(nan_scan_kernel, true, false);
        } else {
            if (txg) D2D_NS(nan_scan_kernel, false, true);
            else D2D_NS(nan_scan_kernel, false, false);
        }
    }
#undef D2D_NSR
#undef D2D_NS
    return hipGetLastError();
}
hipError_t launch_nan_apply(hipStream_t s, const SweepArgs& a, long tiles) {
    hipLaunchKernelGGL(nan_apply_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, a.grad, a.partial, a.nan_cell_bits, a.nan_row_bits,
                       a.nan_row_words, a.N, a.m, a.n, tiles);
    return hipGetLastError();
}
#else
#error "unknown D2D_TU_FAMILY"
#endif

}  // namespace d2d
