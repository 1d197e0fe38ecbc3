// Host-side launch entry points of the sweep kernels.  The kernels are templates (d2d_kernels.hpp); every
// (kernel family, validity mode) pair is instantiated in its own translation unit (d2d_sweep_tu.hip compiled with
// -DD2D_TU_FAMILY / -DD2D_TU_MODE, see the Makefile) so that the library builds in parallel; d2d.hip only sees these
// declarations.  `mode` is a d2d::Mode; every function returns the hipGetLastError() of its launch.
#pragma once
#include <hip/hip_runtime.h>

#include "d2d_kernels.hpp"

namespace d2d {

// `listed`: the LISTED build (orders >= 2 from the region candidate lists, a.rl); otherwise the enumerating build, which
// walks the queue of left-over patches when a.fb_n is set.
// power_fwd_kernel<MODE, STATS, MAXK, false>: one wave per 8 x 8 patch (big launches)
hipError_t launch_fwd(int mode, bool listed, bool stats, int max_order, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);
// power_fwd_kernel<MODE, false, MAXK, true>: the same sweep with the hand-derived adjoint (value + gradient)
hipError_t launch_fwd_grad(int mode, bool listed, int max_order, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);
// power_fwd_split_kernel<MODE, STATS, MAXK, 4>: every patch shared by 4 waves (small launches)
constexpr int SPLIT_W = 4;
hipError_t launch_fwd_split(int mode, bool listed, bool stats, int max_order, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);
// power_fwd_coop_kernel<MODE, MAXK, W>: small launches with region lists, every patch shared by W = 4, 8 or 16 waves candidate by candidate
hipError_t launch_fwd_coop(int mode, int max_order, int W, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);
// power_fwd_txg_kernel<MODE, MAXK, GRADK>: TX grids, culled
hipError_t launch_txg(int mode, bool listed, bool grad, int max_order, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);
// power_vg_kernel<MODE, TXG, GRADK>: exhaustive sweeps (strict_nan value+grad; "txg_exhaustive" values)
hipError_t launch_vg(int mode, bool txg, bool grad, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);

// region_list_kernel<K, GRAD>: candidate lists of order K (2..4) of level `lv` by enumeration; grid = regions x slices
hipError_t launch_region_lists(int K, bool grad, bool txg, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a, const RegionLevel& lv,
                               const ListPool& lp);
// region_refine_kernel<K, GRAD>: the lists of level `lv` (one per region) from those of `parent`; grid = regions of `lv`
hipError_t launch_region_refine(int K, bool grad, bool txg, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a, const RegionLevel& lv,
                                const RegionLevel& parent, const ListPool& lp, int* flag);

// nan_scan_kernel<APPROX, TXG, MAXK> (d2d_nanscan.hpp): the reference's autodiff NaN positions, behind a culled value+grad sweep
// regions: nan_scan_region_kernel (16 waves per region of 4 x 4 patches; grid = regions) instead of one wave per patch
hipError_t launch_nan_scan(bool approx, bool txg, int max_order, bool regions, bool dbg, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a,
                           unsigned long long* stats);
// nan_apply_kernel: the flags of a scan that ran beside the sweep (SweepArgs::nan_cell_bits / nan_row_bits) -> grad / partial
hipError_t launch_nan_apply(hipStream_t stream, const SweepArgs& a, long tiles);

}  // namespace d2d
