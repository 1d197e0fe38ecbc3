// Host-side launch entry points of the sweep kernels.  The kernels are templates (d2d_kernels.hpp); every
// (kernel family, validity mode) pair is instantiated in its own translation unit (d2d_sweep_tu.hip compiled with
// -DD2D_TU_FAMILY / -DD2D_TU_MODE, see the Makefile) so that the library builds in parallel; d2d.hip only sees these
// declarations.  `mode` is a d2d::Mode; every function returns the hipGetLastError() of its launch.
#pragma once
#include <hip/hip_runtime.h>

#include "d2d_kernels.hpp"

namespace d2d {

// power_fwd_kernel<MODE, STATS, MAXK, false>: one wave per 8 x 8 patch (big launches)
hipError_t launch_fwd(int mode, bool stats, int max_order, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);
// power_fwd_kernel<MODE, false, MAXK, true>: the same sweep with the hand-derived adjoint (value + gradient)
hipError_t launch_fwd_grad(int mode, int max_order, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);
// power_fwd_split_kernel<MODE, STATS, MAXK, 4>: every patch shared by 4 waves (small launches)
constexpr int SPLIT_W = 4;
hipError_t launch_fwd_split(int mode, bool stats, int max_order, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);
// power_fwd_txg_kernel<MODE, MAXK, GRADK>: TX grids, culled
hipError_t launch_txg(int mode, bool grad, int max_order, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);
// power_vg_kernel<MODE, TXG, GRADK>: exhaustive sweeps (strict_nan value+grad; "txg_exhaustive" values)
hipError_t launch_vg(int mode, bool txg, bool grad, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a);

// region_list_kernel<K, GRAD>: candidate lists of order K (2..4) per region and slice; grid = regions x slices
hipError_t launch_region_lists(int K, bool grad, dim3 grid, size_t lds, hipStream_t stream, const SweepArgs& a, const RegionLists& rl);

}  // namespace d2d
