// Translation unit of the reverse-mode MinPath / FermatPath value+gradient sweep (d2d_optrev.hpp).
#define D2D_OPTREV_KERNELS 1
#include "d2d_optrev.hpp"

namespace d2d {

template <bool CUST>
static void launch_opt_rev_t(int K, const OptRevArgs& a, int c_first, dim3 grid, size_t lds, hipStream_t stream) {
    switch (K) {
        case 0: hipLaunchKernelGGL((power_opt_rev_kernel<0, CUST>), grid, dim3(64), lds, stream, a, c_first); break;
        case 1: hipLaunchKernelGGL((power_opt_rev_kernel<1, CUST>), grid, dim3(64), lds, stream, a, c_first); break;
        case 2: hipLaunchKernelGGL((power_opt_rev_kernel<2, CUST>), grid, dim3(64), lds, stream, a, c_first); break;
        case 3: hipLaunchKernelGGL((power_opt_rev_kernel<3, CUST>), grid, dim3(64), lds, stream, a, c_first); break;
        default: hipLaunchKernelGGL((power_opt_rev_kernel<4, CUST>), grid, dim3(64), lds, stream, a, c_first); break;
    }
}

hipError_t launch_opt_rev(int K, const OptRevArgs& a, int c_first, dim3 grid, size_t lds, hipStream_t stream) {
    if (a.g.s.fun_id == D2D_FUN_CUSTOM) launch_opt_rev_t<true>(K, a, c_first, grid, lds, stream);
    else launch_opt_rev_t<false>(K, a, c_first, grid, lds, stream);
    return hipGetLastError();
}

}  // namespace d2d
