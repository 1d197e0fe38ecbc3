// Translation unit of the MinPath / FermatPath value+gradient sweep (d2d_optgrad.hpp): the tangent-carrying Adam loop is
// heavy to compile, so it is an object of its own.
#define D2D_OPTGRAD_KERNELS 1
#include "d2d_optgrad.hpp"

namespace d2d {

hipError_t launch_opt_grad(const OptGradArgs& a, dim3 grid, size_t lds, hipStream_t stream) {
    hipLaunchKernelGGL(power_opt_grad_kernel, grid, dim3(64), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_opt_grad_reduce(const float* contrib, const float* gcontrib, int C, long cells, float* out, float* grad, int out_mode,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(opt_grad_reduce_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, stream, contrib, gcontrib, C, cells, out,
                       grad, out_mode);
    return hipGetLastError();
}

}  // namespace d2d
