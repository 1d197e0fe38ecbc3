// Probe (GPU box): how much dynamic LDS one workgroup may ask for on gfx950, with and without
// hipFuncAttributeMaxDynamicSharedMemorySize.   hipcc --offload-arch=gfx950 -o /tmp/lds_probe lds_limit_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o, int n) {
    extern __shared__ float x[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = (float)i;
    __syncthreads();
    if (threadIdx.x == 0) o[blockIdx.x] = x[n - 1];
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
    float* o;
    hipMalloc(&o, 1024 * sizeof(float));
    for (int kb : {48, 64, 96, 128, 156, 160}) {
        const int n = kb * 256;
        hipLaunchKernelGGL(k, dim3(4), dim3(256), (size_t)kb * 1024, 0, o, n);
        hipError_t e1 = hipGetLastError();
        hipError_t e2 = hipDeviceSynchronize();
        printf("%3d KB without attribute: launch %s, sync %s\n", kb, hipGetErrorName(e1), hipGetErrorName(e2));
    }
    hipError_t ea = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("hipFuncSetAttribute(160 KB): %s\n", hipGetErrorName(ea));
    for (int kb : {96, 128, 156, 160}) {
        const int n = kb * 256;
        hipLaunchKernelGGL(k, dim3(4), dim3(256), (size_t)kb * 1024, 0, o, n);
        hipError_t e1 = hipGetLastError();
        hipError_t e2 = hipDeviceSynchronize();
        float h = 0;
        hipMemcpy(&h, o, 4, hipMemcpyDeviceToHost);
        printf("%3d KB with attribute: launch %s, sync %s, x[n-1] = %.0f (want %d)\n", kb, hipGetErrorName(e1), hipGetErrorName(e2), h, n - 1);
    }
    return 0;
}
