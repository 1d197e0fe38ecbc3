// Probe (GPU box): how fast does the MI355X start workgroups, and what does it depend on?  A launch of G workgroups that do
// nothing but spin for `spin` iterations, for several workgroup sizes, dynamic-LDS sizes and register footprints.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/dr scripts/probes/dispatch_rate.hip && /tmp/dr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { float v[120]; };  // a kernel argument block the size of d2d::SweepArgs (~ 500 bytes)

template <int VG>  // VG: extra live VGPRs per lane
__global__ void worker(float* out, int spin, Big big) {
    extern __shared__ float lds[];
    float x[VG];
#pragma unroll
    for (int k = 0; k < VG; ++k) x[k] = threadIdx.x + k + big.v[k % 120];
    for (int i = 0; i < spin; ++i) {
#pragma unroll
        for (int k = 0; k < VG; ++k) x[k] = x[k] * 1.0001f + 0.5f;
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < VG; ++k) s += x[k];
    if (s == 12345.0f) out[blockIdx.x] = s + lds[threadIdx.x];
}

template <int VG>
int run(const char* name, int G, int block, size_t lds, int spin, float* out) {
    Big big{};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(worker<VG>, dim3(G), dim3(block), lds, 0, out, spin, big);
    CK(hipDeviceSynchronize());
    const int R = 10;
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(worker<VG>, dim3(G), dim3(block), lds, 0, out, spin, big);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / R;
    printf("%-28s G=%6d block=%4d lds=%6zu spin=%5d : %8.2f us per launch, %7.1f workgroups / us, %7.1f waves / us\n", name, G, block, lds, spin, us,
           G / us, G * (block / 64.0) / us);
    return 0;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 1 << 22));
    const int G = 20992;
    for (int spin : {0, 200, 1000}) {
        run<1>("1 vgpr", G, 64, 0, spin, out);
        run<1>("1 vgpr, 4 KB lds", G, 64, 4096, spin, out);
        run<1>("1 vgpr, 16 KB lds", G, 64, 16384, spin, out);
        run<64>("64 vgprs", G, 64, 0, spin, out);
        run<64>("64 vgprs, 4 KB lds", G, 64, 4096, spin, out);
        run<1>("1 vgpr, 256 threads", G / 4, 256, 0, spin, out);
        run<1>("1 vgpr, 256 thr, 4 KB lds", G / 4, 256, 4096, spin, out);
        run<64>("64 vgprs, 256 thr, 16 KB lds", G / 4, 256, 16384, spin, out);
        run<1>("1 vgpr, 128 threads", G / 2, 128, 0, spin, out);
        run<64>("64 vgprs, 128 thr, 8 KB lds", G / 2, 128, 8192, spin, out);
    }
    run<1>("1 vgpr, 4x the grid", 4 * G, 64, 0, 0, out);
    run<1>("1 vgpr, 4x, 4 KB lds", 4 * G, 64, 4096, 0, out);
    return 0;
}
