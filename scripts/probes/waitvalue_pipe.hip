// Probe (GPU box): a pipelined sequence of kernels alternating between two streams, each waiting (hipStreamWaitValue32) for the word
// the previous kernel's late workgroup writes; optional hipStreamWriteValue32 behind every kernel.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wvp waitvalue_pipe.hip && /tmp/wvp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(64) worker(float* out, unsigned* flag, unsigned seq, int flag_item, int spin) {
    if (threadIdx.x == 0 && (int)blockIdx.x == flag_item && flag) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    float x = threadIdx.x;
    const int n = spin + (blockIdx.x < 64 ? 8 * spin : 0);  // a few long workgroups that start first: a drain
    for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.0f) out[0] = x;
}

__global__ void flag_max(unsigned* flag, unsigned seq) { __hip_atomic_fetch_max(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

int main() {
    unsigned* flag = nullptr;
    CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    CK(hipMemset(flag, 0, 8));
    hipStream_t s[2];
    for (int i = 0; i < 2; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    float* out;
    CK(hipMalloc(&out, 4));
    const int G = 20000, spin = 600, K = 200;
    for (int mode = 0; mode < 5; ++mode) {
        // 0: one stream, plain; 1: two streams, wait-value; 2: two streams, wait-value + a one-thread atomic-max kernel behind every kernel;
        // 3: two streams, events (k + 1 behind k entirely); 4: as 1 again
        CK(hipDeviceSynchronize());
        CK(hipMemset(flag, 0, 8));
        unsigned seq0 = 1000u * (mode + 1);
        hipEvent_t ev[2];
        for (int i = 0; i < 2; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; ++k) {
            const unsigned seq = seq0 + k;
            hipStream_t S = (mode == 0) ? s[0] : s[k & 1];
            if ((mode == 1 || mode == 2 || mode == 4) && k > 0) CK(hipStreamWaitValue32(S, flag, seq - 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
            if (mode == 3 && k > 0) CK(hipStreamWaitEvent(S, ev[(k - 1) & 1], 0));
            hipLaunchKernelGGL(worker, dim3(G), dim3(64), 0, S, out, flag, seq, G - 1 - 1024, spin);
            if (mode == 2) hipLaunchKernelGGL(flag_max, dim3(1), dim3(1), 0, S, flag, seq);  // (hipStreamWriteValue32 here hung the queues)
            if (mode == 3) CK(hipEventRecord(ev[k & 1], S));
        }
        auto t1 = std::chrono::steady_clock::now();
        CK(hipDeviceSynchronize());
        auto t2 = std::chrono::steady_clock::now();
        printf("mode %d: host enqueue %.1f us per kernel, total %.1f us per kernel\n", mode,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / K, std::chrono::duration<double, std::micro>(t2 - t0).count() / K);
        fflush(stdout);
        for (int i = 0; i < 2; ++i) CK(hipEventDestroy(ev[i]));
    }
    return 0;
}
