// Probe (GPU box): can a stream wait for a word that a RUNNING kernel writes (hipStreamWaitValue32 on signal memory), and how
// soon after the store does the waiting stream's kernel start?   hipcc --offload-arch=gfx950 -O2 -o /tmp/wv waitvalue_probe.hip && /tmp/wv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(64) worker(unsigned long long* stamps, unsigned* flag, unsigned seq, int flag_item, int spin) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && (int)blockIdx.x == flag_item && flag) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0 + (x == 12345.0f);
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    unsigned* flag = nullptr;
    CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
    CK(hipMemset(flag, 0, 8));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const int G = 20000, spin = 20000;
    unsigned long long *da, *db;
    CK(hipMalloc(&da, 2 * G * 8));
    CK(hipMalloc(&db, 2 * G * 8));
    std::vector<unsigned long long> ha(2 * G), hb(2 * G);
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: B free-running beside A; 1: B waits for the flag written by A's LAST workgroup at its start; 2: B behind an event after A
        for (unsigned seq = 1 + 10 * mode; seq < 4 + 10 * mode; ++seq) {
            hipLaunchKernelGGL(worker, dim3(G), dim3(64), 0, sa, da, flag, seq, G - 1, spin);
            hipEvent_t ev;
            CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            if (mode == 1) CK(hipStreamWaitValue32(sb, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
            if (mode == 2) { CK(hipEventRecord(ev, sa)); CK(hipStreamWaitEvent(sb, ev, 0)); }
            hipLaunchKernelGGL(worker, dim3(G), dim3(64), 0, sb, db, (unsigned*)nullptr, 0u, -1, spin);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(ha.data(), da, 2 * G * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hb.data(), db, 2 * G * 8, hipMemcpyDeviceToHost));
            unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0;
            for (int i = 0; i < G; ++i) {
                if (ha[2 * i] < a0) a0 = ha[2 * i];
                if (ha[2 * i + 1] > a1) a1 = ha[2 * i + 1];
                if (hb[2 * i] < b0) b0 = hb[2 * i];
                if (hb[2 * i + 1] > b1) b1 = hb[2 * i + 1];
            }
            printf("mode %d: A spans %.1f us (last workgroup starts at %.1f); B starts %.1f us after A's start (%.1f after A's last workgroup started), B spans %.1f, both done at %.1f\n",
                   mode, (a1 - a0) / 100.0, (ha[2 * (G - 1)] - a0) / 100.0, ((long long)b0 - (long long)a0) / 100.0, ((long long)b0 - (long long)ha[2 * (G - 1)]) / 100.0,
                   (b1 - b0) / 100.0, ((b1 > a1 ? b1 : a1) - a0) / 100.0);
            CK(hipEventDestroy(ev));
        }
    }
    return 0;
}
