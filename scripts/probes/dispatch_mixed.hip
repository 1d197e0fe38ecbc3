// Probe (GPU box): does the MI355X refill free wave slots at once when the workgroups in flight differ a lot in length?
// 20 992 single-wave workgroups; the first `n_long` spin `long_it` iterations, the others `short_it` (the sweep's shape: 6 144
// dear parts in front, cheap patches behind).  Every workgroup stamps its start and end (100 MHz counter); the host prints how
// many are in flight and how many have started over time.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/dm scripts/probes/dispatch_mixed.hip && /tmp/dm
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int VG>
__global__ void __launch_bounds__(64) worker(unsigned long long* stamps, int n_long, int long_it, int short_it, int interleave, const float4* table,
                                             int table_n) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // (first thing: the stamps bracket the whole residency)
    // (table_n > 0: the sweep's head as well -- a 2.4 KB table staged into LDS behind a barrier)
    float4* tab = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < table_n; i += 64) tab[i] = table[i];
    __syncthreads();
    const bool is_long = interleave ? ((int)blockIdx.x % interleave == 0 && (int)blockIdx.x / interleave < n_long) : ((int)blockIdx.x < n_long);
    const int spin = is_long ? long_it : short_it;
    float x[VG];
#pragma unroll
    for (int k = 0; k < VG; ++k) x[k] = threadIdx.x + k + (float)(t0 & 1ull);  // (the work cannot move in front of the start stamp)
    for (int i = 0; i < spin; ++i) {
#pragma unroll
        for (int k = 0; k < VG; ++k) x[k] = x[k] * 1.0001f + 0.5f;
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < VG; ++k) s += x[k];
    asm volatile("" : "+v"(s));  // (... nor behind the end stamp)
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0 + (s == 12345.0f ? 1 : 0) + (lds[(blockIdx.x * 7) % 600] == 1.5f ? 1 : 0);
        stamps[2 * blockIdx.x + 1] = t1;
    }
}

int main() {
    const int G = 20992;
    unsigned long long* d;
    CK(hipMalloc(&d, 2 * G * 8));
    std::vector<unsigned long long> h(2 * G);
    float4* table;
    CK(hipMalloc(&table, 150 * sizeof(float4)));
    CK(hipMemset(table, 0, 150 * sizeof(float4)));
    struct Case { const char* name; int n_long, long_it, short_it, interleave, table_n; };
    const Case cases[] = {{"all short", 0, 0, 60, 0, 0}, {"all short, table staged in front of the stamp", 0, 0, 60, 0, 150},
                          {"6144 long in front", 6144, 240, 60, 0, 0}, {"6144 long in front, table staged", 6144, 240, 60, 0, 150},
                          {"6144 long, every third workgroup", 6144, 240, 60, 3, 0}, {"1536 long in front", 1536, 480, 60, 0, 0}};
    for (const Case& c : cases) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(worker<48>, dim3(G), dim3(64), 4096, 0, d, c.n_long, c.long_it, c.short_it, c.interleave, table, c.table_n);
            CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), d, 2 * G * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        double sum = 0, sum_short = 0;
        int n_short = 0;
        for (int i = 0; i < G; ++i) {
            t0 = std::min(t0, h[2 * i]);
            t1 = std::max(t1, h[2 * i + 1]);
            sum += (double)(h[2 * i + 1] - h[2 * i]);
        }
        const double span = (double)(t1 - t0) / 100.0;
        printf("%s: span %.1f us, mean workgroup %.2f us, wave-time / span = %.0f waves in flight on average (8 192 slots at 64 VGPRs)\n", c.name, span,
               sum / G / 100.0, sum / 100.0 / span);
        const int NB = 16;
        for (int b = 0; b < NB; ++b) {
            const unsigned long long tm = t0 + (unsigned long long)((b + 0.5) * (t1 - t0) / NB);
            int inflight = 0, started = 0;
            for (int i = 0; i < G; ++i) {
                started += h[2 * i] <= tm;
                inflight += h[2 * i] <= tm && h[2 * i + 1] > tm;
            }
            printf("   t=%6.1f us: %5d in flight, %5d started\n", (double)(tm - t0) / 100.0, inflight, started);
        }
        (void)sum_short; (void)n_short;
    }
    return 0;
}
