// ubench_gather.hip — random 2-byte lookups: request-rate ceiling of the memory system as a function of the working set
// (L2 4 MB/XCD, Infinity Cache 256 MB, HBM). The dense traversal's count lookups are exactly this access pattern.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_gather tools/ubench_gather.hip ; run: tools/ubench_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
__device__ __forceinline__ uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
// every lane: `iters` rounds of U independent loads at random element indices in [0, n) (per-wave window of `win` elements when win > 0)
template <int U>
__global__ __launch_bounds__(512) void k_gather(const uint16_t *__restrict__ t, uint64_t n, uint64_t win, int iters, uint32_t *__restrict__ out)
{
    uint64_t s = mix(((uint64_t)blockIdx.x << 20) ^ threadIdx.x ^ 0x1234567ull);
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        uint32_t v[U];
        uint64_t base = 0;
        if (win) base = (mix(((uint64_t)blockIdx.x << 32) ^ ((uint64_t)(threadIdx.x >> 6) << 24) ^ (uint64_t)i) % (n - win));   // wave-uniform window
#pragma unroll
        for (int u = 0; u < U; u++) { s = mix(s + 0x9e3779b97f4a7c15ULL); const uint64_t idx = win ? base + s % win : s % n; v[u] = t[idx]; }
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ __launch_bounds__(512) void k_chase(const uint16_t *__restrict__ t, uint64_t n, int act, int iters, uint32_t *__restrict__ out)
{
    if ((int)threadIdx.x >= act) return;
    uint64_t s = mix(((uint64_t)blockIdx.x << 20) ^ threadIdx.x ^ 0x777ull);
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) { const uint32_t v = t[s % n]; acc += v; s = mix(s + v + 0x9e3779b97f4a7c15ULL); }    // v is always 0x0101: the chain is still a true dependency
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
int main()
{
    int ncu = 256;
    uint16_t *t; uint32_t *out;
    const uint64_t maxb = 8ull << 30;
    hipMalloc(&t, maxb); hipMalloc(&out, 64); hipMemset(t, 1, maxb);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("working set      window   Greq/s   (512 lanes x %d WG, 4 loads in flight per lane)\n", ncu * 3);
    const uint64_t sizes[] = {16ull << 20, 64ull << 20, 128ull << 20, 200ull << 20, 512ull << 20, 2ull << 30, 8ull << 30};
    const uint64_t wins[] = {0, 300000, 4096, 256};
    for (uint64_t win : wins)
        for (uint64_t sz : sizes) {
            const uint64_t n = sz / 2;
            const int iters = 400;
            hipLaunchKernelGGL(k_gather<4>, dim3(ncu * 3), dim3(512), 0, 0, t, n, win, 20, out);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_gather<4>, dim3(ncu * 3), dim3(512), 0, 0, t, n, win, iters, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%8llu MB  %9llu  %7.2f\n", (unsigned long long)(sz >> 20), (unsigned long long)win, (double)ncu * 3 * 512 * 4 * iters / (ms * 1e-3) / 1e9);
        }
    // latency-throughput curve: `act` lanes per workgroup issue ONE dependent load at a time (next index derived from the loaded value);
    // latency = outstanding requests / request rate (Little)
    printf("\ndependent loads, 2 GB working set: active lanes per WG x 768 WG -> Greq/s, implied latency\n");
    for (int act : {1, 8, 32, 64, 128, 256, 512}) {
        const uint64_t n = (2ull << 30) / 2;
        const int iters = 2000;
        hipLaunchKernelGGL(k_chase, dim3(ncu * 3), dim3(512), 0, 0, t, n, act, 50, out);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_chase, dim3(ncu * 3), dim3(512), 0, 0, t, n, act, iters, out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double outstanding = (double)ncu * 3 * act, rate = outstanding * iters / (ms * 1e-3);
        printf("%4d lanes: %9.0f outstanding  %7.2f Greq/s  latency %.2f us\n", act, outstanding, rate / 1e9, outstanding / rate * 1e6);
    }
    return 0;
}
