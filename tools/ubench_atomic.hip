// ubench_atomic.hip — rate ceiling of scattered no-return 32-bit atomic adds (the match-join's count updates) as a function of the
// working set, the scope (agent = performed at the memory side on this multi-XCD part; workgroup = in the XCD's L2) and the occupancy.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_atomic tools/ubench_atomic.hip ; run: tools/ubench_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
template <int SCOPE, bool STORE>
__global__ __launch_bounds__(1024) void k_atomic(uint32_t *__restrict__ t, uint64_t nwords, int iters)
{
    uint64_t s = mix(((uint64_t)blockIdx.x << 20) ^ threadIdx.x ^ 0x1234567ull);
    for (int i = 0; i < iters; i++) {
        s = mix(s + 0x9e3779b97f4a7c15ULL);
        uint32_t *p = t + s % nwords;
        if (STORE) *(volatile uint16_t *)p = (uint16_t)i;                 // scattered 2-byte store, for comparison
        else if (SCOPE == 0) atomicAdd(p, 1u);
        else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
// the join's window: `rows` rows of `stride` words, of which a piece of `piece` words (at piece0) is live - the same bytes as a contiguous
// window of rows * piece words, spread over rows * stride
__global__ __launch_bounds__(1024) void k_atomic_window(uint32_t *__restrict__ t, uint32_t rows, uint64_t stride, uint32_t piece, uint64_t piece0, int iters)
{
    uint64_t s = mix(((uint64_t)blockIdx.x << 20) ^ threadIdx.x ^ 0x1234567ull);
    for (int i = 0; i < iters; i++) {
        s = mix(s + 0x9e3779b97f4a7c15ULL);
        atomicAdd(t + (uint64_t)((uint32_t)s % rows) * stride + piece0 + (uint32_t)(s >> 32) % piece, 1u);
    }
}
static double run_window(uint32_t *t, uint32_t rows, uint64_t stride, uint32_t piece, int wgs, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k_atomic_window, dim3(wgs), dim3(1024), 0, 0, t, rows, stride, piece, (uint64_t)0, 4);
    hipEventRecord(a);
    hipLaunchKernelGGL(k_atomic_window, dim3(wgs), dim3(1024), 0, 0, t, rows, stride, piece, (uint64_t)0, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)wgs * 1024 * iters / (ms * 1e-3) / 1e9;
}
template <int SCOPE, bool STORE> static double run(uint32_t *t, uint64_t nwords, int wgs, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_atomic<SCOPE, STORE>), dim3(wgs), dim3(1024), 0, 0, t, nwords, 4);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_atomic<SCOPE, STORE>), dim3(wgs), dim3(1024), 0, 0, t, nwords, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return (double)wgs * 1024 * iters / (ms * 1e-3) / 1e9;
}
int main()
{
    uint32_t *t; const uint64_t maxb = 8ull << 30;
    hipMalloc(&t, maxb); hipMemset(t, 0, maxb);
    printf("scattered no-return atomicAdd(u32), G atomics/s\nworking set   WGs(x1024)   agent-scope   workgroup-scope   2-byte stores\n");
    for (uint64_t sz : {16ull << 20, 256ull << 20, 2ull << 30, 8ull << 30})
        for (int wgs : {256, 512}) {
            const uint64_t nw = sz / 4;
            printf("%8llu MB   %6d   %10.2f   %14.2f   %12.2f\n", (unsigned long long)(sz >> 20), wgs, run<0, false>(t, nw, wgs, 400), run<1, false>(t, nw, wgs, 400), run<0, true>(t, nw, wgs, 400));
        }
    // the match-join's count matrix: 2500 query rows of 300 032 16-bit counters (150 016 words); a workgroup's chunk is 8192 nodes = 4096 words of
    // every row; ~4.65 chunks are live at a time in chunk-major order, all 37 otherwise
    printf("window of a 2500-row matrix (row stride 150016 words): live piece per row -> G atomics/s (contiguous window of the same bytes)\n");
    for (uint32_t piece : {4096u, 4096u * 2, 4096u * 5, 4096u * 16, 150016u})
        printf("%8u words/row (%6.1f MB)   strided %8.2f   contiguous %8.2f\n", piece, 2500.0 * piece * 4 / 1048576.0, run_window(t, 2500, 150016, piece, 512, 400),
               run<0, false>(t, (uint64_t)2500 * piece, 512, 400));
    return 0;
}
