// ubench_fetchcal.hip — calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of THIS library's kernels
// (MI355X_MICROARCH.md "HBM": only 16 B/lane streaming reads are calibrated - FETCH_SIZE reports half their bytes; "calibrate on a known
// byte count in your own access pattern"). Every kernel below moves a KNOWN number of bytes over an 8 GiB buffer (32x the Infinity Cache):
//   k_stream16   16 B per lane, coalesced                      (row gather of k_hnsw_search, sketch input)
//   k_stream4     4 B per lane, coalesced                      (column stream of k_match_join)
//   k_gather2    random 2-byte reads, one per lane per round   (count look-ups of k_hnsw_search_dense)
//   k_atomic2    scattered no-return 32-bit atomics            (match recording of k_match_join)
//   k_write16    16 B per lane stores
// run under:  rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/ubench_fetchcal   and   --pmc WRITE_SIZE   (tools/pmc_fetchcal.sh);
// the program prints the bytes / requests each launch moved so that the condensed counters can be divided by them.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_fetchcal tools/ubench_fetchcal.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint64_t mix(uint64_t z) { z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
__global__ __launch_bounds__(512) void k_stream16(const uint4 *__restrict__ p, uint64_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(512) void k_stream4(const uint32_t *__restrict__ p, uint64_t n4, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x12345u) out[0] = acc;
}
__global__ __launch_bounds__(512) void k_gather2(const uint16_t *__restrict__ p, uint64_t n2, int iters, uint32_t *out)
{
    uint64_t s = mix(((uint64_t)blockIdx.x << 20) ^ threadIdx.x ^ 0x1234567ull);
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) { s = mix(s + 0x9e3779b97f4a7c15ULL); acc += p[s % n2]; }
    if (acc == 0xFFFFFFFFu) out[0] = acc;
}
__global__ __launch_bounds__(512) void k_atomic2(uint32_t *__restrict__ p, uint64_t n4, int iters)
{
    uint64_t s = mix(((uint64_t)blockIdx.x << 20) ^ threadIdx.x ^ 0x7654321ull);
    for (int i = 0; i < iters; i++) { s = mix(s + 0x9e3779b97f4a7c15ULL); atomicSub(&p[s % n4], 0x10000u); }
}
__global__ __launch_bounds__(512) void k_write16(uint4 *__restrict__ p, uint64_t n16)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
int main()
{
    const uint64_t bytes = 8ull << 30;
    void *buf; uint32_t *out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    const int wgs = 256 * 4, iters = 2048;
    const uint64_t lanes = (uint64_t)wgs * 512;
    hipLaunchKernelGGL(k_stream16, dim3(wgs), dim3(512), 0, 0, (const uint4 *)buf, bytes / 16, out);
    hipLaunchKernelGGL(k_stream4, dim3(wgs), dim3(512), 0, 0, (const uint32_t *)buf, bytes / 4, out);
    hipLaunchKernelGGL(k_gather2, dim3(wgs), dim3(512), 0, 0, (const uint16_t *)buf, bytes / 2, iters, out);
    hipLaunchKernelGGL(k_atomic2, dim3(wgs), dim3(512), 0, 0, (uint32_t *)buf, (2ull << 30) / 4, iters);
    hipLaunchKernelGGL(k_write16, dim3(wgs), dim3(512), 0, 0, (uint4 *)buf, bytes / 16);
    hipDeviceSynchronize();
    printf("known: k_stream16 read_bytes %llu\n", (unsigned long long)bytes);
    printf("known: k_stream4 read_bytes %llu\n", (unsigned long long)bytes);
    printf("known: k_gather2 requests %llu (2 B each, random over 8 GiB)\n", (unsigned long long)(lanes * iters));
    printf("known: k_atomic2 requests %llu (32-bit no-return atomics, random over 2 GiB)\n", (unsigned long long)(lanes * iters));
    printf("known: k_write16 write_bytes %llu\n", (unsigned long long)bytes);
    return 0;
}
