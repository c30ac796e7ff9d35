// build: hipcc -O2 --offload-arch=gfx950 -o tools/ubench_vmm tools/ubench_vmm.hip   (measured: profiles/r05_ubench_vmm.txt)
// Probe: does this device / runtime support HIP virtual memory management (reserve an address range, map physical chunks into it as needed)?
// Prints the allocation granularity and the time to create + map + set access for chunks, then checks a kernel can write across chunk borders.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(unsigned long long *p, size_t first, size_t n) { size_t i = first + blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = i; }
__global__ void check(const unsigned long long *p, size_t n, unsigned long long *bad) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n && p[i] != i) atomicAdd(bad, 1ull); }
int main()
{
    int dev = 0, sup = 0;
    CK(hipSetDevice(dev));
    CK(hipDeviceGetAttribute(&sup, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
    printf("virtual memory management supported: %d\n", sup);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    const size_t chunk = ((size_t)1 << 30), nchunks = 8, total = chunk * nchunks;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, (size_t)256 << 30, 0, nullptr, 0));
    printf("reserved 256 GB of address space at %p\n", va);
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
    hipMemGenericAllocationHandle_t h[nchunks];
    for (size_t i = 0; i < nchunks; i++) {
        auto t0 = std::chrono::steady_clock::now();
        CK(hipMemCreate(&h[i], chunk, &prop, 0));
        CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0));
        CK(hipMemSetAccess((char *)va + i * chunk, chunk, &acc, 1));
        auto t1 = std::chrono::steady_clock::now();
        printf("chunk %zu mapped in %.2f ms\n", i, std::chrono::duration<double, std::milli>(t1 - t0).count());
        // a kernel over everything mapped so far (the earlier chunks keep their content and their addresses)
        const size_t n = (i + 1) * chunk / 8;
        if (i == 0 || i == nchunks - 1) {
            unsigned long long *bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
            if (i == 0) fill<<<(unsigned)((n + 255) / 256), 256>>>((unsigned long long *)va, 0, n);
            else { fill<<<(unsigned)((n - chunk / 8 + 255) / 256), 256>>>((unsigned long long *)va, chunk / 8, n); }
            check<<<(unsigned)((n + 255) / 256), 256>>>((unsigned long long *)va, i == 0 ? n : n, bad);
            unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
            printf("  after chunk %zu: %llu wrong words of %zu\n", i, hb, n);
            CK(hipFree(bad));
        }
    }
    size_t fr, tot; CK(hipMemGetInfo(&fr, &tot)); printf("free %zu of %zu\n", fr, tot);
    // bigger chunks: how long, and does a kernel see them at full speed?
    for (size_t gb : {2, 8, 32}) {
        hipMemGenericAllocationHandle_t hb; const size_t big = gb << 30;
        auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipMemCreate(&hb, big, &prop, 0); if (e != hipSuccess) { printf("%zu GB: hipMemCreate -> %s\n", gb, hipGetErrorString(e)); continue; }
        e = hipMemMap((char *)va + total, big, 0, hb, 0); if (e != hipSuccess) { printf("%zu GB: hipMemMap -> %s\n", gb, hipGetErrorString(e)); (void)hipMemRelease(hb); continue; }
        e = hipMemSetAccess((char *)va + total, big, &acc, 1);
        auto t1 = std::chrono::steady_clock::now();
        if (e != hipSuccess) printf("%zu GB: hipMemSetAccess -> %s\n", gb, hipGetErrorString(e));
        else {
            printf("%zu GB chunk mapped in %.2f ms\n", gb, std::chrono::duration<double, std::milli>(t1 - t0).count());
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            for (int rep = 0; rep < 2; rep++) {
                CK(hipEventRecord(a)); fill<<<(unsigned)((big / 8 + 255) / 256), 256>>>((unsigned long long *)((char *)va + total), 0, big / 8); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms = 0; CK(hipEventElapsedTime(&ms, a, b)); printf("  fill pass %d: %.2f ms (%.2f TB/s)\n", rep, ms, big / ms / 1e9);
            }
        }
        (void)hipMemUnmap((char *)va + total, big); (void)hipMemRelease(hb);
    }
    for (size_t i = 0; i < nchunks; i++) { CK(hipMemUnmap((char *)va + i * chunk, chunk)); CK(hipMemRelease(h[i])); }
    CK(hipMemAddressFree(va, (size_t)256 << 30));
    CK(hipMemGetInfo(&fr, &tot)); printf("after release: free %zu of %zu\n", fr, tot);
    // hipMalloc for comparison, three times over (the first large allocation of a process pays for more than itself), then the same 8 GB as 1 GB mappings
    for (int rep = 0; rep < 3; rep++) {
        void *p; auto t0 = std::chrono::steady_clock::now(); CK(hipMalloc(&p, (size_t)8 << 30)); auto t1 = std::chrono::steady_clock::now();
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a)); fill<<<(unsigned)((((size_t)8 << 30) / 8 + 255) / 256), 256>>>((unsigned long long *)p, 0, ((size_t)8 << 30) / 8); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        auto t2 = std::chrono::steady_clock::now(); CK(hipFree(p)); auto t3 = std::chrono::steady_clock::now();
        printf("hipMalloc 8 GB %.2f ms, first fill %.2f ms, hipFree %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), ms, std::chrono::duration<double, std::milli>(t3 - t2).count());
    }
    for (int rep = 0; rep < 3; rep++) {
        void *v2 = nullptr; CK(hipMemAddressReserve(&v2, (size_t)8 << 30, 0, nullptr, 0));
        hipMemGenericAllocationHandle_t hh[8];
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 8; i++) { CK(hipMemCreate(&hh[i], chunk, &prop, 0)); CK(hipMemMap((char *)v2 + i * chunk, chunk, 0, hh[i], 0)); CK(hipMemSetAccess((char *)v2 + i * chunk, chunk, &acc, 1)); }
        CK(hipDeviceSynchronize());
        auto t1 = std::chrono::steady_clock::now();
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a)); fill<<<(unsigned)((((size_t)8 << 30) / 8 + 255) / 256), 256>>>((unsigned long long *)v2, 0, ((size_t)8 << 30) / 8); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
        auto t2 = std::chrono::steady_clock::now();
        for (int i = 0; i < 8; i++) { CK(hipMemUnmap((char *)v2 + i * chunk, chunk)); CK(hipMemRelease(hh[i])); }
        CK(hipMemAddressFree(v2, (size_t)8 << 30));
        auto t3 = std::chrono::steady_clock::now();
        printf("8 x 1 GB mapped %.2f ms, first fill %.2f ms, released %.2f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), ms, std::chrono::duration<double, std::milli>(t3 - t2).count());
    }
    printf("OK\n");
    return 0;
}
