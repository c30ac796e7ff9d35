// micro-benchmark: peak rate of "count mismatches" formulations on gfx950 (pair-elements per second, registers only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define ITERS 2048
template <int V>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t c[8] = {0,0,0,0,0,0,0,0}, b[8], a = threadIdx.x ^ seed;
    for (int i = 0; i < 8; i++) b[i] = threadIdx.x * 3 + i + seed;
    for (int it = 0; it < ITERS; it++) {
        if (V == 0) {
            asm volatile("v_cmp_ne_u32 vcc, %8, %9\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_cmp_ne_u32 vcc, %8, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
                         "v_cmp_ne_u32 vcc, %8, %11\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\tv_cmp_ne_u32 vcc, %8, %12\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
                         "v_cmp_ne_u32 vcc, %8, %13\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\tv_cmp_ne_u32 vcc, %8, %14\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"
                         "v_cmp_ne_u32 vcc, %8, %15\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\tv_cmp_ne_u32 vcc, %8, %16\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc"
                         : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                         : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
        } else if (V == 1) {   // 4 compares into distinct SGPR pairs, then 4 adds-with-carry (VOP3 forms)
            asm volatile("v_cmp_ne_u32 s[20:21], %8, %9\n\tv_cmp_ne_u32 s[22:23], %8, %10\n\tv_cmp_ne_u32 s[24:25], %8, %11\n\tv_cmp_ne_u32 s[26:27], %8, %12\n\t"
                         "v_addc_co_u32 %0, s[20:21], 0, %0, s[20:21]\n\tv_addc_co_u32 %1, s[22:23], 0, %1, s[22:23]\n\tv_addc_co_u32 %2, s[24:25], 0, %2, s[24:25]\n\tv_addc_co_u32 %3, s[26:27], 0, %3, s[26:27]\n\t"
                         "v_cmp_ne_u32 s[20:21], %8, %13\n\tv_cmp_ne_u32 s[22:23], %8, %14\n\tv_cmp_ne_u32 s[24:25], %8, %15\n\tv_cmp_ne_u32 s[26:27], %8, %16\n\t"
                         "v_addc_co_u32 %4, s[20:21], 0, %4, s[20:21]\n\tv_addc_co_u32 %5, s[22:23], 0, %5, s[22:23]\n\tv_addc_co_u32 %6, s[24:25], 0, %6, s[24:25]\n\tv_addc_co_u32 %7, s[26:27], 0, %7, s[26:27]"
                         : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                         : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "s20","s21","s22","s23","s24","s25","s26","s27");
        } else if (V == 2) {   // xor ; min(x,1) ; add  (no carry chain)
            for (int i = 0; i < 8; i++) { uint32_t x; asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(a), "v"(b[i])); asm volatile("v_min_u32 %0, 1, %0" : "+v"(x)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[i]) : "v"(x)); }
        } else if (V == 3) {   // sub-with-borrow trick: cnt -= -1 when ne  (v_cmp ; v_subb) — same shape, checks port symmetry
            asm volatile("v_cmp_ne_u32 vcc, %8, %9\n\tv_subb_co_u32 %0, vcc, %0, 0, vcc\n\tv_cmp_ne_u32 vcc, %8, %10\n\tv_subb_co_u32 %1, vcc, %1, 0, vcc\n\t"
                         "v_cmp_ne_u32 vcc, %8, %11\n\tv_subb_co_u32 %2, vcc, %2, 0, vcc\n\tv_cmp_ne_u32 vcc, %8, %12\n\tv_subb_co_u32 %3, vcc, %3, 0, vcc\n\t"
                         "v_cmp_ne_u32 vcc, %8, %13\n\tv_subb_co_u32 %4, vcc, %4, 0, vcc\n\tv_cmp_ne_u32 vcc, %8, %14\n\tv_subb_co_u32 %5, vcc, %5, 0, vcc\n\t"
                         "v_cmp_ne_u32 vcc, %8, %15\n\tv_subb_co_u32 %6, vcc, %6, 0, vcc\n\tv_cmp_ne_u32 vcc, %8, %16\n\tv_subb_co_u32 %7, vcc, %7, 0, vcc"
                         : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])
                         : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]) : "vcc");
        }
        a += 1;
    }
    uint32_t r = 0; for (int i = 0; i < 8; i++) r ^= c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int V> void run(const char *name, uint32_t *d, int blocks_per_cu)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(a); hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, 2u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double pe = (double)blocks * 256 * ITERS * 8;
    printf("%-28s waves/SIMD=%d  %8.3f ms  %.3e pair-elements/s\n", name, blocks_per_cu, ms, pe / ms * 1e3);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {2, 4, 8}) {
        if (w == 2) { run<0>("cmp+addc via vcc", d, 2); run<1>("cmp->sgpr x4, addc x4", d, 2); run<2>("xor,min,add", d, 2); run<3>("cmp+subb via vcc", d, 2); }
        if (w == 4) { run<0>("cmp+addc via vcc", d, 4); run<1>("cmp->sgpr x4, addc x4", d, 4); run<2>("xor,min,add", d, 4); }
        if (w == 8) { run<0>("cmp+addc via vcc", d, 8); run<1>("cmp->sgpr x4, addc x4", d, 8); run<2>("xor,min,add", d, 8); }
    }
    return 0;
}
