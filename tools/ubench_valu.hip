// micro-benchmark: issue rate of the integer VALU instructions the sketch kernels lean on (gfx950).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP 64
#define ITERS 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x1234567, a3 = a0 + 77, b = seed | 1, c = seed + 5;
    uint64_t q0 = ((uint64_t)a0 << 32) | a1, q1 = ((uint64_t)a2 << 32) | a3, q2 = q0 * 3, q3 = q1 + 9;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (OP == 0) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a3) : "v"(b)); }
            if (OP == 1) { asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a3) : "v"(b)); }
            if (OP == 2) { asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a3) : "v"(b)); }
            if (OP == 3) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q0) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q1) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q2) : "v"(b), "v"(c) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q3) : "v"(b), "v"(c) : "vcc"); }
            if (OP == 4) { asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(q0)); asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(q1)); asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(q2)); asm volatile("v_lshrrev_b64 %0, 7, %0" : "+v"(q3)); }
            if (OP == 5) { asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a0) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a1) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a2) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a3) : "v"(b)); }
            if (OP == 6) { asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a3) : "v"(b)); }
            if (OP == 7) { asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q0) : "v"(q3)); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q1) : "v"(q3)); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q2) : "v"(q3)); asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q0) : "v"(q3)); }
            if (OP == 8) { asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c)); }
            if (OP == 9) { asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c)); }
            if (OP == 10) { asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(q0)); asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(q1)); asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(q2)); asm volatile("v_lshlrev_b64 %0, 7, %0" : "+v"(q3)); }
            if (OP == 11) { asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q0), "v"(q1) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q2), "v"(q3) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q0), "v"(q3) : "vcc"); asm volatile("v_cmp_lt_u64 vcc, %0, %1" :: "v"(q1), "v"(q2) : "vcc"); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)q0 ^ (uint32_t)q1 ^ (uint32_t)q2 ^ (uint32_t)q3;
}
template <int OP> void run(const char *name, uint32_t *d)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int blocks = 256 * 8;                     // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(a); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * ITERS * REP;       // wave-instructions
    double per_simd = winstr / (256.0 * 4);                 // per SIMD
    printf("%-16s %8.3f ms  -> %.2f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_xor_b32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_u32", d); run<3>("v_mad_u64_u32", d); run<4>("v_lshrrev_b64", d);
    run<10>("v_lshlrev_b64", d); run<5>("v_alignbit_b32", d); run<6>("v_mul_u32_u24", d); run<9>("v_mad_u32_u24", d); run<7>("v_lshl_add_u64", d); run<8>("v_add3_u32", d); run<11>("v_cmp_lt_u64", d);
    return 0;
}
