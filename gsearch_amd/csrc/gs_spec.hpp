// gs_spec.hpp — SPEC.md section 2 (hashing + random numbers) for device and host code of the product.
// Every [CHOICE] of SPEC.md that touches arithmetic lives in this header (and, independently restated,
// in oracle/gs_oracle.c). Replaces what gsearch reaches through fxhash / rand_xoshiro / rand
// (Cargo.toml:26,122 of the reference; crates not vendored).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GS_HD __host__ __device__ __forceinline__

namespace gs {

GS_HD uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
GS_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// fxhash::FxHasher64 over one integer write
GS_HD uint64_t fx64(uint64_t v) { return v * 0x517cc1b727220a95ULL; }
// fxhash::FxHasher32: 32-bit words, low first
GS_HD uint64_t fx32_w32(uint32_t v) { return (uint64_t)(uint32_t)(v * 0x9e3779b9u); }

enum { ALGO_PROB3A = 0, ALGO_SUPER = 1, ALGO_SUPER2 = 2, ALGO_HLL = 3, ALGO_OPTDENS = 4, ALGO_REVOPTDENS = 5 };

// SPEC 2 table "element hash"
template <int ALGO, int VBITS>
GS_HD uint64_t elem_hash(uint64_t v)
{
    if (ALGO == ALGO_PROB3A) return v;
    if (ALGO == ALGO_HLL) return fx64(v);
    if (ALGO == ALGO_SUPER2 && VBITS == 32) return fx32_w32((uint32_t)v);
    return fx64(v);
}

// one SplitMix64 output for counter value x (already advanced)
GS_HD uint64_t splitmix_mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
#define GS_GAMMA 0x9e3779b97f4a7c15ULL

struct Rng {   // xoshiro256++ seeded through SplitMix64 (rand_xoshiro seed_from_u64)
    uint64_t s0, s1, s2, s3;
    GS_HD void seed(uint64_t x)
    {
        s0 = splitmix_mix(x + GS_GAMMA);
        s1 = splitmix_mix(x + 2 * GS_GAMMA);
        s2 = splitmix_mix(x + 3 * GS_GAMMA);
        s3 = splitmix_mix(x + 4 * GS_GAMMA);
    }
    GS_HD uint64_t next64()
    {
        uint64_t r = rotl64(s0 + s3, 23) + s0;
        uint64_t t = s1 << 17;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t; s3 = rotl64(s3, 45);
        return r;
    }
    GS_HD uint32_t next32() { return (uint32_t)(next64() >> 32); }
    GS_HD uint32_t r23() { return next32() >> 9; }                       // U32f = r23 * 2^-23
    GS_HD double u64f() { return (double)(next64() >> 12) * 0x1.0p-52; }  // U64f
};

GS_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(a, b);
#else
    return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// rand 0.8 UniformInt<usize>::sample for range [0,n): zone = 2^64-1 - (2^64 mod n)
GS_HD uint64_t uint_zone(uint64_t n) { return ~(uint64_t)0 - ((0 - n) % n); }
GS_HD uint64_t rng_uint(Rng &g, uint64_t n, uint64_t zone)
{
    for (;;) {
        uint64_t x = g.next64();
        uint64_t lo = x * n;
        if (lo <= zone) return mulhi64(x, n);
    }
}

// ---- the two-draw fast path of optdens (SPEC 3.1): r = U32f bits, b = Uint(m) ---------------------
// Only s0,s1,s3 are needed for two outputs; the full generator is re-run in the (probability m/2^64)
// rejection case so that the result is exactly the sequential definition.
// two_draw: first output o1 (the r draw: U32f bits = o1>>41, next32 = o1>>32, next64 = o1) and b = Uint(m) from the second.
GS_HD void two_draw(uint64_t h, uint32_t m, uint64_t zone, uint64_t &o1, uint32_t &bin)
{
    uint64_t s0 = splitmix_mix(h + GS_GAMMA);
    uint64_t s1 = splitmix_mix(h + 2 * GS_GAMMA);
    uint64_t s3 = splitmix_mix(h + 4 * GS_GAMMA);
    o1 = rotl64(s0 + s3, 23) + s0;
    uint64_t n3 = s3 ^ s1;           // s3 after the first step, before rotation
    uint64_t n0 = s0 ^ n3;           // s0 after the first step
    uint64_t o2 = rotl64(n0 + rotl64(n3, 45), 23) + n0;
    // Uint(g, m) with m < 2^32: the 96-bit product o2 * m from two 32 x 32 -> 64 multiply-adds. Its top 32 bits are the draw, its low 64
    // bits the rejection test `lo <= zone` - and zone >= 2^64 - 2^32, so a rejection needs the upper half of lo to be all ones: one
    // 32-bit compare on the hot path, the exact 64-bit test (and the re-run of the full generator) only behind it.
    const uint64_t t = (uint64_t)(uint32_t)o2 * m;
    const uint64_t u = (uint64_t)(uint32_t)(o2 >> 32) * m + (t >> 32);
    bin = (uint32_t)(u >> 32);
    if (__builtin_expect((uint32_t)u != 0xFFFFFFFFu, 1)) return;
    const uint64_t lo = (u << 32) | (uint32_t)t;
    if (lo <= zone) return;
    Rng g; g.seed(h); (void)g.next64();
    bin = (uint32_t)rng_uint(g, (uint64_t)m, zone);
}
GS_HD void oph_draw(uint64_t h, uint32_t m, uint64_t zone, uint32_t &r23, uint32_t &bin)
{
    uint64_t o1;
    two_draw(h, m, zone, o1, bin);
    r23 = (uint32_t)(o1 >> 41);
}

// ---- SPEC 2 LN / TEXP, SPEC 3.4 hll (SetSketch1): natural logarithm from IEEE + - * / only (no fma contraction: the library is
// built with -ffp-contract=off), so the host oracle and the device evaluate the same instruction sequence
GS_HD double spec_ln(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t bits = (uint64_t)__double_as_longlong(x);
#else
    uint64_t bits; __builtin_memcpy(&bits, &x, 8);
#endif
    long long e = (long long)((bits >> 52) & 0x7FF) - 1023;
    const uint64_t mb = (bits & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL;
#if defined(__HIP_DEVICE_COMPILE__)
    double t = __longlong_as_double((long long)mb);
#else
    double t; __builtin_memcpy(&t, &mb, 8);
#endif
    if (t > 1.4142135623730951) { t = t * 0.5; e += 1; }
    const double s = (t - 1.0) / (t + 1.0), z = s * s;
    double p = 1.0 / 23.0;
    p = p * z + 1.0 / 21.0; p = p * z + 1.0 / 19.0; p = p * z + 1.0 / 17.0; p = p * z + 1.0 / 15.0; p = p * z + 1.0 / 13.0;
    p = p * z + 1.0 / 11.0; p = p * z + 1.0 / 9.0; p = p * z + 1.0 / 7.0; p = p * z + 1.0 / 5.0; p = p * z + 1.0 / 3.0; p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}
#define GS_HLL_B 1.001
#define GS_HLL_A 20.0
#define GS_HLL_Q 65534u
GS_HD uint32_t hll_k(double x, double inv_lnb)
{
    if (!(x > 0.0)) return GS_HLL_Q + 1;
    const double y = 1.0 - spec_ln(x) * inv_lnb;
    if (y < 0.0) return 0;
    if (y >= (double)(GS_HLL_Q + 1)) return GS_HLL_Q + 1;
    return (uint32_t)y;
}

}  // namespace gs
