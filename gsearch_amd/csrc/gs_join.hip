// gs_join.hip — batched DistHamming as an equi-join ("match-join").
//
// DistHamming::eval (anndists; /root/reference/src/dna/dnasketch.rs:72, src/bin/bindash.rs:93-99) counts MISmatches:
//   c(q,e) = m - #{s : q[s] == e[s]}.
// Sketches of unrelated genomes agree in ~0.5 of 18000 slots, so the matches are ~10^4 times rarer than the mismatches the
// tile kernel has to touch. With a column-major copy of the database (cols[s][e]) the matches of a whole query batch are the
// equi-join, slot by slot, of the batch's column (sorted once per batch) with the database column (streamed once per batch):
//   per slot s:  for every node e:  binary-search cols[s][e] in the sorted query values; every hit (q,e) -> matches[q][e] += 1
// HBM traffic: the database once per BATCH (21.6 GB for 300 k x 18000 f32) instead of once per 128 queries; arithmetic: one
// 12-step LDS binary search per (slot, node) instead of nq compares. Output is bit-identical to the tile kernel.
#include <hipcub/hipcub.hpp>
#include "gs_internal.hpp"

namespace gs {

template <typename T> struct ElemKey;
template <> struct ElemKey<uint32_t> { static constexpr bool F = false; };

// canonical key of an element under the element type's `==`: f32: -0 -> +0 (NaN handled by the caller), integers: identity
template <int KIND, typename T>
__device__ __forceinline__ T canon(T v)
{
    if (KIND == GS_KIND_F32) return (v == (T)0x80000000u) ? (T)0 : v;
    return v;
}
template <int KIND, typename T>
__device__ __forceinline__ bool never_equal(T v)      // f32 NaN != anything, itself included
{
    if (KIND == GS_KIND_F32) return ((uint32_t)v & 0x7FFFFFFFu) > 0x7F800000u;
    return false;
}

// rows (row-major, strided) -> columns: cols[s * colcap + first + i] = rows[i][s]   (32 x 32 LDS transpose)
template <typename T>
__global__ __launch_bounds__(256) void k_rows_to_cols(const uint8_t *__restrict__ rows, uint64_t stride, uint64_t nrows, uint32_t m, T *__restrict__ cols,
                                                       uint64_t colcap, uint64_t first)
{
    __shared__ T tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const uint64_t r0 = (uint64_t)blockIdx.x * 32, s0 = (uint64_t)blockIdx.y * 32;
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint64_t r = r0 + j, s = s0 + tx;
        tile[j][tx] = (r < nrows && s < m) ? ((const T *)(rows + r * stride))[s] : (T)0;
    }
    __syncthreads();
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint64_t s = s0 + j, r = r0 + tx;
        if (s < m && r < nrows) cols[s * colcap + first + r] = tile[tx][j];
    }
}

// query batch -> per-slot (key, query index) lists: qkey[s * nq + q], qidx[s * nq + q]
template <int KIND, typename T>
__global__ __launch_bounds__(256) void k_query_cols(const uint8_t *__restrict__ rows, uint64_t stride, uint32_t nq, uint32_t m, T *__restrict__ qkey, uint32_t *__restrict__ qidx)
{
    __shared__ T tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint64_t r0 = (uint64_t)blockIdx.x * 32, s0 = (uint64_t)blockIdx.y * 32;
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint64_t r = r0 + j, s = s0 + tx;
        tile[j][tx] = (r < nq && s < m) ? canon<KIND, T>(((const T *)(rows + r * stride))[s]) : (T)0;
    }
    __syncthreads();
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint64_t s = s0 + j, r = r0 + tx;
        if (s < m && r < nq) { qkey[s * nq + r] = tile[tx][j]; qidx[s * nq + r] = (uint32_t)r; }
    }
}
__global__ void k_seg_offsets(uint64_t *off, uint32_t m, uint32_t nq)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= m) off[i] = (uint64_t)i * nq;
}

constexpr int JQ_MAX = 4096;      // queries per join call (LDS: sorted keys + indices)
constexpr int JT = 256;

// grid: (node chunks, slots). matches[q * ld + e] (16-bit counters, incremented through their 32-bit container)
template <int KIND, typename T>
__global__ __launch_bounds__(JT) void k_match_join(const T *__restrict__ qkey, const uint32_t *__restrict__ qidx, uint32_t nq, const T *__restrict__ cols,
                                                    uint64_t colcap, uint64_t n, uint64_t chunk, uint32_t *__restrict__ mm32, uint64_t ld)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    T *skey = (T *)s_raw;
    uint16_t *sq = (uint16_t *)(s_raw + sizeof(T) * (size_t)nq);
    const uint64_t s = blockIdx.y;
    for (uint32_t i = threadIdx.x; i < nq; i += JT) { skey[i] = qkey[s * nq + i]; sq[i] = (uint16_t)qidx[s * nq + i]; }
    __syncthreads();
    const uint64_t e0 = (uint64_t)blockIdx.x * chunk, e1 = e0 + chunk < n ? e0 + chunk : n;
    const T *col = cols + s * colcap;
    for (uint64_t e = e0 + threadIdx.x; e < e1; e += JT) {
        T v = col[e];
        if (never_equal<KIND, T>(v)) continue;
        v = canon<KIND, T>(v);
        uint32_t lo = 0, hi = nq;
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (skey[mid] < v) lo = mid + 1; else hi = mid; }
        for (; lo < nq && skey[lo] == v; lo++) {
            const uint64_t idx = (uint64_t)sq[lo] * ld + e;
            atomicAdd(&mm32[idx >> 1], (idx & 1) ? 0x10000u : 1u);
        }
    }
}
// matches -> mismatch counts, in place:  c = m - matches
__global__ void k_match_to_count(uint16_t *mm, uint64_t nq, uint64_t n, uint64_t ld, uint32_t m)
{
    const uint64_t total = nq * n;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t q = i / n, e = i % n;
        mm[q * ld + e] = (uint16_t)(m - mm[q * ld + e]);
    }
}

template <int KIND, typename T>
static int join_impl(gs_ctx *c, uint32_t m, const uint8_t *qrows, uint64_t qstride, uint32_t nq, const void *cols, uint64_t colcap, uint64_t n, uint16_t *out16,
                     uint64_t ld, DevBuf *scratch /* [5] reusable */)
{
    int rc;
    const size_t items = (size_t)m * nq;
    DevBuf &k0 = scratch[0], &k1 = scratch[1], &i0 = scratch[2], &i1 = scratch[3], &tmp = scratch[4];
    if ((rc = k0.ensure(sizeof(T) * items))) return rc;
    if ((rc = k1.ensure(sizeof(T) * items))) return rc;
    if ((rc = i0.ensure(4 * items))) return rc;
    if ((rc = i1.ensure(4 * items))) return rc;
    DevBuf off;
    if ((rc = off.alloc(8 * ((size_t)m + 1)))) return rc;
    // zero the 16-bit match counters of the used rows (ld may exceed n: only [0,n) of each row is used)
    GS_HIP_CHECK(hipMemset2DAsync(out16, ld * 2, 0, n * 2, nq, c->stream));
    dim3 tg((nq + 31) / 32, (m + 31) / 32);
    hipLaunchKernelGGL((k_query_cols<KIND, T>), tg, dim3(256), 0, c->stream, qrows, qstride, nq, m, k0.as<T>(), i0.as<uint32_t>());
    hipLaunchKernelGGL(k_seg_offsets, dim3((m + 256) / 256), dim3(256), 0, c->stream, off.as<uint64_t>(), m, nq);
    GS_HIP_CHECK(hipGetLastError());
    size_t tb = 0;
    GS_HIP_CHECK(hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, tb, k0.as<T>(), k1.as<T>(), i0.as<uint32_t>(), i1.as<uint32_t>(), (int)items, (int)m,
                                                             off.as<uint64_t>(), off.as<uint64_t>() + 1, 0, (int)(8 * sizeof(T)), c->stream));
    if ((rc = tmp.ensure(tb))) return rc;
    GS_HIP_CHECK(hipcub::DeviceSegmentedRadixSort::SortPairs(tmp.p, tb, k0.as<T>(), k1.as<T>(), i0.as<uint32_t>(), i1.as<uint32_t>(), (int)items, (int)m,
                                                             off.as<uint64_t>(), off.as<uint64_t>() + 1, 0, (int)(8 * sizeof(T)), c->stream));
    const uint64_t chunk = 32768;
    dim3 jg((uint32_t)((n + chunk - 1) / chunk), m);
    const size_t lds = (sizeof(T) + 2) * (size_t)nq + 16;
    {
        ProfScope ps(c, FAM_HAMMING);
        auto kern = k_match_join<KIND, T>;
        if (lds > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, jg, dim3(JT), lds, c->stream, k1.as<T>(), i1.as<uint32_t>(), nq, (const T *)cols, colcap, n, chunk, (uint32_t *)out16, ld);
        GS_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_match_to_count, dim3(c->n_cu * 8), dim3(256), 0, c->stream, out16, (uint64_t)nq, n, ld, m);
    GS_HIP_CHECK(hipGetLastError());
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));       // `off` is freed on return
    return GS_OK;
}

int match_join_counts(gs_ctx *c, int kind, uint32_t m, const void *qrows, uint64_t qstride, uint64_t nq, const void *cols, uint64_t colcap, uint64_t n,
                      uint16_t *out16, uint64_t ld, DevBuf *scratch)
{
    GS_REQUIRE(nq >= 1 && nq <= (uint64_t)JQ_MAX && m <= 65535 && (ld % 2) == 0 && ((uintptr_t)out16 % 4) == 0, GS_ERR_INVALID, "match_join_counts: bad shape");
    GS_REQUIRE((uint64_t)m * nq < ((uint64_t)1 << 31), GS_ERR_INVALID, "match_join_counts: batch too large");
    if (kind == GS_KIND_U64) return join_impl<GS_KIND_U64, uint64_t>(c, m, (const uint8_t *)qrows, qstride, (uint32_t)nq, cols, colcap, n, out16, ld, scratch);
    if (kind == GS_KIND_F32) return join_impl<GS_KIND_F32, uint32_t>(c, m, (const uint8_t *)qrows, qstride, (uint32_t)nq, cols, colcap, n, out16, ld, scratch);
    return join_impl<GS_KIND_U32, uint32_t>(c, m, (const uint8_t *)qrows, qstride, (uint32_t)nq, cols, colcap, n, out16, ld, scratch);
}

int rows_to_cols(gs_ctx *c, int kind, uint32_t m, const void *rows, uint64_t stride, uint64_t nrows, void *cols, uint64_t colcap, uint64_t first)
{
    if (nrows == 0) return GS_OK;
    dim3 g((uint32_t)((nrows + 31) / 32), (m + 31) / 32);
    if (kind == GS_KIND_U64) hipLaunchKernelGGL((k_rows_to_cols<uint64_t>), g, dim3(256), 0, c->stream, (const uint8_t *)rows, stride, nrows, m, (uint64_t *)cols, colcap, first);
    else hipLaunchKernelGGL((k_rows_to_cols<uint32_t>), g, dim3(256), 0, c->stream, (const uint8_t *)rows, stride, nrows, m, (uint32_t *)cols, colcap, first);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
