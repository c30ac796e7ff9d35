// gs_hamming.hip — batched DistHamming::eval (anndists; bound at /root/reference/src/dna/dnasketch.rs:72,139,
// direct use src/bin/bindash.rs:93-99,120-157). dist = (f32)#{a[i] != b[i]} / (f32)m  (SPEC.md 4).
//
//  * k_hamming_qxc  : dense Q x C tile kernel (bindash-style all-pairs, exact top-k ground truth, batch-mate
//                     distances of the index build). 64x64 output tile per workgroup, 4x4 per lane, K-chunks of
//                     32 words staged through LDS; every candidate word is fetched once per 64 queries.
//  * k_hamming_pairs: one wavefront per (a,b) pair, 16-byte coalesced row reads, v_cmp + ballot/popcount.
#include "gs_internal.hpp"

namespace gs {

template <int KIND> struct ElemCmp;
template <> struct ElemCmp<GS_KIND_F32> { static constexpr int EW = 1; };
template <> struct ElemCmp<GS_KIND_U32> { static constexpr int EW = 1; };
template <> struct ElemCmp<GS_KIND_U64> { static constexpr int EW = 2; };

template <int KIND>
__device__ __forceinline__ uint32_t word_ne(uint32_t a, uint32_t b)
{
    if (KIND == GS_KIND_F32) return __uint_as_float(a) != __uint_as_float(b) ? 1u : 0u;   // float `!=` semantics
    return a != b ? 1u : 0u;
}

constexpr int HT = 64;      // tile edge (queries and candidates)
constexpr int HKW = 32;     // words per K-chunk

template <int KIND>
__global__ __launch_bounds__(256) void k_hamming_qxc(const uint32_t *__restrict__ Q, uint64_t nq, uint64_t strideQ, const uint32_t *__restrict__ C, uint64_t nc,
                                                      uint64_t strideC, uint32_t m, float *__restrict__ out, uint32_t *__restrict__ out_cnt)
{
    constexpr int EW = ElemCmp<KIND>::EW;
    __shared__ uint32_t sq[HT][HKW + 1];
    __shared__ uint32_t sc[HT][HKW + 1];
    const uint64_t q0 = (uint64_t)blockIdx.y * HT, c0 = (uint64_t)blockIdx.x * HT;
    const uint32_t tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const uint64_t roww = (uint64_t)m * EW;          // words per row
    uint32_t cnt[4][4] = {};
    for (uint64_t w0 = 0; w0 < roww; w0 += HKW) {
        // stage 64 query rows + 64 candidate rows x 32 words; rows past the end are clamped (results discarded),
        // words past the row end read as 0 on both sides (equal)
        const uint32_t lw = threadIdx.x & 31, lr = threadIdx.x >> 5;
        const bool wv = (w0 + lw) < roww;
#pragma unroll
        for (int i = 0; i < HT / 8; i++) {
            uint32_t r = lr + 8 * i;
            uint64_t qr = q0 + r < nq ? q0 + r : nq - 1;
            uint64_t cr = c0 + r < nc ? c0 + r : nc - 1;
            sq[r][lw] = wv ? Q[qr * strideQ + w0 + lw] : 0u;
            sc[r][lw] = wv ? C[cr * strideC + w0 + lw] : 0u;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < HKW; kk += EW) {
            uint32_t a[4][EW], b[4][EW];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int e = 0; e < EW; e++) { a[i][e] = sq[ty * 4 + i][kk + e]; b[i][e] = sc[tx * 4 + i][kk + e]; }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t ne = word_ne<KIND>(a[i][0], b[j][0]);
                    if (EW == 2) ne |= (a[i][EW - 1] != b[j][EW - 1]) ? 1u : 0u;
                    cnt[i][j] += ne;
                }
        }
        __syncthreads();
    }
    const float fm = (float)m;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t qi = q0 + ty * 4 + i;
        if (qi >= nq) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint64_t cj = c0 + tx * 4 + j;
            if (cj < nc) { if (out) out[qi * nc + cj] = (float)cnt[i][j] / fm; else out_cnt[qi * nc + cj] = cnt[i][j]; }
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_hamming_pairs(const uint32_t *__restrict__ A, const uint32_t *__restrict__ B, const uint64_t *__restrict__ ia,
                                                        const uint64_t *__restrict__ ib, uint64_t npairs, uint32_t m, float *__restrict__ out)
{
    constexpr int EW = ElemCmp<KIND>::EW;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t roww = (uint64_t)m * EW;
    for (uint64_t p = wave; p < npairs; p += nwaves) {
        const uint32_t *a = A + ia[p] * roww, *b = B + ib[p] * roww;
        uint32_t cnt = 0;      // wave-uniform
        for (uint64_t e = lane; e < m; e += 64) {
            bool ne;
            if (EW == 1) ne = word_ne<KIND>(a[e], b[e]) != 0;
            else ne = (a[2 * e] != b[2 * e]) || (a[2 * e + 1] != b[2 * e + 1]);
            cnt += (uint32_t)__popcll(__ballot(ne));
        }
        if (lane == 0) out[p] = (float)cnt / (float)m;
    }
}

int hamming_qxc_strided(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, uint64_t strideQ_bytes, const void *C, uint64_t nc,
                        uint64_t strideC_bytes, float *out, uint32_t *out_cnt)
{
    GS_REQUIRE(c && m > 0, GS_ERR_INVALID, "bad argument");
    GS_REQUIRE(kind == GS_KIND_F32 || kind == GS_KIND_U32 || kind == GS_KIND_U64, GS_ERR_UNSUPPORTED, "DistHamming kind %d not on the device path", kind);
    if (nq == 0 || nc == 0) return GS_OK;
    GS_REQUIRE(Q && C && (out || out_cnt), GS_ERR_INVALID, "null argument");
    dim3 grid((uint32_t)((nc + HT - 1) / HT), (uint32_t)((nq + HT - 1) / HT)), block(256);
    GS_REQUIRE(grid.y <= 65535, GS_ERR_INVALID, "too many query rows for one call (max %d)", 65535 * HT);
    ProfScope ps(c, FAM_HAMMING);
    const uint64_t sq = strideQ_bytes / 4, sc = strideC_bytes / 4;
    if (kind == GS_KIND_F32) hipLaunchKernelGGL(k_hamming_qxc<GS_KIND_F32>, grid, block, 0, c->stream, (const uint32_t *)Q, nq, sq, (const uint32_t *)C, nc, sc, m, out, out_cnt);
    else if (kind == GS_KIND_U32) hipLaunchKernelGGL(k_hamming_qxc<GS_KIND_U32>, grid, block, 0, c->stream, (const uint32_t *)Q, nq, sq, (const uint32_t *)C, nc, sc, m, out, out_cnt);
    else hipLaunchKernelGGL(k_hamming_qxc<GS_KIND_U64>, grid, block, 0, c->stream, (const uint32_t *)Q, nq, sq, (const uint32_t *)C, nc, sc, m, out, out_cnt);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}
static int hamming_qxc_dev(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc, float *out)
{
    const uint64_t row = kind_bytes(kind) * (uint64_t)m;
    return hamming_qxc_strided(c, kind, m, Q, nq, row, C, nc, row, out, nullptr);
}

}  // namespace gs

extern "C" {

int gs_hamming_qxc_dev(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc, float *out)
{
    return gs::hamming_qxc_dev(c, kind, m, Q, nq, C, nc, out);
}

int gs_hamming_qxc(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc, float *out)
{
    GS_REQUIRE(c && m > 0, GS_ERR_INVALID, "bad argument");
    if (nq == 0 || nc == 0) return GS_OK;
    GS_REQUIRE(Q && C && out, GS_ERR_INVALID, "null argument");
    GS_HIP_CHECK(hipSetDevice(c->device));
    const size_t row = gs::kind_bytes(kind) * (size_t)m;
    gs::DevBuf dq, dc, dout;
    int rc;
    if ((rc = dq.alloc(row * nq))) return rc;
    if ((rc = dc.alloc(row * nc))) return rc;
    if ((rc = dout.alloc(4 * nq * nc))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(dq.p, Q, row * nq, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dc.p, C, row * nc, hipMemcpyHostToDevice, c->stream));
    if ((rc = gs::hamming_qxc_dev(c, kind, m, dq.p, nq, dc.p, nc, dout.as<float>()))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(out, dout.p, 4 * nq * nc, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

int gs_hamming_pairs(gs_ctx *c, int kind, uint32_t m, const void *A, uint64_t na, const void *B, uint64_t nb, const uint64_t *ia,
                     const uint64_t *ib, uint64_t npairs, float *out)
{
    GS_REQUIRE(c && m > 0, GS_ERR_INVALID, "bad argument");
    GS_REQUIRE(kind == GS_KIND_F32 || kind == GS_KIND_U32 || kind == GS_KIND_U64, GS_ERR_UNSUPPORTED, "DistHamming kind %d not on the device path", kind);
    if (npairs == 0) return GS_OK;
    GS_REQUIRE(A && B && ia && ib && out, GS_ERR_INVALID, "null argument");
    for (uint64_t p = 0; p < npairs; p++) GS_REQUIRE(ia[p] < na && ib[p] < nb, GS_ERR_INVALID, "pair %llu out of range", (unsigned long long)p);
    GS_HIP_CHECK(hipSetDevice(c->device));
    const size_t row = gs::kind_bytes(kind) * (size_t)m;
    gs::DevBuf da, db, dia, dib, dout;
    int rc;
    if ((rc = da.alloc(row * na))) return rc;
    if ((rc = db.alloc(row * nb))) return rc;
    if ((rc = dia.alloc(8 * npairs))) return rc;
    if ((rc = dib.alloc(8 * npairs))) return rc;
    if ((rc = dout.alloc(4 * npairs))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(da.p, A, row * na, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(db.p, B, row * nb, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dia.p, ia, 8 * npairs, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dib.p, ib, 8 * npairs, hipMemcpyHostToDevice, c->stream));
    uint32_t blocks = (uint32_t)((npairs + 3) / 4);
    if (blocks > (uint32_t)c->n_cu * 8) blocks = (uint32_t)c->n_cu * 8;
    {
        gs::ProfScope ps(c, gs::FAM_HAMMING);
        if (kind == GS_KIND_F32) hipLaunchKernelGGL(gs::k_hamming_pairs<GS_KIND_F32>, dim3(blocks), dim3(256), 0, c->stream, da.as<uint32_t>(), db.as<uint32_t>(), dia.as<uint64_t>(), dib.as<uint64_t>(), npairs, m, dout.as<float>());
        else if (kind == GS_KIND_U32) hipLaunchKernelGGL(gs::k_hamming_pairs<GS_KIND_U32>, dim3(blocks), dim3(256), 0, c->stream, da.as<uint32_t>(), db.as<uint32_t>(), dia.as<uint64_t>(), dib.as<uint64_t>(), npairs, m, dout.as<float>());
        else hipLaunchKernelGGL(gs::k_hamming_pairs<GS_KIND_U64>, dim3(blocks), dim3(256), 0, c->stream, da.as<uint32_t>(), db.as<uint32_t>(), dia.as<uint64_t>(), dib.as<uint64_t>(), npairs, m, dout.as<float>());
    }
    GS_HIP_CHECK(hipGetLastError());
    GS_HIP_CHECK(hipMemcpyAsync(out, dout.p, 4 * npairs, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

}  // extern "C"
