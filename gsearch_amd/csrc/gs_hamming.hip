// gs_hamming.hip — batched DistHamming::eval (anndists; bound at /root/reference/src/dna/dnasketch.rs:72,139,
// direct use src/bin/bindash.rs:93-99,120-157). dist = (f32)#{a[i] != b[i]} / (f32)m  (SPEC.md 4).
//
//  * k_hamming_qxc  : dense Q x C tile kernel (bindash-style all-pairs, exact top-k ground truth, batch-mate
//                     distances of the index build). 64x64 output tile per workgroup, 4x4 per lane, K-chunks of
//                     32 words staged through LDS; every candidate word is fetched once per 64 queries.
//  * k_hamming_pairs: one wavefront per (a,b) pair, 16-byte coalesced row reads, v_cmp + ballot/popcount.
#include <algorithm>
#include "gs_internal.hpp"
#include <type_traits>

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

// ---- dense tile kernel ------------------------------------------------------------------------------
// 128 x 128 outputs per workgroup (256 lanes, 8 x 8 per lane), K-chunks of 32 words double-buffered through LDS.
// Per pair-element: one v_cmp_ne + one v_addc_co (the carry-in IS the mismatch bit) — 2 VALU instructions, which is
// what bounds the kernel (it reads every candidate word once per 128 queries, so HBM traffic is ~1/128 of the
// algorithmic bytes). Lane (tx,ty) owns query rows ty+16i and candidate rows tx+16j: with the 33-word LDS row
// pitch every ds_read_b32 of a wavefront is conflict-free (distinct banks across tx, broadcast across ty).
constexpr int HT = 128;     // tile edge (queries and candidates)
constexpr int HKW = 32;     // words per K-chunk
constexpr int HP = HKW + 1; // LDS row pitch in words

// one query value against 8 candidate values: 8 x (v_cmp_ne ; v_addc_co). A single asm statement so that hipcc does
// not put a hazard s_nop between the pairs (it cannot see inside asm); VALU->VALU VCC forwarding needs no wait state.
#define GS_CMP8(OP)                                                                                                   \
    asm volatile(OP " vcc, %8, %9\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"                                           \
                 OP " vcc, %8, %10\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"                                          \
                 OP " vcc, %8, %11\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"                                          \
                 OP " vcc, %8, %12\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"                                          \
                 OP " vcc, %8, %13\n\tv_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"                                          \
                 OP " vcc, %8, %14\n\tv_addc_co_u32 %5, vcc, 0, %5, vcc\n\t"                                          \
                 OP " vcc, %8, %15\n\tv_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"                                          \
                 OP " vcc, %8, %16\n\tv_addc_co_u32 %7, vcc, 0, %7, vcc"                                              \
                 : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7])       \
                 : "v"(a), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7])       \
                 : "vcc")
template <int KIND>
__device__ __forceinline__ void cmp_acc8(uint32_t (&c)[8], uint32_t a, const uint32_t (&b)[8])
{
    if (KIND == GS_KIND_F32) GS_CMP8("v_cmp_neq_f32");
    else GS_CMP8("v_cmp_ne_u32");
}
__device__ __forceinline__ void cmp_acc8_64(uint32_t (&c)[8], uint64_t a, const uint64_t (&b)[8])
{
    GS_CMP8("v_cmp_ne_u64");
}

// IDX: the tiles of a work list instead of a grid over Q x C (the heavy blocks of the match-join, gs_join.hip): item = {first entry and length
// of its query rows in qlist, first entry and length of its candidate rows in clist}; the rows are Q + qlist[.] * strideQ / C + clist[.] *
// strideC and the counts go to out_cnt16[qlist[.] * ld_out + clist[.]].
template <int KIND, bool VEC4, bool IDX>
__global__ __launch_bounds__(256) void k_hamming_qxc(const uint32_t *__restrict__ Q, uint64_t nq, uint64_t strideQ, const uint32_t *__restrict__ C, uint64_t nc,
                                                      uint64_t strideC, uint32_t m, float *__restrict__ out, uint32_t *__restrict__ out_cnt, uint16_t *__restrict__ out_cnt16, uint64_t ld_out,
                                                      uint32_t ksplit_words, const uint4 *__restrict__ items, const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ clist)
{
    constexpr int EW = ElemCmp<KIND>::EW;
    __shared__ uint32_t sq[2][HT * HP];
    __shared__ uint32_t sc[2][HT * HP];
    // query tiles on the FAST grid dimension: blocks that share a candidate tile are dispatched together, so the candidate rows
    // come from HBM once and from L2 / Infinity Cache for the other query tiles
    uint64_t q0 = (uint64_t)blockIdx.x * HT, c0 = (uint64_t)blockIdx.y * HT;
    if (IDX) { const uint4 it = items[blockIdx.x]; q0 = it.x; nq = q0 + it.y; c0 = it.z; nc = c0 + it.w; }      // (list positions; at most HT rows each)
    const uint32_t tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int ni = IDX ? (int)((nq - q0 + 15) / 16) : 8;           // (workgroup-uniform) see THIN below
    const uint64_t roww_full = (uint64_t)m * EW;     // words per row
    // K-split (small Q x C problems): blockIdx.z owns words [kbeg, roww); partial counts are added atomically (zeroed out_cnt)
    const uint64_t kbeg = ksplit_words ? (uint64_t)blockIdx.z * ksplit_words : 0;
    const uint64_t roww = ksplit_words ? (kbeg + ksplit_words < roww_full ? kbeg + ksplit_words : roww_full) : roww_full;
    uint32_t cnt[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) cnt[i][j] = 0;

    // staging: 128 rows x 32 words per operand = 1024 x 16 B; lane handles 4 + 4 of them
    uint4 rq[4], rc[4];
    auto stage_load = [&](uint64_t w0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t idx = threadIdx.x + 256 * i, r = idx >> 3, kq = (idx & 7) * 4;
            uint64_t qr = q0 + r < nq ? q0 + r : nq - 1, cr = c0 + r < nc ? c0 + r : nc - 1;   // clamped rows: results discarded
            if (IDX) { qr = qlist[qr]; cr = clist[cr]; }
            const uint32_t *pq = Q + qr * strideQ + w0 + kq, *pc = C + cr * strideC + w0 + kq;
            if (VEC4 && w0 + kq + 4 <= roww) { rq[i] = *(const uint4 *)pq; rc[i] = *(const uint4 *)pc; }
            else {
                uint32_t a[4], b[4];
#pragma unroll
                for (int e = 0; e < 4; e++) { const bool v = w0 + kq + e < roww; a[e] = v ? pq[e] : 0u; b[e] = v ? pc[e] : 0u; }   // past the row end: equal on both sides
                rq[i] = make_uint4(a[0], a[1], a[2], a[3]); rc[i] = make_uint4(b[0], b[1], b[2], b[3]);
            }
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t idx = threadIdx.x + 256 * i, r = idx >> 3, kq = (idx & 7) * 4;
            uint32_t *dq = &sq[buf][r * HP + kq], *dc = &sc[buf][r * HP + kq];
            dq[0] = rq[i].x; dq[1] = rq[i].y; dq[2] = rq[i].z; dq[3] = rq[i].w;
            dc[0] = rc[i].x; dc[1] = rc[i].y; dc[2] = rc[i].z; dc[3] = rc[i].w;
        }
    };
    stage_load(kbeg);
    stage_store(0);
    __syncthreads();
    int cur = 0;
    for (uint64_t w0 = kbeg; w0 < roww; w0 += HKW) {
        const bool more = w0 + HKW < roww;
        if (more) stage_load(w0 + HKW);              // global loads in flight during the compare block
        const uint32_t *pa = &sq[cur][ty * HP], *pb = &sc[cur][tx * HP];
        // THIN (work-list tiles with fewer than 113 query rows: a cluster of five isolates against its 2000 genomes is sixteen tiles of FIVE rows): lane (tx, ty) owns
        // the query rows ty + 16 i, so only i < ni = ceil(rows / 16) hold real rows - the others (clamped copies) are neither read nor compared. A skewed database's
        // blocks are mostly such tiles (bench leg request_skewed: 230 clusters of 5.8 queries on average per batch): 20 us per tile whatever its height before.
        auto kblock = [&](auto thin_tag) {
            constexpr bool THIN = decltype(thin_tag)::value;
#pragma unroll 4
            for (int kk = 0; kk < HKW; kk += EW) {
                if (EW == 1) {
                    uint32_t a[8], b[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) { if (!THIN || i < ni) a[i] = pa[i * 16 * HP + kk]; b[i] = pb[i * 16 * HP + kk]; }
#pragma unroll
                    for (int i = 0; i < 8; i++) if (!THIN || i < ni) cmp_acc8<KIND>(cnt[i], a[i], b);
                } else {
                    uint64_t a[8], b[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        if (!THIN || i < ni) a[i] = (uint64_t)pa[i * 16 * HP + kk] | ((uint64_t)pa[i * 16 * HP + kk + 1] << 32);
                        b[i] = (uint64_t)pb[i * 16 * HP + kk] | ((uint64_t)pb[i * 16 * HP + kk + 1] << 32);
                    }
#pragma unroll
                    for (int i = 0; i < 8; i++) if (!THIN || i < ni) cmp_acc8_64(cnt[i], a[i], b);
                }
            }
        };
        if (IDX && ni < 8) kblock(std::true_type{}); else kblock(std::false_type{});
        if (more) stage_store(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    const float fm = (float)m;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t qi = q0 + ty + 16 * i;
        if (qi >= nq) continue;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const uint64_t cj = c0 + tx + 16 * j;
            if (cj < nc) {
                if (out) out[qi * ld_out + cj] = (float)cnt[i][j] / fm;
                else if (out_cnt) { if (ksplit_words) atomicAdd(&out_cnt[qi * ld_out + cj], cnt[i][j]); else out_cnt[qi * ld_out + cj] = cnt[i][j]; }
                else if (IDX) {
                    const uint64_t idx = (uint64_t)qlist[qi] * ld_out + clist[cj];
                    if (!ksplit_words) out_cnt16[idx] = (uint16_t)cnt[i][j];
                    else {          // K-split: the counter was set to m (k_blocks_set); this part takes its MATCHES off, through the 32-bit container
                        const uint32_t matches = (uint32_t)((roww - kbeg) / EW) - cnt[i][j];
                        if (matches) atomicSub((uint32_t *)out_cnt16 + (idx >> 1), matches << ((idx & 1) * 16));
                    }
                }
                else out_cnt16[qi * ld_out + cj] = (uint16_t)cnt[i][j];
            }
        }
    }
}

// ---- thin blocks --------------------------------------------------------------------------------------
// Work-list items with at most HTQ query rows (a cluster of a few isolates against the hundreds or thousands of genomes of its family: most blocks of a skewed
// database, bench leg request_skewed). The tile kernel above spends a 128 x 128 tile's LDS traffic on them whatever their height (its THIN path: 9 LDS reads per 16
// VALU instructions); here nothing goes through LDS: a wavefront takes HTR candidate rows, lane l the words 4 l .. 4 l + 3 of every 256-word step of
// those rows (one 16-byte load each, coalesced) and of each query row (the cluster's <= 16 rows: re-read by every pass, L2-resident), 2 VALU instructions per
// (query, candidate) word pair into per-lane counters, one wavefront reduction at the end. 4-byte elements with 16-byte aligned rows only (hamming_blocks).
constexpr int HTQ = 16;     // query rows of a thin item at most
constexpr int HTR = 4;      // candidate rows per wavefront
constexpr int HTS = HT / (4 * HTR);     // workgroups per item
#define GS_CMP4(OP)                                                                                                   \
    asm volatile(OP " vcc, %4, %5\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"                                           \
                 OP " vcc, %4, %6\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"                                           \
                 OP " vcc, %4, %7\n\tv_addc_co_u32 %2, vcc, 0, %2, vcc\n\t"                                           \
                 OP " vcc, %4, %8\n\tv_addc_co_u32 %3, vcc, 0, %3, vcc"                                               \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)                                                              \
                 : "v"(a), "v"(b0), "v"(b1), "v"(b2), "v"(b3)                                                          \
                 : "vcc")
template <int KIND>
__device__ __forceinline__ void cmp_acc4(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t a, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3)
{
    if (KIND == GS_KIND_F32) GS_CMP4("v_cmp_neq_f32");
    else GS_CMP4("v_cmp_ne_u32");
}
template <int KIND>
__global__ __launch_bounds__(256) void k_hamming_thin(const uint32_t *__restrict__ Q, uint64_t strideQ, const uint32_t *__restrict__ C, uint64_t strideC, uint32_t m,
                                                       uint16_t *__restrict__ out_cnt16, uint64_t ld_out, const uint4 *__restrict__ items, const uint32_t *__restrict__ qlist,
                                                       const uint32_t *__restrict__ clist)
{
    // grid: HTS workgroups per item, each with 4 x HTR = 16 of its candidate rows - a few hundred items of 128 rows would leave the chip at one or two workgroups per CU,
    // and the kernel has nothing but other wavefronts to hide its load latency behind
    const uint4 it = items[blockIdx.x / HTS];                     // {first entry of qlist, query rows (<= HTQ), first entry of clist, candidate rows (<= 128)}
    const uint32_t nqr = it.y, ncr = it.w, lane = threadIdx.x & 63, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t steps = (m + 255) / 256;
    {
        const uint32_t r0 = (blockIdx.x % HTS) * 4 * HTR + wv * HTR;   // (wavefront-uniform) this wavefront's candidate rows of the item
        if (r0 >= ncr) return;
        uint32_t cnt[HTR][HTQ];
#pragma unroll
        for (int r = 0; r < HTR; r++)
#pragma unroll
            for (int q = 0; q < HTQ; q++) cnt[r][q] = 0;
        const uint32_t *cp[HTR];
#pragma unroll
        for (int r = 0; r < HTR; r++) cp[r] = C + (uint64_t)clist[it.z + (r0 + r < ncr ? r0 + r : ncr - 1)] * strideC;      // (clamped rows: results discarded)
        for (uint32_t s = 0; s < steps; s++) {
            const uint32_t w = s * 256 + lane * 4;
            const bool in = w + 4 <= m;                            // (m is a multiple of 4 here or the last lanes fall back to word loads below)
            uint4 cb[HTR];
#pragma unroll
            for (int r = 0; r < HTR; r++) {
                if (in) cb[r] = *(const uint4 *)(cp[r] + w);
                else { cb[r].x = w < m ? cp[r][w] : 0u; cb[r].y = w + 1 < m ? cp[r][w + 1] : 0u; cb[r].z = w + 2 < m ? cp[r][w + 2] : 0u; cb[r].w = 0u; }      // past the row end: equal on both sides
            }
#pragma unroll
            for (int q = 0; q < HTQ; q++) {
                if ((uint32_t)q < nqr) {                           // (workgroup-uniform)
                    const uint32_t *qp = Q + (uint64_t)qlist[it.x + q] * strideQ;
                    uint4 qa;
                    if (in) qa = *(const uint4 *)(qp + w);
                    else { qa.x = w < m ? qp[w] : 0u; qa.y = w + 1 < m ? qp[w + 1] : 0u; qa.z = w + 2 < m ? qp[w + 2] : 0u; qa.w = 0u; }
                    cmp_acc4<KIND>(cnt[0][q], cnt[1][q], cnt[2][q], cnt[3][q], qa.x, cb[0].x, cb[1].x, cb[2].x, cb[3].x);
                    cmp_acc4<KIND>(cnt[0][q], cnt[1][q], cnt[2][q], cnt[3][q], qa.y, cb[0].y, cb[1].y, cb[2].y, cb[3].y);
                    cmp_acc4<KIND>(cnt[0][q], cnt[1][q], cnt[2][q], cnt[3][q], qa.z, cb[0].z, cb[1].z, cb[2].z, cb[3].z);
                    cmp_acc4<KIND>(cnt[0][q], cnt[1][q], cnt[2][q], cnt[3][q], qa.w, cb[0].w, cb[1].w, cb[2].w, cb[3].w);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < HTQ; q++) {
            if ((uint32_t)q < nqr) {
                const uint64_t orow = (uint64_t)qlist[it.x + q] * ld_out;
#pragma unroll
                for (int r = 0; r < HTR; r++) {
                    uint32_t v = cnt[r][q];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
                    if (lane == 0 && r0 + r < ncr) out_cnt16[orow + clist[it.z + r0 + r]] = (uint16_t)v;
                }
            }
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
                        uint64_t strideC_bytes, float *out, uint32_t *out_cnt, uint16_t *out_cnt16, uint64_t ld_out)
{
    GS_REQUIRE(c && m > 0, GS_ERR_INVALID, "bad argument");
    GS_REQUIRE(kind == GS_KIND_F32 || kind == GS_KIND_U32 || kind == GS_KIND_U64, GS_ERR_UNSUPPORTED, "DistHamming kind %d not on the device path", kind);
    if (nq == 0 || nc == 0) return GS_OK;
    GS_REQUIRE(Q && C && (out || out_cnt || out_cnt16), GS_ERR_INVALID, "null argument");
    GS_REQUIRE(!out_cnt16 || m <= 65535, GS_ERR_INVALID, "16-bit counts need m <= 65535");
    if (ld_out == 0) ld_out = nc;
    dim3 grid((uint32_t)((nq + HT - 1) / HT), (uint32_t)((nc + HT - 1) / HT)), block(256);
    // few tiles (e.g. the batch-mates matrix of an insert): split the rows over blockIdx.z so the whole chip works on them
    uint32_t ksplit_words = 0;
    const uint64_t roww_all = (uint64_t)m * (kind == GS_KIND_U64 ? 2 : 1);
    if (out_cnt && (uint64_t)grid.x * grid.y * 2 <= (uint64_t)c->n_cu && roww_all >= 8 * HKW) {
        uint32_t nz = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * 2 / ((uint64_t)grid.x * grid.y), roww_all / (4 * HKW));
        if (nz > 1) {
            ksplit_words = (uint32_t)(((roww_all + nz - 1) / nz + HKW - 1) / HKW * HKW);
            grid.z = (uint32_t)((roww_all + ksplit_words - 1) / ksplit_words);
            GS_HIP_CHECK(hipMemset2DAsync(out_cnt, ld_out * 4, 0, nc * 4, nq, c->stream));
        }
    }
    GS_REQUIRE(grid.y <= 65535, GS_ERR_INVALID, "too many candidate rows for one call (max %d)", 65535 * HT);
    ProfScope ps(c, FAM_HAMMING);
    const uint64_t sq = strideQ_bytes / 4, sc = strideC_bytes / 4;
    const bool vec4 = (strideQ_bytes % 16 == 0) && (strideC_bytes % 16 == 0) && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)C % 16 == 0);
#define GS_LAUNCH_QXC(K, V) hipLaunchKernelGGL((k_hamming_qxc<K, V, false>), grid, block, 0, c->stream, (const uint32_t *)Q, nq, sq, (const uint32_t *)C, nc, sc, m, out, out_cnt, out_cnt16, ld_out, ksplit_words, (const uint4 *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr)
    if (kind == GS_KIND_F32) { if (vec4) GS_LAUNCH_QXC(GS_KIND_F32, true); else GS_LAUNCH_QXC(GS_KIND_F32, false); }
    else if (kind == GS_KIND_U32) { if (vec4) GS_LAUNCH_QXC(GS_KIND_U32, true); else GS_LAUNCH_QXC(GS_KIND_U32, false); }
    else { if (vec4) GS_LAUNCH_QXC(GS_KIND_U64, true); else GS_LAUNCH_QXC(GS_KIND_U64, false); }
#undef GS_LAUNCH_QXC
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}
// every counter of the work list's tiles = m (before a K-split run takes the matches off)
__global__ __launch_bounds__(256) void k_blocks_set(const uint4 *__restrict__ items, const uint32_t *__restrict__ qlist, const uint32_t *__restrict__ clist, uint16_t *__restrict__ out16,
                                                    uint64_t ld, uint32_t m)
{
    const uint4 it = items[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < it.y * it.w; i += 256) {
        const uint32_t a = i / it.w, b = i - a * it.w;
        out16[(uint64_t)qlist[it.x + a] * ld + clist[it.z + b]] = (uint16_t)m;
    }
}
// the tiles of a work list (see k_hamming_qxc IDX): n_items x {qlist offset, rows (<= 128), clist offset, rows (<= 128)}, all in device memory
// n_thin: the first n_thin items have at most HTQ (16) query rows - they go to k_hamming_thin where it applies (4-byte elements, 16-byte aligned rows)
int hamming_blocks(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t strideQ_bytes, const void *C, uint64_t strideC_bytes, const void *items_dev,
                   uint32_t n_items, const uint32_t *qlist_dev, const uint32_t *clist_dev, uint16_t *out_cnt16, uint64_t ld_out, uint32_t n_thin)
{
    if (n_items == 0) return GS_OK;
    GS_REQUIRE(kind == GS_KIND_F32 || kind == GS_KIND_U32 || kind == GS_KIND_U64, GS_ERR_UNSUPPORTED, "DistHamming kind %d not on the device path", kind);
    GS_REQUIRE(n_thin <= n_items, GS_ERR_INVALID, "bad argument");
    const uint64_t sq = strideQ_bytes / 4, sc = strideC_bytes / 4;
    const bool vec4 = (strideQ_bytes % 16 == 0) && (strideC_bytes % 16 == 0) && ((uintptr_t)Q % 16 == 0) && ((uintptr_t)C % 16 == 0);
    if (n_thin && vec4 && kind != GS_KIND_U64 && !getenv("GS_BLOCKS_THIN_OFF")) {
        ProfScope ps(c, FAM_HAMMING);
        if (kind == GS_KIND_F32)
            hipLaunchKernelGGL(k_hamming_thin<GS_KIND_F32>, dim3(n_thin * HTS), dim3(256), 0, c->stream, (const uint32_t *)Q, sq, (const uint32_t *)C, sc, m, out_cnt16, ld_out, (const uint4 *)items_dev, qlist_dev, clist_dev);
        else
            hipLaunchKernelGGL(k_hamming_thin<GS_KIND_U32>, dim3(n_thin * HTS), dim3(256), 0, c->stream, (const uint32_t *)Q, sq, (const uint32_t *)C, sc, m, out_cnt16, ld_out, (const uint4 *)items_dev, qlist_dev, clist_dev);
        GS_HIP_CHECK(hipGetLastError());
        items_dev = (const uint4 *)items_dev + n_thin;
        n_items -= n_thin;
        if (n_items == 0) return GS_OK;
    }
    dim3 grid(n_items), block(256);
    // few tiles (thirty species in a batch = thirty tiles): the rows are split over blockIdx.z so that the whole chip works on them
    uint32_t ksplit_words = 0;
    const uint64_t roww_all = (uint64_t)m * (kind == GS_KIND_U64 ? 2 : 1);
    if ((uint64_t)n_items * 2 <= (uint64_t)c->n_cu * 2 && roww_all >= 8 * HKW) {
        const uint32_t nz = (uint32_t)std::min<uint64_t>((uint64_t)c->n_cu * 4 / n_items, roww_all / (4 * HKW));
        if (nz > 1) {
            ksplit_words = (uint32_t)(((roww_all + nz - 1) / nz + HKW - 1) / HKW * HKW);
            grid.z = (uint32_t)((roww_all + ksplit_words - 1) / ksplit_words);
            hipLaunchKernelGGL(k_blocks_set, dim3(n_items), dim3(256), 0, c->stream, (const uint4 *)items_dev, qlist_dev, clist_dev, out_cnt16, ld_out, m);
        }
    }
#define GS_LAUNCH_BLK(K, V) hipLaunchKernelGGL((k_hamming_qxc<K, V, true>), grid, block, 0, c->stream, (const uint32_t *)Q, (uint64_t)0, sq, (const uint32_t *)C, (uint64_t)0, sc, m, (float *)nullptr, (uint32_t *)nullptr, out_cnt16, ld_out, ksplit_words, (const uint4 *)items_dev, qlist_dev, clist_dev)
    if (kind == GS_KIND_F32) { if (vec4) GS_LAUNCH_BLK(GS_KIND_F32, true); else GS_LAUNCH_BLK(GS_KIND_F32, false); }
    else if (kind == GS_KIND_U32) { if (vec4) GS_LAUNCH_BLK(GS_KIND_U32, true); else GS_LAUNCH_BLK(GS_KIND_U32, false); }
    else { if (vec4) GS_LAUNCH_BLK(GS_KIND_U64, true); else GS_LAUNCH_BLK(GS_KIND_U64, false); }
#undef GS_LAUNCH_BLK
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}
static int hamming_qxc_dev_native(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc, float *out);
// u16 signatures (hll, SPEC 3.4) run through the u32 kernels: a zero-extended u16 compares equal exactly when the u16 does, so the
// mismatch counts - and everything built on them - are identical; rows are widened once when they enter device memory.
__global__ void k_widen_u16(const uint16_t *__restrict__ src, uint64_t src_pitch_elems, uint64_t nrows, uint32_t m, uint8_t *__restrict__ dst, uint64_t dst_stride_bytes)
{
    const uint64_t total = nrows * m;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / m, s = i % m;
        ((uint32_t *)(dst + r * dst_stride_bytes))[s] = src[r * src_pitch_elems + s];
    }
}
__global__ void k_narrow_u16(const uint8_t *__restrict__ src, uint64_t src_stride_bytes, uint64_t nrows, uint32_t m, uint16_t *__restrict__ dst)
{
    const uint64_t total = nrows * m;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = i / m, s = i % m;
        dst[i] = (uint16_t)((const uint32_t *)(src + r * src_stride_bytes))[s];
    }
}
int widen_u16_rows(gs_ctx *c, const void *src_dev, uint64_t nrows, uint32_t m, void *dst_dev, uint64_t dst_stride_bytes)
{
    if (nrows == 0) return GS_OK;
    hipLaunchKernelGGL(k_widen_u16, dim3((uint32_t)std::min<uint64_t>((nrows * m + 255) / 256, (uint64_t)c->n_cu * 16)), dim3(256), 0, c->stream,
                       (const uint16_t *)src_dev, (uint64_t)m, nrows, m, (uint8_t *)dst_dev, dst_stride_bytes);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}
int narrow_u16_rows(gs_ctx *c, const void *src_dev, uint64_t src_stride_bytes, uint64_t nrows, uint32_t m, void *dst_dev)
{
    if (nrows == 0) return GS_OK;
    hipLaunchKernelGGL(k_narrow_u16, dim3((uint32_t)std::min<uint64_t>((nrows * m + 255) / 256, (uint64_t)c->n_cu * 16)), dim3(256), 0, c->stream,
                       (const uint8_t *)src_dev, src_stride_bytes, nrows, m, (uint16_t *)dst_dev);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

static int hamming_qxc_dev(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc, float *out)
{
    if (kind == GS_KIND_U16) {
        PoolBuf wq(c, 30), wc(c, 31);
        int rc;
        if ((rc = wq.alloc((size_t)4 * m * nq))) return rc;
        if ((rc = wc.alloc((size_t)4 * m * nc))) return rc;
        if ((rc = widen_u16_rows(c, Q, nq, m, wq.p, (uint64_t)4 * m))) return rc;
        if ((rc = widen_u16_rows(c, C, nc, m, wc.p, (uint64_t)4 * m))) return rc;
        return hamming_qxc_strided(c, GS_KIND_U32, m, wq.p, nq, (uint64_t)4 * m, wc.p, nc, (uint64_t)4 * m, out, nullptr, nullptr, nc);
    }
    return hamming_qxc_dev_native(c, kind, m, Q, nq, C, nc, out);
}
static int hamming_qxc_dev_native(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc, float *out)
{
    const uint64_t row = kind_bytes(kind) * (uint64_t)m;
    return hamming_qxc_strided(c, kind, m, Q, nq, row, C, nc, row, out, nullptr, nullptr, nc);
}

}  // namespace gs

extern "C" {

int gs_hamming_qxc_dev(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc, float *out)
{
    GS_REQUIRE(c, GS_ERR_INVALID, "null context");
    GS_CTX_LOCK(c);
    return gs::hamming_qxc_dev(c, kind, m, Q, nq, C, nc, out);
}

int gs_hamming_qxc(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, const void *C, uint64_t nc, float *out)
{
    GS_REQUIRE(c && m > 0, GS_ERR_INVALID, "bad argument");
    if (nq == 0 || nc == 0) return GS_OK;
    GS_REQUIRE(Q && C && out, GS_ERR_INVALID, "null argument");
    return gs::on_worker(c, [&](gs_ctx *c) -> int {
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    const size_t row = gs::kind_bytes(kind) * (size_t)m;
    gs::PoolBuf dq(c, 53), dc(c, 54), dout(c, 55);
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
    });
}

int gs_hamming_pairs(gs_ctx *c, int kind, uint32_t m, const void *A, uint64_t na, const void *B, uint64_t nb, const uint64_t *ia,
                     const uint64_t *ib, uint64_t npairs, float *out)
{
    GS_REQUIRE(c && m > 0, GS_ERR_INVALID, "bad argument");
    GS_REQUIRE(kind == GS_KIND_F32 || kind == GS_KIND_U32 || kind == GS_KIND_U64 || kind == GS_KIND_U16, GS_ERR_UNSUPPORTED, "DistHamming kind %d not on the device path", kind);
    if (npairs == 0) return GS_OK;
    GS_REQUIRE(A && B && ia && ib && out, GS_ERR_INVALID, "null argument");
    for (uint64_t p = 0; p < npairs; p++) GS_REQUIRE(ia[p] < na && ib[p] < nb, GS_ERR_INVALID, "pair %llu out of range", (unsigned long long)p);
    return gs::on_worker(c, [&, kind](gs_ctx *c) mutable -> int {
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    const size_t row = gs::kind_bytes(kind) * (size_t)m;
    gs::PoolBuf da(c, 53), db(c, 54), dia(c, 55), dib(c, 56), dout(c, 57);
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
    gs::PoolBuf wa(c, 58), wb(c, 59);
    if (kind == GS_KIND_U16) {                                   // widen once, then the u32 kernel (see k_widen_u16)
        if ((rc = wa.alloc((size_t)4 * m * na))) return rc;
        if ((rc = wb.alloc((size_t)4 * m * nb))) return rc;
        if ((rc = gs::widen_u16_rows(c, da.p, na, m, wa.p, (uint64_t)4 * m))) return rc;
        if ((rc = gs::widen_u16_rows(c, db.p, nb, m, wb.p, (uint64_t)4 * m))) return rc;
        std::swap(da.p, wa.p); std::swap(da.bytes, wa.bytes); std::swap(db.p, wb.p); std::swap(db.bytes, wb.bytes);
        kind = GS_KIND_U32;
    }
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
    });
}

}  // extern "C"
