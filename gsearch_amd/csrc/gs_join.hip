// gs_join.hip — batched DistHamming as an equi-join ("match-join").
//
// DistHamming::eval (anndists; /root/reference/src/dna/dnasketch.rs:72, src/bin/bindash.rs:93-99) counts MISmatches:
//   c(q,e) = m - #{s : q[s] == e[s]}.
// Sketches of unrelated genomes agree in ~0.5 of 18000 slots, so the matches are ~10^4 times rarer than the mismatches the
// tile kernel has to touch. With a column-major copy of the database (cols[s][e]) the matches of a whole query batch are the
// equi-join, slot by slot, of the batch's column (hashed once per workgroup) with the database column (streamed once per batch):
//   per slot s:  for every node e:  probe cols[s][e] in an LDS hash table of the query values; every hit (q,e) -> matches[q][e] += 1
// HBM traffic: the database once per BATCH (21.6 GB for 300 k x 18000 f32) instead of once per 128 queries; arithmetic: ~1.5 LDS
// probes per (slot, node) instead of nq compares. Output is bit-identical to the tile kernel.
#include <algorithm>
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

// query batch -> per-slot key lists: qkey[s * nq + q] (canonical keys)
template <int KIND, typename T>
__global__ __launch_bounds__(256) void k_query_cols(const uint8_t *__restrict__ rows, uint64_t stride, uint32_t nq, uint32_t m, T *__restrict__ qkey)
{
    __shared__ T tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint64_t r0 = (uint64_t)blockIdx.x * 32, s0 = (uint64_t)blockIdx.y * 32;
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint64_t r = r0 + j, s = s0 + tx;
        tile[j][tx] = (r < nq && s < m) ? ((const T *)(rows + r * stride))[s] : (T)0;
    }
    __syncthreads();
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint64_t s = s0 + j, r = r0 + tx;
        if (s < m && r < nq) qkey[s * nq + r] = tile[tx][j];
    }
}

constexpr int JT = 1024;          // lanes per join workgroup (they share one hash table)
constexpr int JU = 4;             // database values in flight per lane
constexpr int JP_MAX_LOG2 = 13;   // largest table: 8192 entries (64 KB for 4-byte keys, 96 KB for 8-byte keys)
constexpr int JQ_MAX = 3276;      // queries per join call: load factor <= 0.4
constexpr int JB_LOG2 = 16;       // bits of the pre-filter bitmap (8 KB)

__device__ __forceinline__ uint32_t join_hash(uint32_t k) { return k * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t join_hash(uint64_t k) { return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 32); }
#define GS_SEL4(a, i) (((i) & 2) ? (((i) & 1) ? a[3] : a[2]) : (((i) & 1) ? a[1] : a[0]))

constexpr int JN = 8;             // nodes per lane: a workgroup owns JT * JN nodes for a whole block of slots
// Barrier that only orders LDS traffic (the hash table of a slot): a __syncthreads() also waits for vmcnt(0), i.e. for every count atomic
// and column prefetch the wave has in flight - three times per slot
__device__ __forceinline__ void join_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// one match of (tag = query + 1) with the lane's node `slot`: run-length accumulate in the lane's "sticky" register for that node
// (tag in bits 0-11, run length above) and only send an atomic when the run is evicted. A node related to a query matches it in
// thousands of slots, so its counter would otherwise take thousands of atomics; unrelated matches (run length 1) evict nothing
// that has proved itself (length >= 2) and go straight to memory.
#define GS_JOIN_HIT(st, tagv, e)                                                                              \
    do {                                                                                                      \
        if (((st) & 0xFFFu) == (tagv)) (st) += 0x1000u;                                                       \
        else if ((st) < 0x2000u) { if (st) { join_flush(mm32, ld, (st), (e)); natom++; } (st) = (tagv) | 0x1000u; } \
        else { join_flush(mm32, ld, (tagv) | 0x1000u, (e)); natom++; }                                        \
    } while (0)
__device__ __forceinline__ void join_flush(uint32_t *mm32, uint64_t ld, uint32_t st, uint64_t e)
{
    const uint64_t idx = (uint64_t)((st & 0xFFFu) - 1) * ld + e;
    atomicSub(&mm32[idx >> 1], (st >> 12) << ((idx & 1) * 16));      // counters start at m and count DOWN: a half-word never borrows (matches <= m)
}

// grid: (node chunks of JT*JN, slot blocks). matches[q * ld + e] (16-bit counters, incremented through their 32-bit container).
// For every slot of its block the workgroup builds, in LDS, an open-addressing hash table (linear probing, load <= 0.4) of the slot's
// query values - tag = query index + 1, 0 = empty; equal values of different queries occupy several entries of one probe run -
// plus a 64-kbit bitmap of their hashes, and streams its nodes' values of that slot through them. ~96 % of the values stop at the
// bitmap (one LDS read); the survivors (true matches plus false positives) are probed by a flattened per-lane state machine: every
// trip of the loop advances each lane by one table entry of whichever of its JU values is pending, so the wave pays for the
// longest per-lane total, not for JU times the longest probe. Keys are canonical under the element type's `==` (f32: -0 -> +0,
// NaN never inserted / never probed), so bitwise equality is the reference's equality.
// SR = slots per round: the workgroup clears, builds and probes the tables of SR consecutive slots between one set of barriers. With the 256
// queries of an insert batch a table is 1024 entries and a slot's fixed work (three barriers, clearing 3 K words, 256 inserts on a quarter of
// the lanes) weighs on its 8192 probes; four slots per round: -10.5 % of the join time of a 300 k build (tools/build_trace.sh). SR = 1 for
// request batches, whose 72 KB tables fill the LDS.
template <int KIND, typename T, int SR>
__global__ __launch_bounds__(JT, 8) void k_match_join(const T *__restrict__ qkey, uint32_t nq, uint32_t log2p, const T *__restrict__ cols, uint64_t colcap, uint64_t n,
                                                    uint32_t m, uint32_t slots_per_wg, uint32_t *__restrict__ mm32, uint64_t ld,
                                                    unsigned long long *__restrict__ stats, int chunk_major, uint64_t col0)
{
    static_assert(JU == 4 && JN % JU == 0, "GS_SEL4 / pending mask are written for JU = 4");
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    const uint32_t P = 1u << log2p, mask = P - 1, sh = 32 - log2p;
    constexpr uint32_t BMW = (1u << JB_LOG2) / 32;                // bitmap words per slot
    uint32_t *tag = (uint32_t *)s_raw;                            // [SR][P]
    uint32_t *bm = (uint32_t *)(s_raw + 4 * (size_t)P * SR);      // [SR][BMW]
    T *key = (T *)(s_raw + (4 * (size_t)P + (size_t)BMW * 4) * SR);   // [SR][P]
    // chunk_major: consecutive workgroups sweep the slot blocks of ONE node chunk, so the counters being updated at any time are
    // those of a few chunks (a window of the count matrix that fits the 256 MB Infinity Cache) instead of all of them
    const uint32_t bchunk = chunk_major ? blockIdx.y : blockIdx.x, bslot = chunk_major ? blockIdx.x : blockIdx.y;
    const uint64_t e0 = (uint64_t)bchunk * (JT * JN) + threadIdx.x;      // the lane's nodes: e0 + i * JT, i < JN
    const uint32_t s0 = bslot * slots_per_wg, s1 = s0 + slots_per_wg < m ? s0 + slots_per_wg : m;
    uint32_t sticky[JN];
    uint32_t natom = 0;                                           // memory-side atomics this lane sends (work counter for the bench's roofline)
#pragma unroll
    for (int i = 0; i < JN; i++) sticky[i] = 0;
    T vn[JU];
#pragma unroll
    for (int u = 0; u < JU; u++) { const uint64_t e = e0 + (uint64_t)u * JT; vn[u] = (e < n && s0 < s1) ? cols[(uint64_t)s0 * colcap + e] : (T)0; }
    for (uint32_t sr = s0; sr < s1; sr += SR) {
        const uint32_t nr = s1 - sr < (uint32_t)SR ? s1 - sr : (uint32_t)SR;       // slots of this round
        join_lds_barrier();                                       // the previous round's probes are done
        for (uint32_t i = threadIdx.x; i < P * SR; i += JT) tag[i] = 0;
        for (uint32_t i = threadIdx.x; i < BMW * SR; i += JT) bm[i] = 0;
        join_lds_barrier();
        for (uint32_t i = threadIdx.x; i < nq * nr; i += JT) {
            const uint32_t r = SR == 1 ? 0 : i / nq, q = SR == 1 ? i : i - r * nq;
            T k = qkey[(uint64_t)(sr + r) * nq + q];
            if (never_equal<KIND, T>(k)) continue;
            k = canon<KIND, T>(k);
            const uint32_t hq = join_hash(k);
            atomicOr(&bm[r * BMW + (hq >> (32 - JB_LOG2 + 5))], 1u << ((hq >> (32 - JB_LOG2)) & 31));
            uint32_t h = hq >> sh;
            while (atomicCAS(&tag[r * P + h], 0u, q + 1) != 0u) h = (h + 1) & mask;
            key[r * P + h] = k;
        }
        join_lds_barrier();
#pragma unroll 1
        for (uint32_t r = 0; r < nr; r++) {
            const uint32_t s = sr + r, tb = r * P, bb = r * BMW;
            const T *col = cols + (uint64_t)s * colcap;
#pragma unroll
            for (int it = 0; it < JN / JU; it++) {
                // (the four bitmap words are read unconditionally and together: behind `&&` each read sat in its own branch with its own wait)
                T v[JU]; uint32_t hs[JU]; uint32_t pend = 0; uint32_t bw[JU];
#pragma unroll
                for (int u = 0; u < JU; u++) {
                    v[u] = canon<KIND, T>(vn[u]);
                    hs[u] = join_hash(v[u]);
                    bw[u] = bm[bb + (hs[u] >> (32 - JB_LOG2 + 5))];
                }
#pragma unroll
                for (int u = 0; u < JU; u++) {
                    const uint64_t e = e0 + (uint64_t)(it * JU + u) * JT;
                    const uint32_t bit = hs[u] >> (32 - JB_LOG2);
                    const uint32_t pass = (uint32_t)(e < n) & (uint32_t)!never_equal<KIND, T>(vn[u]) & (bw[u] >> (bit & 31)) & 1u;
                    pend |= pass << u;
                }
                // next values: the following nodes of this slot, or the first nodes of the next slot
#pragma unroll
                for (int u = 0; u < JU; u++) {
                    const bool wrap = it + 1 == JN / JU;
                    const uint64_t e = e0 + (uint64_t)((wrap ? 0 : (it + 1) * JU) + u) * JT;
                    const T *src = wrap ? col + colcap : col;
                    vn[u] = (e < n && (!wrap || s + 1 < s1)) ? src[e] : (T)0;
                }
                uint32_t hh = 0, uu = 0; T vv = 0; bool have = false;
                for (;;) {
                    if (!have && pend) { uu = (uint32_t)__ffs((int)pend) - 1; pend &= pend - 1; vv = GS_SEL4(v, uu); hh = GS_SEL4(hs, uu) >> sh; have = true; }
                    if (!have) break;
                    const uint32_t t = tag[tb + hh];
                    const T k = key[tb + hh];
                    if (t == 0u) { have = false; continue; }
                    if (k == vv) {
                        const uint64_t e = col0 + e0 + (uint64_t)(it * JU + uu) * JT;       // column of the count matrix (col0: the node range starts there)
                        uint32_t st = GS_SEL4((sticky + it * JU), uu);
                        GS_JOIN_HIT(st, t, e);
#pragma unroll
                        for (int u = 0; u < JU; u++) if (uu == (uint32_t)u) sticky[it * JU + u] = st;
                    }
                    hh = (hh + 1) & mask;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < JN; i++) if (sticky[i]) { join_flush(mm32, ld, sticky[i], col0 + e0 + (uint64_t)i * JT); natom++; }
    if (stats) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) natom += __shfl_down(natom, o);
        if ((threadIdx.x & 63) == 0 && natom) atomicAdd(stats, (unsigned long long)natom);
    }
}
// Match-density probe: what would the join cost on THIS batch? Its work is proportional to the number of matches it has to
// record, and a redundant query set against a redundant database (hundreds of near-identical genomes on both sides) has orders of
// magnitude more of them than unrelated data - there the compare tile kernel, whose cost is fixed, is the better producer.
// Each workgroup takes one node chunk and one sampled PAIR of consecutive slots (s, s+1): it counts the matches of slot s+1 and
// how many of them repeat the (query, node) pair that node matched first in slot s - the matches the run-length accumulator
// of k_match_join absorbs. out[0] += matches, out[1] += repeats.
template <int KIND, typename T>
__global__ __launch_bounds__(JT) void k_match_sample(const T *__restrict__ qkey, uint32_t nq, uint32_t log2p, const T *__restrict__ cols, uint64_t colcap, uint64_t n,
                                                      uint32_t m, uint32_t nsamp, unsigned long long *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    const uint32_t P = 1u << log2p, mask = P - 1, sh = 32 - log2p;
    uint32_t *tag = (uint32_t *)s_raw;
    T *key = (T *)(s_raw + 4 * (size_t)P);
    const uint64_t e0 = (uint64_t)blockIdx.x * (JT * JN) + threadIdx.x;
    const uint32_t sbase = (uint32_t)(((uint64_t)blockIdx.y * (m - 1)) / nsamp);      // sbase + 1 <= m - 1
    uint32_t first[JN];
#pragma unroll
    for (int i = 0; i < JN; i++) first[i] = 0;
    uint32_t hits = 0, reps = 0;
    for (uint32_t pass = 0; pass < 2; pass++) {
        const uint32_t s = sbase + pass;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < P; i += JT) tag[i] = 0;
        __syncthreads();
        for (uint32_t q = threadIdx.x; q < nq; q += JT) {
            T k = qkey[(uint64_t)s * nq + q];
            if (never_equal<KIND, T>(k)) continue;
            k = canon<KIND, T>(k);
            uint32_t h = join_hash(k) >> sh;
            while (atomicCAS(&tag[h], 0u, q + 1) != 0u) h = (h + 1) & mask;
            key[h] = k;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < JN; i++) {
            const uint64_t e = e0 + (uint64_t)i * JT;
            if (e >= n) continue;
            T v = cols[(uint64_t)s * colcap + e];
            if (never_equal<KIND, T>(v)) continue;
            v = canon<KIND, T>(v);
            uint32_t h = join_hash(v) >> sh, t;
            while ((t = tag[h]) != 0u) {
                if (key[h] == v) {
                    if (pass == 0) { if (!first[i]) first[i] = t; }
                    else { hits++; reps += (t == first[i]); }
                }
                h = (h + 1) & mask;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { hits += __shfl_down(hits, o); reps += __shfl_down(reps, o); }
    if ((threadIdx.x & 63) == 0 && hits) { atomicAdd(&out[0], (unsigned long long)hits); atomicAdd(&out[1], (unsigned long long)reps); }
}

template <int KIND, typename T>
static int join_impl(gs_ctx *c, uint32_t m, const uint8_t *qrows, uint64_t qstride, uint32_t nq, const void *cols, uint64_t colcap, uint64_t n, uint16_t *out16,
                     uint64_t ld, DevBuf *scratch /* [5] reusable */, int *declined, unsigned long long *stats, bool init, uint64_t col0)
{
    int rc;
    if (declined) *declined = 0;
    const size_t items = (size_t)m * nq;
    DevBuf &k0 = scratch[0];
    if ((rc = k0.ensure(sizeof(T) * items))) return rc;
    // every 16-bit counter starts at m and every match takes one off: the matrix leaves the join as mismatch counts, without the pass that
    // used to turn matches into mismatches (6 GB read + written per 10 000-query request)
    if (init) GS_HIP_CHECK(hipMemsetD16Async((hipDeviceptr_t)out16, (unsigned short)m, (size_t)nq * ld, c->stream));
    dim3 tg((nq + 31) / 32, (m + 31) / 32);
    hipLaunchKernelGGL((k_query_cols<KIND, T>), tg, dim3(256), 0, c->stream, qrows, qstride, nq, m, k0.as<T>());
    GS_HIP_CHECK(hipGetLastError());
    uint32_t log2p = 6;
    while (log2p < (uint32_t)JP_MAX_LOG2 && (double)(1u << log2p) * 0.4 < (double)nq) log2p++;
    if (declined && m >= 64 && n >= 4096) {
        // sampled match density -> estimated join time (column stream + one memory-side atomic per match the accumulator does not
        // absorb, 1.6e10/s: profiles/r01_match_join_pmc.txt) against the fixed cost of the compare tile kernel
        const uint32_t nsamp = 24;
        DevBuf &cnt = scratch[1];
        if ((rc = cnt.ensure(16))) return rc;
        GS_HIP_CHECK(hipMemsetAsync(cnt.p, 0, 16, c->stream));
        const uint32_t chunks_s = (uint32_t)((n + (uint64_t)JT * JN - 1) / ((uint64_t)JT * JN));
        const size_t lds_s = (sizeof(T) + 4) * ((size_t)1 << log2p);
        auto ks = k_match_sample<KIND, T>;
        if (lds_s > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
        hipLaunchKernelGGL(ks, dim3(chunks_s, nsamp), dim3(JT), lds_s, c->stream, k0.as<T>(), nq, log2p, (const T *)cols, colcap, n, m, nsamp, cnt.as<unsigned long long>());
        GS_HIP_CHECK(hipGetLastError());
        unsigned long long hr[2] = {0, 0};
        GS_HIP_CHECK(hipMemcpyAsync(hr, cnt.p, 16, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        const double scale = (double)m / (double)nsamp;
        const double atomics = (double)(hr[0] - hr[1]) * scale + (double)hr[1] * scale / 64.0;       // absorbed runs still flush now and then
        const double t_join = (double)n * m * sizeof(T) / 3.5e12 + atomics / 1.6e10 + (double)hr[0] * scale / 2.0e11;
        const double t_tile = (double)((nq + 127) / 128 * 128) * (double)n * (double)m / (KIND == GS_KIND_U64 ? 1.4e13 : 1.6e13);   // 128-query tiles
        const char *force = getenv("GS_JOIN_DECLINE");
        const bool decline = force ? atoi(force) != 0 : t_join > t_tile;
        if (getenv("GS_JOIN_VERBOSE"))
            fprintf(stderr, "[GS_JOIN] nq=%u n=%llu sampled matches %llu repeats %llu -> est. join %.2f ms, tile %.2f ms: %s\n", nq, (unsigned long long)n, hr[0], hr[1],
                    t_join * 1e3, t_tile * 1e3, decline ? "tile" : "join");
        if (decline) { *declined = 1; return GS_OK; }
    }
    // one workgroup = JT * JN nodes x a block of slots; blocks sized so that the grid is about eight rounds of 2 workgroups per CU
    // (one round leaves the slowest workgroup's tail exposed: 145 -> 125 ms per 10 k-query request), at least 32 slots each
    const uint32_t chunks = (uint32_t)((n + (uint64_t)JT * JN - 1) / ((uint64_t)JT * JN));
    uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((16 * c->n_cu) / chunks, std::max<uint32_t>(m / 32, (2 * c->n_cu) / chunks)));
    if (getenv("GS_JOIN_BLOCKS")) blocks = (uint32_t)atoi(getenv("GS_JOIN_BLOCKS"));
    blocks = std::min<uint32_t>(std::max<uint32_t>(blocks, 1), m);
    const uint32_t slots_per_wg = (m + blocks - 1) / blocks;
    const int chunk_major = getenv("GS_JOIN_CHUNK_MAJOR") ? atoi(getenv("GS_JOIN_CHUNK_MAJOR")) : 0;
    const uint32_t nblk = (m + slots_per_wg - 1) / slots_per_wg;
    dim3 jg(chunk_major ? nblk : chunks, chunk_major ? chunks : nblk);
    const size_t lds1 = (sizeof(T) + 4) * ((size_t)1 << log2p) + ((size_t)1 << JB_LOG2) / 8;
    // small tables (an insert batch): several slots per barrier round while two workgroups still fit a CU
    int sr = 1;
    if (4 * lds1 <= 80 * 1024 && slots_per_wg >= 8) sr = 4; else if (2 * lds1 <= 80 * 1024 && slots_per_wg >= 4) sr = 2;
    if (getenv("GS_JOIN_SLOTS_PER_ROUND")) { const int e = atoi(getenv("GS_JOIN_SLOTS_PER_ROUND")); if (e == 1 || (e == 2 && 2 * lds1 <= 80 * 1024) || (e == 4 && 4 * lds1 <= 80 * 1024)) sr = e; }
    const size_t lds = lds1 * sr;
    {
        ProfScope ps(c, FAM_HAMMING);
        auto kern = sr == 4 ? k_match_join<KIND, T, 4> : sr == 2 ? k_match_join<KIND, T, 2> : k_match_join<KIND, T, 1>;
        if (lds > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, jg, dim3(JT), lds, c->stream, k0.as<T>(), nq, log2p, (const T *)cols, colcap, n, m, slots_per_wg, (uint32_t *)out16, ld, stats, chunk_major, col0);
        GS_HIP_CHECK(hipGetLastError());
    }
    return GS_OK;
}

uint64_t match_join_max_queries() { const char *e = getenv("GS_JOIN_MAXQ"); return e ? (uint64_t)std::max(1, std::min(4094, atoi(e))) : JQ_MAX; }

// `declined` (optional): set to 1 - and out16 is left at its initial value m - when the sampled match density says the compare tile kernel is the
// cheaper producer for this batch (the caller then runs it)
int match_join_counts(gs_ctx *c, int kind, uint32_t m, const void *qrows, uint64_t qstride, uint64_t nq, const void *cols, uint64_t colcap, uint64_t n,
                      uint16_t *out16, uint64_t ld, DevBuf *scratch, int *declined, unsigned long long *stats, bool init, uint64_t col0)
{
    GS_REQUIRE(nq >= 1 && nq <= 4094 && m <= 65535 && (ld % 2) == 0 && ((uintptr_t)out16 % 4) == 0, GS_ERR_INVALID, "match_join_counts: bad shape");
    GS_REQUIRE((uint64_t)m * nq < ((uint64_t)1 << 31), GS_ERR_INVALID, "match_join_counts: batch too large");
    if (kind == GS_KIND_U64) return join_impl<GS_KIND_U64, uint64_t>(c, m, (const uint8_t *)qrows, qstride, (uint32_t)nq, cols, colcap, n, out16, ld, scratch, declined, stats, init, col0);
    if (kind == GS_KIND_F32) return join_impl<GS_KIND_F32, uint32_t>(c, m, (const uint8_t *)qrows, qstride, (uint32_t)nq, cols, colcap, n, out16, ld, scratch, declined, stats, init, col0);
    return join_impl<GS_KIND_U32, uint32_t>(c, m, (const uint8_t *)qrows, qstride, (uint32_t)nq, cols, colcap, n, out16, ld, scratch, declined, stats, init, col0);
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
