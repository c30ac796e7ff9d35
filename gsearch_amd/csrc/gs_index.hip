// gs_index.hip — Hnsw<Sig, DistHamming> on gfx950 (SPEC.md 5).
//
// Replaces hnsw_rs::Hnsw::{new, parallel_insert, parallel_search} as gsearch drives them
// (/root/reference/src/dna/dnasketch.rs:139-141,159-160,435; src/dna/dnarequest.rs:353; src/aa/aasketch.rs:407;
// src/aa/aarequest.rs:344) with the DistHamming::eval inner loop (dnasketch.rs:72) fused in.
//
// Data layout in HBM: signature rows padded to a 256-byte stride (zero padded, so padding never mismatches);
// layer-0 adjacency dense [n][2M] ids (+ mismatch counts), upper layers dense for the few nodes of level >= 1.
//
// Search kernel: ONE workgroup (1024 lanes) per query, persistent over a query queue. The candidate heap C and
// the bounded result set R of Malkov's search_layer live in LDS as *sorted arrays* of 64-bit keys
// (mismatch count << 32 | id) — the total order of SPEC 5. One iteration = pop the closest candidate, gather its
// unvisited neighbours (visited bitmap in L2-resident global memory), evaluate all their distances at once
// (the HBM-bound part: every 16-byte chunk of a candidate row is fetched exactly once, compared with
// v_cmp + ballot/popcount against the query chunk held in registers), then apply the sequential accept rule of
// search_layer in closed form and merge the accepted keys into R and C with one in-LDS parallel merge.
#include <algorithm>
#include <functional>
#include <chrono>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <sys/stat.h>
#include "gs_internal.hpp"
#include "gs_spec.hpp"

namespace gs {

constexpr int ST = 1024;          // lanes per search workgroup
constexpr int SMAXI = 16;         // staged keys per lane in a merge -> arrays up to 16384 keys
constexpr int RIF = 8;            // candidate rows in flight per lane
#define KEY(c, id) (((uint64_t)(c) << 32) | (uint64_t)(id))
#define KCNT(k) ((uint32_t)((k) >> 32))
#define KID(k) ((uint32_t)(k))
#define INF_CNT 0xFFFFFFFFu

struct IndexDev {
    const uint8_t *data; uint64_t stride; uint32_t nchunks; uint32_t m;
    uint32_t M, max_layer;
    const uint8_t *levels; const uint32_t *deg0; const uint32_t *nbr0;
    const int32_t *upidx; const uint32_t *degU; const uint32_t *nbrU;
    const uint64_t *rowptr;      // pair cache: rowptr[a] -> uint16 counts of node a against every node b < a (0 = not cached)
    // sparse pair rows (round 5; what replaces the n^2-byte cache beyond ~400 k nodes): node a keeps the (b, c(a,b)) of ITS CLOSEST older nodes -
    // every b < a with c(a,b) <= cut(a), at most sp_L of them, ascending by b. sp_meta[a] = valid << 63 | len << 16 | cut(a); a pair that is not in
    // the list of its younger node has a count ABOVE that node's cut - all the selection heuristic needs to know while its threshold is <= the cut.
    // sp_bm[a] (optional, 0 = none): bitmap over b < a of "m - c(a,b) >= J0(a) - 1", the level just below the list's cut - kept for the nodes whose list
    // is shorter than ~ef_construction, where a selection walk reaches candidates of that level (a pair that is not listed has fewer than J0 matches,
    // so the bit says "exactly J0 - 1": all a threshold of cut + 1 asks)
    // The lists have the length they need (the cut level of a node of a 1.5 M-genome index holds ~0.0036 a + its family, not sp_L): they live in a
    // grow-in-place arena (VmArena) with the level bitmaps, sp_off[a] = offset of a's ids from sp_base in 32-byte units, its 16-bit counts follow the ids.
    const uint8_t *sp_base; const uint32_t *sp_off; const uint64_t *sp_meta; const uint64_t *sp_bm; uint32_t sp_L;
    uint64_t n; int64_t entry; int top;
};

struct SearchLds {              // carve-up of the dynamic LDS region
    uint64_t *R, *C, *A, *As;   // R[ef], C[capC], A[maxdeg] sorted accepted keys, As[maxdeg] staging
    uint32_t *Eid, *Ecnt;       // [maxdeg]
    uint32_t *wsum;             // [ST/64]
    uint64_t *scal;             // small scalars
};
__host__ __device__ inline size_t search_lds_bytes(uint32_t ef, uint32_t maxdeg)
{
    size_t capC = 2 * (size_t)ef + maxdeg + 64;
    return 8 * (size_t)ef + 8 * capC + 16 * (size_t)maxdeg + 4 * (size_t)maxdeg * 2 + 4 * (ST / 64) + 8 * 8 + 64;
}

// number of keys in sorted a[0..n) that are < k
__device__ __forceinline__ uint32_t lower_bound_keys(const uint64_t *a, uint32_t n, uint64_t k)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (a[mid] < k) lo = mid + 1; else hi = mid; }
    return lo;
}

// #{j < na : A[j] < k}. Most merges insert a handful of keys: those are held in registers (ar[], padded with ~0) and the
// rank is a few compares instead of a dependent chain of LDS reads.
struct SmallA { uint64_t ar[8]; const uint64_t *A; uint32_t na; };
__device__ __forceinline__ SmallA load_small_a(const uint64_t *A, uint32_t na)
{
    SmallA s; s.A = A; s.na = na;
#pragma unroll
    for (int j = 0; j < 8; j++) s.ar[j] = (uint32_t)j < na ? A[j] : ~(uint64_t)0;
    return s;
}
__device__ __forceinline__ uint32_t lb_a(const SmallA &s, uint64_t k)
{
    if (s.na <= 8) {
        uint32_t r = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) r += (s.ar[j] < k);
        return r;
    }
    return lower_bound_keys(s.A, s.na, k);
}
// Merge sorted A[0..na) (na <= blockDim) into the sorted live range keys[head..n); the result starts at
// keys[0] and is truncated to `keep`. All lanes call; returns the new length. Keys are unique.
__device__ __forceinline__ uint32_t block_merge(uint64_t *keys, uint32_t head, uint32_t n, const uint64_t *A, uint32_t na, uint32_t keep, const SmallA &sa)
{
    uint64_t kv[SMAXI]; uint32_t pos[SMAXI];
    const uint32_t live = n - head;
#pragma unroll
    for (int it = 0; it < SMAXI; it++) {
        uint32_t idx = threadIdx.x + it * ST;
        pos[it] = 0xFFFFFFFFu; kv[it] = 0;
        if (idx < live) { uint64_t k = keys[head + idx]; kv[it] = k; pos[it] = idx + lb_a(sa, k); }
    }
    uint64_t ak = 0; uint32_t apos = 0xFFFFFFFFu;
    if (threadIdx.x < na) { ak = A[threadIdx.x]; apos = threadIdx.x + lower_bound_keys(keys + head, live, ak); }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < SMAXI; it++) if (pos[it] < keep) keys[pos[it]] = kv[it];
    if (apos < keep) keys[apos] = ak;
    __syncthreads();
    uint32_t tot = live + na;
    return tot < keep ? tot : keep;
}

template <int KIND>
__device__ __forceinline__ uint32_t chunk_mismatch_wave(const uint4 &a, const uint4 &b, bool valid)
{
    // wave-uniform number of mismatching elements of this 16-byte chunk summed over the wavefront
    if (KIND == GS_KIND_U64) {
        bool n0 = valid && ((a.x != b.x) || (a.y != b.y));
        bool n1 = valid && ((a.z != b.z) || (a.w != b.w));
        return (uint32_t)__popcll(__ballot(n0)) + (uint32_t)__popcll(__ballot(n1));
    } else if (KIND == GS_KIND_F32) {
        bool n0 = valid && (__uint_as_float(a.x) != __uint_as_float(b.x));
        bool n1 = valid && (__uint_as_float(a.y) != __uint_as_float(b.y));
        bool n2 = valid && (__uint_as_float(a.z) != __uint_as_float(b.z));
        bool n3 = valid && (__uint_as_float(a.w) != __uint_as_float(b.w));
        return (uint32_t)__popcll(__ballot(n0)) + (uint32_t)__popcll(__ballot(n1)) + (uint32_t)__popcll(__ballot(n2)) + (uint32_t)__popcll(__ballot(n3));
    } else {
        bool n0 = valid && (a.x != b.x), n1 = valid && (a.y != b.y), n2 = valid && (a.z != b.z), n3 = valid && (a.w != b.w);
        return (uint32_t)__popcll(__ballot(n0)) + (uint32_t)__popcll(__ballot(n1)) + (uint32_t)__popcll(__ballot(n2)) + (uint32_t)__popcll(__ballot(n3));
    }
}

// Ecnt[e] = mismatch count between query row q and data row Eid[e], e < ne. All lanes call.
// `matrow` != nullptr: the counts of this query against every node were precomputed by the dense tile kernel
// (DESIGN.md 3.5 "dense mode") and are simply looked up.
template <int KIND>
__device__ __forceinline__ void block_distances(const IndexDev &ix, const uint4 *__restrict__ q, const uint32_t *Eid, uint32_t ne, uint32_t *Ecnt,
                                                const uint16_t *__restrict__ matrow = nullptr)
{
    if (matrow) {
        for (uint32_t e = threadIdx.x; e < ne; e += ST) Ecnt[e] = matrow[Eid[e]];
        __syncthreads();
        return;
    }
    for (uint32_t e = threadIdx.x; e < ne; e += ST) Ecnt[e] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t j0 = 0; j0 < ix.nchunks; j0 += ST) {
        const uint32_t ch = j0 + threadIdx.x;
        const bool valid = ch < ix.nchunks;
        if (j0 + (threadIdx.x & ~63u) >= ix.nchunks) break;          // whole wave past the row end
        uint4 qv = make_uint4(0, 0, 0, 0);
        if (valid) qv = q[ch];
        for (uint32_t e0 = 0; e0 < ne; e0 += RIF) {
            uint4 rv[RIF];
#pragma unroll
            for (int r = 0; r < RIF; r++) {
                rv[r] = make_uint4(0, 0, 0, 0);
                if (e0 + r < ne && valid) rv[r] = *(const uint4 *)(ix.data + (uint64_t)Eid[e0 + r] * ix.stride + (uint64_t)ch * 16);
            }
#pragma unroll
            for (int r = 0; r < RIF; r++) {
                if (e0 + r < ne) {
                    uint32_t s = chunk_mismatch_wave<KIND>(qv, rv[r], valid);
                    if (lane == 0 && s) atomicAdd(&Ecnt[e0 + r], s);
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void node_neighbours(const IndexDev &ix, uint32_t node, int L, const uint32_t *&nbr, uint32_t &deg)
{
    if (L == 0) { nbr = ix.nbr0 + (uint64_t)node * 2 * ix.M; deg = ix.deg0[node]; }
    else {
        int32_t u = ix.upidx[node];
        nbr = ix.nbrU + ((uint64_t)u * ix.max_layer + (uint32_t)(L - 1)) * ix.M;
        deg = ix.degU[(uint64_t)u * ix.max_layer + (uint32_t)(L - 1)];
    }
}

// Malkov alg. 2 / hnsw_rs::search_layer (SPEC 5) for one query by one workgroup. On return S.R[0..nR) holds the
// result sorted ascending by (count,id). `vis` = this workgroup's visited bitmap (cleared by the caller).
template <int KIND>
__device__ __forceinline__ uint32_t search_layer_block(const IndexDev &ix, const uint4 *__restrict__ q, const SearchLds &S, uint32_t *vis,
                                                       uint32_t ep, uint32_t ep_cnt, uint32_t ef, int L, uint64_t &evals,
                                                       const uint16_t *__restrict__ matrow = nullptr)
{
    const uint32_t maxdeg = 2 * ix.M;
    const uint32_t capC = 2 * ef + maxdeg + 64;
    uint32_t nR = 1, nC = 1, headC = 0, tieT = 0;
    if (threadIdx.x == 0) { S.R[0] = KEY(ep_cnt, ep); S.C[0] = KEY(ep_cnt, ep); __hip_atomic_fetch_or(&vis[ep >> 5], 1u << (ep & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __syncthreads();
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // software prefetch of the NEXT candidate's adjacency (valid unless this expansion inserts a closer candidate)
    uint64_t pre_c = ~(uint64_t)0; uint32_t pre_id = 0, pre_deg = 0;
    for (;;) {
        if (headC >= nC) break;
        const uint64_t c = S.C[headC];
        const uint32_t dmax = (nR == ef) ? KCNT(S.R[ef - 1]) : INF_CNT;
        if (KCNT(c) > dmax) break;
        headC++;
        // ---- gather unvisited neighbours of c, in stored order
        uint32_t id, deg;
        if (pre_c == c) { id = pre_id; deg = pre_deg; }
        else {
            const uint32_t *nbr;
            node_neighbours(ix, KID(c), L, nbr, deg);
            id = threadIdx.x < maxdeg ? nbr[threadIdx.x] : 0;      // lists are allocated to full width: in-bounds past deg
        }
        bool unv = false; uint32_t cntv = 0;
        if (threadIdx.x < deg) {
            const uint32_t bit = 1u << (id & 31);
            // the bitmap is private to this workgroup (LDS, or global where workgroup scope keeps the RMW in the XCD's L2 - device
            // scope went to memory: rocprof WRITE_SIZE showed ~146 B of HBM writes per evaluation)
            const uint32_t old = __hip_atomic_fetch_or(&vis[id >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            unv = !(old & bit);
            if (matrow && unv) cntv = matrow[id];                    // dense mode: every 2-byte lookup costs an HBM sector, only for the unvisited
        }
        if (headC < nC) {                                            // issue the next candidate's adjacency loads now
            pre_c = S.C[headC];
            const uint32_t *nbr2;
            node_neighbours(ix, KID(pre_c), L, nbr2, pre_deg);
            pre_id = threadIdx.x < maxdeg ? nbr2[threadIdx.x] : 0;
        } else pre_c = ~(uint64_t)0;
        const uint64_t bal = __ballot(unv);
        if (lane == 0) S.wsum[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = 0, ne = 0;
#pragma unroll
        for (int w = 0; w < ST / 64; w++) { uint32_t x = S.wsum[w]; if (w < (int)wv) off += x; ne += x; }
        if (unv) { const uint32_t pos = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1)); S.Eid[pos] = id; if (matrow) S.Ecnt[pos] = cntv; }
        __syncthreads();
        if (ne == 0) continue;
        evals += ne;
        // ---- all distances of this expansion (gather mode: the HBM-bound part)
        if (!matrow) block_distances<KIND>(ix, q, S.Eid, ne, S.Ecnt);
        // ---- closed form of the sequential accept rule (DESIGN.md "accept rule"):
        //      e_i accepted  <=>  #{x in R : c(x) <= c_i} + #{j < i : c_j <= c_i}  <  ef
        uint64_t mykey = ~(uint64_t)0; bool acc = false;
        uint32_t na;
        {
            // closed-form accept rule with the two fast paths of 3.6: a full R rejects c_i >= c(worst) outright, and when
            // the B candidates below c(worst) number <= tieT (keys of R tied at the worst count) all of them are accepted.
            const uint32_t ci = threadIdx.x < ne ? S.Ecnt[threadIdx.x] : INF_CNT;
            const bool below = nR == ef && threadIdx.x < ne && ci < dmax;
            const uint32_t B = nR == ef ? (uint32_t)__syncthreads_count(below) : 0xFFFFFFFFu;
            if (B == 0) na = 0;
            else if (B <= tieT) { acc = below; na = B; if (acc) mykey = KEY(ci, S.Eid[threadIdx.x]); }
            else {
                if (threadIdx.x < ne && !(nR == ef && ci >= dmax)) {
                    uint32_t le = lower_bound_keys(S.R, nR, KEY(ci, 0xFFFFFFFFu));       // keys < (ci,max) ; ids never reach 2^32-1
                    if (le < ef) {
#pragma unroll 8
                        for (uint32_t j = 0; j < threadIdx.x; j++) le += (S.Ecnt[j] <= ci);
                    }
                    acc = le < ef;
                    if (acc) mykey = KEY(ci, S.Eid[threadIdx.x]);
                }
                na = (uint32_t)__syncthreads_count(acc);
            }
        }
        if (na == 0) continue;
        {   // accepted keys: compact (ballot prefix) into As, rank-sort the na (usually a handful) keys into A
            const uint64_t ab = __ballot(acc);
            if (lane == 0) S.wsum[wv] = (uint32_t)__popcll(ab);
            __syncthreads();
            uint32_t aoff = 0;
#pragma unroll
            for (int w = 0; w < ST / 64; w++) if (w < (int)wv) aoff += S.wsum[w];
            if (acc) S.As[aoff + (uint32_t)__popcll(ab & ((1ull << lane) - 1))] = mykey;
            __syncthreads();
            if (threadIdx.x < na) {
                const uint64_t k = S.As[threadIdx.x];
                uint32_t rank = 0;
#pragma unroll 8
                for (uint32_t j = 0; j < na; j++) rank += (S.As[j] < k);
                S.A[rank] = k;
            }
            __syncthreads();
        }
        // ---- R <- ef smallest of R u A ; C <- live C u A ; drop dead tail of C
        const SmallA sa = load_small_a(S.A, na);
        nR = block_merge(S.R, 0, nR, S.A, na, ef, sa);
        nC = block_merge(S.C, headC, nC, S.A, na, capC, sa);
        tieT = nR == ef ? ef - lower_bound_keys(S.R, nR, KEY(KCNT(S.R[ef - 1]), 0)) : 0;
        headC = 0;
        if (nR == ef) {
            uint32_t alive = lower_bound_keys(S.C, nC, KEY(KCNT(S.R[ef - 1]), 0xFFFFFFFFu));
            if (alive < nC) nC = alive;
        }
    }
    __syncthreads();
    return nR;
}

// greedy descent on one upper layer (hnsw_rs::search outer loop, SPEC 5)
template <int KIND>
__device__ __forceinline__ void greedy_layer_block(const IndexDev &ix, const uint4 *__restrict__ q, const SearchLds &S, uint32_t &ep,
                                                   uint32_t &ep_cnt, int L, uint64_t &evals, const uint16_t *__restrict__ matrow = nullptr)
{
    for (;;) {
        const uint32_t *nbr; uint32_t deg;
        node_neighbours(ix, ep, L, nbr, deg);
        if (deg == 0) return;
        if (threadIdx.x < deg) S.Eid[threadIdx.x] = nbr[threadIdx.x];
        if (threadIdx.x == 0) S.scal[0] = ~(uint64_t)0;
        __syncthreads();
        evals += deg;
        block_distances<KIND>(ix, q, S.Eid, deg, S.Ecnt, matrow);
        if (threadIdx.x < deg && S.Ecnt[threadIdx.x] < ep_cnt) atomicMin((unsigned long long *)&S.scal[0], (unsigned long long)KEY(S.Ecnt[threadIdx.x], threadIdx.x));
        __syncthreads();
        const uint64_t best = S.scal[0];
        const uint32_t bid = best == ~(uint64_t)0 ? 0 : S.Eid[KID(best)];
        __syncthreads();
        if (best == ~(uint64_t)0) return;
        ep = bid; ep_cnt = KCNT(best);
    }
}

// ======================================================================================================
// Dense-mode traversal: all counts of the query against every node are in `matrow`, so an iteration is pure
// latency (adjacency load, visited test, lookup) and the kernel runs up to THREE 512-lane workgroups per CU to overlap it.
//
// The result set R is NOT kept as a sorted key array here. The accept rule of search_layer_block only needs, of R, the
// multiset of its counts (rank queries, worst count, number of keys tied at the worst count), and the answer only needs the
// knbn smallest keys ever accepted (R is always "the ef smallest accepted keys", so its head is exactly that). So R becomes
//   Hf : u16 histogram of the counts in R (one bin per count), H2 : sums per 16 bins, H1 : sums per 64 bins,
//        registers dmax / tieT / nR;   T : sorted array of the min(knbn, nR) smallest keys (LDS)
// which turns the O(ef) merge per accepting expansion into O(#accepted) histogram updates; dropping the largest keys of a
// full R = decrementing the top bins. Candidates: sorted bulk array G (global ping-pong, read through a 64-key LDS window) plus
// the small sorted LDS array N that receives the accepted keys.
//
// Two placements (template VLDS), chosen on the host by what fits in LDS:
//   VLDS = true : visited bitmap in LDS (n/8 bytes), Hf in global (the hot path only sends fire-and-forget atomics to it; H1/H2
//                 stay in LDS). A visited bitmap in global memory costs one L2 atomic per neighbour and, with ~100 bitmaps per
//                 XCD, thrashes the L2 (rocprofv3: 101 GB of write-backs per 2500-query launch) - in LDS it is free, and cheap
//                 enough to read ahead, so the lookups of the NEXT candidate are prefetched while the current one is expanded.
//   VLDS = false: visited bitmap in global, Hf in LDS (large n).
// Semantics identical to search_layer_block (same closed-form accept rule, same pruning of dead candidates): ids, distances
// and evaluation counts are bit-identical.
// ======================================================================================================
constexpr int DT = 512;           // lanes per dense-mode workgroup (two halves of 256: maxdeg <= 256)
constexpr int DWIN = 64;          // keys of C mirrored in LDS
constexpr int DCN = 432;          // capacity of the LDS-resident candidate buffer N (>= 2M: an empty N takes a whole expansion; <= DT: one key per lane)
constexpr int TMAXI = 2;          // staged T keys per lane in a merge (knbn <= TMAXI*DT)
constexpr int HB = 64;            // histogram bins per H1 block
constexpr int HG = 16;            // histogram bins per H2 counter (HB / HG counters per block): 2.2 kB of LDS at m = 18000 (8-bin counters took
                                  // 4.5 kB - the difference keeps three workgroups per CU up to n = 318 k, the size of the NCBI prokaryote set)
struct DenseLds { uint64_t *T, *A, *As, *W, *N; uint32_t *Hf, *H2, *H1, *P1, *vis, *Eid, *Ecnt, *hist, *wsum; uint64_t *scal; };
__host__ __device__ inline uint32_t dense_nblocks(uint32_t m) { return m / HB + 1; }
__host__ __device__ inline size_t dense_lds_bytes(uint32_t m, uint32_t knbn, uint32_t maxdeg, uint64_t n, bool vlds, uint32_t dcn = DCN)
{
    const size_t nb = dense_nblocks(m);
    size_t histb = 4 * ((size_t)dcn + 8); if (histb < 4 * nb) histb = 4 * nb;                   // fold histogram, aliased by P1
    return 8 * (size_t)((knbn + 1) & ~1u) + 8 * (size_t)maxdeg /*A*/ + 8 * (DWIN + 4) + 8 * (size_t)(dcn + 4) + 64 + 4 * nb * (HB / HG / 2) /*H2*/ + 4 * nb /*H1*/ +
           8 * (size_t)maxdeg /*Eid,Ecnt (aliased by As)*/ + histb + 4 * 48 + (vlds ? 4 * (size_t)((n + 31) / 32 + 1) : 4 * nb * (HB / 2));
}
__device__ __forceinline__ DenseLds carve_dense(uint8_t *base, uint32_t m, uint32_t knbn, uint32_t maxdeg, uint64_t n, bool vlds, uint32_t dcn)
{
    DenseLds S;
    const size_t nb = dense_nblocks(m);
    size_t histb = 4 * ((size_t)dcn + 8); if (histb < 4 * nb) histb = 4 * nb;
    S.T = (uint64_t *)base; base += 8 * (size_t)((knbn + 1) & ~1u);
    S.A = (uint64_t *)base; base += 8 * (size_t)maxdeg;
    S.W = (uint64_t *)base; base += 8 * (DWIN + 4);       // + 3 sentinels (~0) behind the last key, so the head reads need no bounds tests
    S.N = (uint64_t *)base; base += 8 * (size_t)(dcn + 4);
    S.scal = (uint64_t *)base; base += 64;
    S.Eid = (uint32_t *)base; S.As = (uint64_t *)base; base += 4 * (size_t)maxdeg;     // As (compaction of accepted keys) reuses Eid/Ecnt, dead by then
    S.Ecnt = (uint32_t *)base; base += 4 * (size_t)maxdeg;
    S.H2 = (uint32_t *)base; base += 4 * nb * (HB / HG / 2);
    S.H1 = (uint32_t *)base; base += 4 * nb;
    S.hist = (uint32_t *)base; S.P1 = (uint32_t *)base; base += histb;                  // P1 (slow accept path) and hist (fold) are never live together
    S.wsum = (uint32_t *)base; base += 4 * 48;                                          // 5 call-site private slots of 8 words
    S.Hf = nullptr; S.vis = nullptr;
    if (vlds) S.vis = (uint32_t *)base; else S.Hf = (uint32_t *)base;
    return S;
}
// 16-bit counters packed two per word
__device__ __forceinline__ uint32_t h16(const uint32_t *H, uint32_t b) { return (H[b >> 1] >> ((b & 1) * 16)) & 0xFFFFu; }
__device__ __forceinline__ void h16add(uint32_t *H, uint32_t b, uint32_t v) { atomicAdd(&H[b >> 1], v << ((b & 1) * 16)); }
__device__ __forceinline__ void h16sub(uint32_t *H, uint32_t b, uint32_t v) { atomicSub(&H[b >> 1], v << ((b & 1) * 16)); }
// fine bins: in LDS, or in global memory where they are only ever touched by workgroup-scope atomics (performed in the L2), so a
// read goes through an atomic as well rather than through the (non-coherent) vector L1
template <bool VLDS> __device__ __forceinline__ uint32_t hf_word(uint32_t *Hf, uint32_t w)
{
    if (VLDS) return __hip_atomic_fetch_add(&Hf[w], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return Hf[w];
}
template <bool VLDS> __device__ __forceinline__ uint32_t hf_get(uint32_t *Hf, uint32_t b) { return (hf_word<VLDS>(Hf, b >> 1) >> ((b & 1) * 16)) & 0xFFFFu; }
struct Hist3 { uint32_t *Hf, *H2, *H1; };
template <bool VLDS> __device__ __forceinline__ void hist_add(const Hist3 &h, uint32_t c, uint32_t v)
{
    if (VLDS) __hip_atomic_fetch_add(&h.Hf[c >> 1], v << ((c & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicAdd(&h.Hf[c >> 1], v << ((c & 1) * 16));
    h16add(h.H2, c / HG, v); atomicAdd(&h.H1[c / HB], v);
}
template <bool VLDS> __device__ __forceinline__ void hist_sub(const Hist3 &h, uint32_t c, uint32_t v)
{
    if (VLDS) __hip_atomic_fetch_sub(&h.Hf[c >> 1], v << ((c & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else atomicSub(&h.Hf[c >> 1], v << ((c & 1) * 16));
    h16sub(h.H2, c / HG, v); atomicSub(&h.H1[c / HB], v);
}
// one wave: highest non-empty bin <= d (d wave-uniform) and its multiplicity; bin 0 / 0 when there is none
template <bool VLDS> __device__ __forceinline__ uint32_t hist_find_down(const Hist3 &h, uint32_t d, uint32_t lane, uint32_t &mult)
{
    constexpr uint32_t GPB = HB / HG;                         // H2 counters per H1 block
    const uint32_t lf = lane % HG, lg = lane % GPB;
    uint32_t grp = d / HG;
    {   // bins of d's own group
        const uint32_t b = grp * HG + lf;
        const uint32_t v = (lane < (uint32_t)HG && b <= d) ? hf_get<VLDS>(h.Hf, b) : 0;
        const uint64_t bal = __ballot(v != 0);
        if (bal) { const uint32_t top = 63 - (uint32_t)__clzll((long long)bal); mult = __shfl(v, top); return grp * HG + top; }
    }
    bool found = false;
    {   // lower groups of d's block
        const uint32_t blk = d / HB, g = blk * GPB + lg;
        const uint32_t v = (lane < GPB && g < grp) ? h16(h.H2, g) : 0;
        const uint64_t bal = __ballot(v != 0);
        if (bal) { grp = blk * GPB + (63 - (uint32_t)__clzll((long long)bal)); found = true; }
        else {
            for (int base = (int)blk - 1; base >= 0 && !found; base -= 64) {
                const int bi = base - (int)lane;
                const uint32_t v1 = bi >= 0 ? h.H1[bi] : 0;
                const uint64_t b1 = __ballot(v1 != 0);
                if (b1) {
                    const uint32_t blk2 = (uint32_t)base - (uint32_t)(__ffsll((long long)b1) - 1);
                    const uint32_t v2 = lane < GPB ? h16(h.H2, blk2 * GPB + lg) : 0;
                    const uint64_t b2 = __ballot(v2 != 0);
                    if (b2) { grp = blk2 * GPB + (63 - (uint32_t)__clzll((long long)b2)); found = true; }
                    else base = -1;                       // inconsistent summaries cannot happen; stop
                }
            }
        }
    }
    if (!found) { mult = 0; return 0; }
    const uint32_t v = lane < (uint32_t)HG ? hf_get<VLDS>(h.Hf, grp * HG + lf) : 0;
    const uint64_t bal = __ballot(v != 0);
    if (!bal) { mult = 0; return 0; }
    const uint32_t top = 63 - (uint32_t)__clzll((long long)bal);
    mult = __shfl(v, top);
    return grp * HG + top;
}
// Barrier that only orders LDS traffic: global loads issued before it (prefetches) stay in flight across it. Everything the
// dense traversal exchanges between lanes inside a pop goes through LDS; the few places that hand global data between lanes
// (G window refill, fold, histogram bins read back in the slow paths) keep the full __syncthreads().
__device__ __forceinline__ void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// number of lanes of the workgroup with pred set (slot: 8 words of LDS private to the call site)
__device__ __forceinline__ uint32_t lds_count(bool pred, uint32_t *slot)
{
    const uint64_t b = __ballot(pred);
    if ((threadIdx.x & 63) == 0) slot[threadIdx.x >> 6] = (uint32_t)__popcll(b);
    lds_barrier();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < DT / 64; w++) t += slot[w];
    return t;
}
// merge sorted A (na <= DT keys) into the sorted T[0..n) keeping `keep` keys (DT lanes)
__device__ __forceinline__ uint32_t dense_merge_T(uint64_t *keys, uint32_t n, const uint64_t *A, uint32_t na, uint32_t keep, const SmallA &sa)
{
    uint64_t kv[TMAXI]; uint32_t pos[TMAXI];
#pragma unroll
    for (int it = 0; it < TMAXI; it++) {
        const uint32_t idx = threadIdx.x + it * DT;
        pos[it] = 0xFFFFFFFFu; kv[it] = 0;
        if (idx < n) { const uint64_t k = keys[idx]; kv[it] = k; pos[it] = idx + lb_a(sa, k); }
    }
    uint64_t ak = 0; uint32_t apos = 0xFFFFFFFFu;
    if (threadIdx.x < na) { ak = A[threadIdx.x]; apos = threadIdx.x + lower_bound_keys(keys, n, ak); }
    lds_barrier();
#pragma unroll
    for (int it = 0; it < TMAXI; it++) if (pos[it] < keep) keys[pos[it]] = kv[it];
    if (apos < keep) keys[apos] = ak;
    lds_barrier();
    const uint32_t tot = n + na;
    return tot < keep ? tot : keep;
}

// tau = the efs-th smallest mismatch count of the query over ALL n nodes (INF_CNT when n < efs): a lower bound of the worst count a full R
// can ever reach - fewer than efs nodes of the whole database lie below it. Once dmax == tau the accept rule is "count < tau" for good and
// the rest of the traversal does not depend on the order of the pops any more (phase 2 below). One scan of the query's count row (16-byte
// loads, 8 counts per lane): counts are binned 64 to a bin (`coarse`, nb bins) and, for the bin that holds m itself - where the flood
// regime puts the answer - one by one (`fine`); the four largest values (m .. m-3: ~99 % of the nodes of an unrelated database) are
// tallied with ballots instead of LDS atomics on one address. A second scan only when the answer lies in another bin.
__device__ __forceinline__ uint32_t dense_row_tau(const uint16_t *__restrict__ matrow, uint64_t n, uint32_t m, uint32_t efs, uint32_t *coarse, uint32_t *fine, uint32_t *res)
{
    if (n < efs) return INF_CNT;
    const uint32_t nb = dense_nblocks(m), lane = threadIdx.x & 63;
    const uint4 *row16 = (const uint4 *)matrow;
    uint32_t target = m / HB;
    for (int pass = 0; pass < 2; pass++) {
        for (uint32_t i = threadIdx.x; i < nb; i += DT) coarse[i] = 0;
        if (threadIdx.x < (uint32_t)HB) fine[threadIdx.x] = 0;
        __syncthreads();
        // per lane: byte-wide tallies of m, m-1, m-2, m-3 packed in one register (flushed every 16 loads: 128 counts < 256), anything
        // lower goes to the LDS bins one by one (~1 % of an unrelated database)
        uint32_t top[4] = {0, 0, 0, 0}, acc = 0, since = 0;
        auto tally = [&](uint32_t c) {
            const uint32_t d = m - c;                          // c <= m always
            acc += d <= 3 ? (1u << (8 * d)) : 0u;
            if (d > 3) { atomicAdd(&coarse[c / HB], 1u); if (c / HB == target) atomicAdd(&fine[c % HB], 1u); }
        };
        const uint32_t nfull = (uint32_t)(n / 8);              // groups of 8 counts that lie entirely inside the row
        for (uint32_t i = threadIdx.x; i < nfull; i += DT) {
            const uint4 v = row16[i];
            tally(v.x & 0xFFFFu); tally(v.x >> 16); tally(v.y & 0xFFFFu); tally(v.y >> 16);
            tally(v.z & 0xFFFFu); tally(v.z >> 16); tally(v.w & 0xFFFFu); tally(v.w >> 16);
            if (++since == 16) { top[0] += acc & 255u; top[1] += (acc >> 8) & 255u; top[2] += (acc >> 16) & 255u; top[3] += acc >> 24; acc = 0; since = 0; }
        }
        if (threadIdx.x == 0) for (uint64_t e = (uint64_t)nfull * 8; e < n; e++) tally(matrow[e]);      // (the padding behind the row is not data)
        top[0] += acc & 255u; top[1] += (acc >> 8) & 255u; top[2] += (acc >> 16) & 255u; top[3] += acc >> 24;
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) top[t] += __shfl_xor(top[t], o);
        }
        if (lane == 0) {
#pragma unroll
            for (int t = 0; t < 4; t++) if (top[t] && m >= (uint32_t)t) { const uint32_t c = m - t; atomicAdd(&coarse[c / HB], top[t]); if (c / HB == target) atomicAdd(&fine[c % HB], top[t]); }
        }
        __syncthreads();
        if (threadIdx.x < 64) {                               // one wave: the bin where the running total reaches efs
            uint32_t run = 0, found = 0xFFFFFFFFu, before = 0;
            for (uint32_t b0 = 0; b0 < nb && found == 0xFFFFFFFFu; b0 += 64) {
                const uint32_t x = b0 + lane < nb ? coarse[b0 + lane] : 0;
                uint32_t inc = x;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
                const uint64_t hit = __ballot(run + inc >= efs);
                if (hit) { const uint32_t l = (uint32_t)__ffsll((long long)hit) - 1; found = b0 + l; before = run + __shfl(inc, l) - __shfl(x, l); }
                run += __shfl(inc, 63);
            }
            if (lane == 0) { res[0] = found; res[1] = before; }
        }
        __syncthreads();
        const uint32_t bin = res[0], before = res[1];
        if (bin == target) {
            if (threadIdx.x == 0) { uint32_t run = before, f = 0; for (; f < (uint32_t)HB; f++) { run += fine[f]; if (run >= efs) break; } res[2] = bin * HB + f; }
            __syncthreads();
            const uint32_t tau = res[2];
            __syncthreads();
            return tau;
        }
        target = bin;                                          // rare: the answer is not near m - tally that bin one by one in a second scan
        __syncthreads();
    }
    return INF_CNT;                                            // (not reached: the second pass always finds its bin)
}

// a wave-uniform value the compiler cannot prove uniform (it came through LDS): pin it to an SGPR
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// ================= phase 2: the order-free rest of search_layer =================
// dmax == tau, the efs-th smallest count of the whole database: fewer than efs nodes lie below it, so dmax can never drop
// again, the accept rule is "count < tau" from here on, every accepted key stays and is popped, and so is every candidate
// already waiting with count <= tau. The set of nodes that get evaluated - hence ids, distances AND the evaluation count -
// is the closure of the waiting candidates under "neighbours with count < tau", whatever the order of the pops. So the
// candidates become a plain work list that the eight wavefronts drain independently, one adjacency row per wavefront and
// step, with one barrier per generation instead of several per pop; the 2-byte count look-up (an HBM sector each) is only
// made for neighbours that pass a Bloom filter of the < efs nodes below tau, built from one scan of the query's count row
// in LDS that phase 1 no longer needs.
// (kept as a function for readability; making it a real call - noinline - was tried: the whole kernel then pays for a stack, 124 -> 490 ms)

struct Phase2IO { uint32_t nT, evals, pops; bool tm; long long c_build, c_drain; uint64_t *wl; uint32_t cap_log, nlog; };
// SPLIT (round 5, indexes beyond what one workgroup's LDS can map): the visited bitmap covers node ids [0, vis_w) in LDS and [vis_w, n) in this
// workgroup's global scratch `visg`. Here, where nothing depends on the order of the pops, a HIGH id (>= vis_w) goes to the Bloom filter FIRST: the
// ~98 % that cannot lie below tau are dropped without touching memory at all, only the rest pays a memory-side test-and-set + the count look-up. What
// that skips is the evaluation COUNT of the dropped ids - so after the drain the popped nodes' adjacency rows are walked once more per LDS-sized
// range of high ids, with the range's global bits (phase 1's marks and the Bloom positives, both counted when they were set) copied into the LDS
// bitmap: every high id that is still clear there is a distinct evaluation. ids, distances and evaluation counts stay the oracle's.
template <bool ONEG, bool WLOG, bool SPLIT>
__device__ __forceinline__ void dense_phase2(const IndexDev &ix, const DenseLds &S, uint32_t *vis, const uint16_t *__restrict__ matrow, uint32_t tau, uint32_t knbn,
                                                       uint64_t *Gold, uint64_t *Gnew, uint32_t headG, uint32_t nG, uint32_t headN, uint32_t nN, uint64_t Tmax, Phase2IO &io,
                                                       uint32_t vis_w, uint32_t *__restrict__ visg)
{
    const uint32_t maxdeg = 2 * ix.M;
    constexpr uint32_t CN = ONEG ? 512u : (uint32_t)DCN;
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t nT = io.nT, st_p2 = 0;
    uint32_t *s_cnt = (uint32_t *)&S.scal[4];                // [0] work-list tail, [1] keys for T, [2] evaluations, [3] log length
    uint32_t *WL = (uint32_t *)Gnew;                  // node ids to expand (capacity 2 * capC >= waiting + < efs accepted)
    uint64_t *TA = Gold;                                  // accepted keys that may enter T (the old G array: consumed below first)
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = threadIdx.x == 3 ? io.nlog : 0u;      // [3] WLOG: length of the accepted-key log
    __syncthreads();
    {   // waiting candidates with count <= tau: live G (global, sorted) and live N (LDS)
        const uint64_t *src = Gold + headG;
        const uint32_t liveG = nG - headG, liveN = nN - headN;
        for (uint32_t i = threadIdx.x; i < liveG; i += DT) { const uint64_t k = src[i]; if (KCNT(k) <= tau) WL[atomicAdd(&s_cnt[0], 1u)] = KID(k); }
        for (uint32_t i = threadIdx.x; i < liveN; i += DT) { const uint64_t k = S.N[headN + i]; if (KCNT(k) <= tau) WL[atomicAdd(&s_cnt[0], 1u)] = KID(k); }
    }
    __syncthreads();
    // Bloom filter of {node : count < tau} in two dead LDS regions (A + G window + N, and Eid .. fold histogram), one hash each
    uint32_t *bfA = (uint32_t *)S.A, *bfB = S.Eid;
    const uint32_t bytesA = 8 * maxdeg + 8 * (DWIN + 4) + 8 * (CN + 4), bytesB = (uint32_t)((uint8_t *)S.wsum - (uint8_t *)S.Eid);
    uint32_t lgA = 5, lgB = 5;
    while ((2u << lgA) <= 8 * bytesA) lgA++;
    while ((2u << lgB) <= 8 * bytesB) lgB++;
    for (uint32_t w = threadIdx.x; w < (1u << (lgA - 5)); w += DT) bfA[w] = 0;
    for (uint32_t w = threadIdx.x; w < (1u << (lgB - 5)); w += DT) bfB[w] = 0;
    __syncthreads();
    {
        const uint4 *row16 = (const uint4 *)matrow;
        const uint32_t n16 = (uint32_t)((ix.n + 7) / 8);
        for (uint32_t i = threadIdx.x; i < n16; i += DT) {
            const uint4 v = row16[i];
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int h = 0; h < 8; h++) {
                const uint32_t cc = (w4[h >> 1] >> ((h & 1) * 16)) & 0xFFFFu, id = i * 8 + h;
                if (cc < tau && id < ix.n) {
                    const uint32_t ha = (id * 0x9E3779B1u) >> (32 - lgA), hb = (id * 0x85EBCA77u) >> (32 - lgB);
                    atomicOr(&bfA[ha >> 5], 1u << (ha & 31)); atomicOr(&bfB[hb >> 5], 1u << (hb & 31));
                }
            }
        }
    }
    const long long pt0 = io.tm ? clock64() : 0;
    const bool t_open = nT < knbn;                           // T not full: every accepted key is a candidate for it
    const uint64_t t_max = Tmax;
    uint32_t nev = 0, head = 0;
    constexpr int NR = ONEG ? 8 : 4;                         // adjacency ids per lane: 64 * NR >= maxdeg
    // one adjacency row per wavefront and step, software-pipelined three deep: the node id of the item after next, the row of
    // the next item and the work on this one are all in flight together (items are independent: order does not matter here)
    auto expand = [&](const uint32_t (&rid)[NR], uint32_t deg) {
#pragma unroll
        for (int j = 0; j < NR; j++) {
            const uint32_t idx = lane + 64 * j;
            if (idx >= deg) continue;
            const uint32_t id = rid[j];
            const uint32_t bit = 1u << (id & 31);
            const uint32_t ha = (id * 0x9E3779B1u) >> (32 - lgA), hb = (id * 0x85EBCA77u) >> (32 - lgB);
            if (SPLIT && id >= vis_w) {
                if (!((bfA[ha >> 5] >> (ha & 31)) & (bfB[hb >> 5] >> (hb & 31)) & 1u)) continue;  // certainly not below tau: counted by the walk after the drain
                const uint32_t oldg = __hip_atomic_fetch_or(&visg[(id - vis_w) >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (oldg & bit) continue;
                nev++;
            } else {
                const uint32_t old = atomicOr(&vis[id >> 5], bit);
                if (old & bit) continue;
                nev++;
                if (!((bfA[ha >> 5] >> (ha & 31)) & (bfB[hb >> 5] >> (hb & 31)) & 1u)) continue;      // certainly not below tau
            }
            const uint32_t cc = matrow[id];
            if (cc >= tau) continue;
            WL[atomicAdd(&s_cnt[0], 1u)] = id;               // accepted: expanded in the next generation
            const uint64_t key = KEY(cc, id);
            if (t_open || key < t_max) TA[atomicAdd(&s_cnt[1], 1u)] = key;
            if (WLOG) { const uint32_t lp = atomicAdd(&s_cnt[3], 1u); if (lp < io.cap_log) io.wl[lp] = key; }      // the insert pre-pass wants every accepted key
        }
    };
    auto fetch_row = [&](uint32_t node, uint32_t (&rid)[NR], uint32_t &deg) {
        deg = ix.deg0[node];
        const uint32_t *row = ix.nbr0 + (uint64_t)node * maxdeg;
#pragma unroll
        // (nontemporal, like every adjacency row of this kernel: a row is read once per query and only competes with the queries' count rows for the Infinity Cache;
        // 38.8 -> 38.4 ms per 10 000 queries, two A/B pairs on one box)
        for (int j = 0; j < NR; j++) { const uint32_t idx = lane + 64 * j; rid[j] = idx < maxdeg ? __builtin_nontemporal_load(row + idx) : 0u; }
    };
    constexpr uint32_t WS = DT / 64;
    for (;;) {
        __syncthreads();
        const uint32_t tail = s_cnt[0];
        if (head >= tail) break;
        uint32_t itx = head + wv;
        uint32_t rowB[NR], degB = 0, nodeA = 0;
        if (itx < tail) fetch_row(uni32(WL[itx]), rowB, degB);
        if (itx + WS < tail) nodeA = WL[itx + WS];
        for (; itx < tail; itx += WS) {
            uint32_t rowC[NR], degC = degB;
#pragma unroll
            for (int j = 0; j < NR; j++) rowC[j] = rowB[j];
            const uint32_t nn = uni32(nodeA);
            if (itx + WS < tail) fetch_row(nn, rowB, degB);
            if (itx + 2 * WS < tail) nodeA = WL[itx + 2 * WS];
            expand(rowC, degC);
        }
        st_p2 += tail - head;
        head = tail;
    }
    if (io.tm) { io.c_build = pt0; io.c_drain = clock64(); }
    if (SPLIT && vis_w < ix.n) {
        // the evaluation count of the high ids the Bloom filter dropped: one more walk over the popped nodes' rows per range of vis_w high ids
        const uint32_t npop = head, lw = vis_w >> 5;
        for (uint64_t base64 = vis_w; base64 < ix.n; base64 += vis_w) {
            const uint32_t base = (uint32_t)base64;
            const uint32_t span = (uint32_t)(ix.n - base64 < (uint64_t)vis_w ? ix.n - base64 : (uint64_t)vis_w), sw = (span + 31) >> 5;
            __syncthreads();
            for (uint32_t w = threadIdx.x; w < lw; w += DT)
                vis[w] = w < sw ? __hip_atomic_load(&visg[((base - vis_w) >> 5) + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            __syncthreads();
            uint32_t itx = wv;
            uint32_t rowB[NR], degB = 0, nodeA = 0;
            if (itx < npop) fetch_row(uni32(WL[itx]), rowB, degB);
            if (itx + WS < npop) nodeA = WL[itx + WS];
            for (; itx < npop; itx += WS) {
                uint32_t rowC[NR], degC = degB;
#pragma unroll
                for (int j = 0; j < NR; j++) rowC[j] = rowB[j];
                const uint32_t nn = uni32(nodeA);
                if (itx + WS < npop) fetch_row(nn, rowB, degB);
                if (itx + 2 * WS < npop) nodeA = WL[itx + 2 * WS];
#pragma unroll
                for (int j = 0; j < NR; j++) {
                    const uint32_t idx = lane + 64 * j;
                    if (idx >= degC) continue;
                    const uint32_t r = rowC[j] - base;               // (ids below base wrap around: out of the span)
                    if (r >= span) continue;
                    const uint32_t bit = 1u << (r & 31);
                    if (!(atomicOr(&vis[r >> 5], bit) & bit)) nev++;
                }
            }
        }
    }
    if (nev) atomicAdd(&s_cnt[2], nev);
    __syncthreads();
    // T <- knbn smallest of T u TA, in chunks the existing merge takes (sorted A of at most maxdeg keys; A's region is free again)
    const uint32_t nta = s_cnt[1];
    const uint32_t CHK = maxdeg < (uint32_t)DT ? maxdeg : (uint32_t)DT;
    for (uint32_t c0 = 0; c0 < nta; c0 += CHK) {
        const uint32_t na = nta - c0 < CHK ? nta - c0 : CHK;
        __syncthreads();
        if (threadIdx.x < na) S.As[threadIdx.x] = TA[c0 + threadIdx.x];
        __syncthreads();
        if (threadIdx.x < na) {
            const uint64_t k = S.As[threadIdx.x];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < na; j++) rank += (S.As[j] < k);
            S.A[rank] = k;
        }
        __syncthreads();
        const SmallA sa = load_small_a(S.A, na);
        nT = dense_merge_T(S.T, nT, S.A, na, knbn, sa);
    }
    __syncthreads();
    io.nT = nT; io.evals += s_cnt[2]; io.pops = st_p2; io.nlog = s_cnt[3];
}

// wave-wide register moves for the PQ front (one key per lane of wavefront 0): DPP wave shifts and v_readlane instead of __shfl (ds_bpermute: an LDS
// round trip per 32 bits - the 64-bit min-reduction and the shuffles of one insertion cost more than the merge they replaced)
__device__ __forceinline__ uint64_t wave_shr1_u64(uint64_t v)          // lane i <- lane i - 1 (lane 0 keeps its value)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x138, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xF, 0xF, false);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t wave_shl1_u64(uint64_t v)          // lane i <- lane i + 1 (lane 63 keeps its value)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)v, (int)(uint32_t)v, 0x130, 0xF, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(v >> 32), (int)(uint32_t)(v >> 32), 0x130, 0xF, 0xF, false);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l)   // l wave-uniform
{
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32);
}
// PQ (late round 5; see the kernel): refill of the front / compaction of the overflow U. All lanes call. src[0..nU) -> dst: keys whose count lies above
// `dlimit` are dropped (dead: nothing that large is ever popped again). sel: the front is empty and takes the <= 64 smallest keys of what is left:
// a histogram over the count LEVELS (relative to the smallest live key's) says which levels fit whole, and inside the level that straddles the quota
// an id boundary is found by histogram refinement - 256 bins over the id range, as many leading bins as fit, then the bin that straddles the rest is
// binned again (ids are distinct inside a level, so bins of one id always resolve; <= 4 rounds for 2^32 ids, one in practice). The selection is
// "key < cutkey". The level the front was last cut in is, as a rule, where the next refill starts, so its two histograms are made speculatively
// during the pass that finds the smallest key: a refill is then two passes over U (four keys in flight per lane), three when the guess fails.
// Leaves the new front in wavefront 0's registers and its first three keys in S.W[fb..], the new length of U in S.scal[4] (the caller swaps the
// buffers). LDS scratch: S.hist[0..256) (ids), S.N[128..256) (levels), S.N[0..128) (staging), S.scal[5..6], S.wsum[32..40).
__device__ __forceinline__ void pq_rebuild(const IndexDev &ix, const DenseLds &S, const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint32_t capC, uint32_t nU,
                                           bool sel, uint32_t dlimit, uint64_t &pkey, uint64_t &pv, uint32_t &nP, uint32_t fb, unsigned long long &stat)
{
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = nU < capC ? nU : capC;
    uint32_t rounds = 0;
    unsigned long long *smin = (unsigned long long *)&S.scal[5];
    uint32_t *cnt = (uint32_t *)&S.scal[6];                  // [0] selected, [1] kept
    uint32_t *ctl = S.wsum + 32;                             // [0] done, [1..2] cut (64 bit), [3] quota used so far, [4] the level the histograms were made for, [5] levels that fit whole
    uint32_t *lvl = (uint32_t *)(S.N + 128);                 // 256 level bins
    constexpr int NF = 4;                                    // keys of U in flight per lane: the passes are chains of global round trips otherwise
    __syncthreads();                                         // every wave is done with the pop before (S.scal[4..5], its LDS scratch)
    if (threadIdx.x == 0) {
        *smin = ~0ull; cnt[0] = 0; cnt[1] = 0;
        ctl[4] = pv != ~(uint64_t)0 ? KCNT(pv) : 0xFFFFFFFFu;
    }
    if (threadIdx.x < 256) { S.hist[threadIdx.x] = 0; lvl[threadIdx.x] = 0; }
    __syncthreads();                                         // U's appends (global stores of wavefront 0) are in; the pop's LDS scratch is dead
    uint64_t cutkey = 0;
    uint32_t bits = 1; while (bits < 32 && ((uint64_t)1 << bits) < ix.n) bits++;
    // level tally: the two or three levels that hold nearly everything go through ballots (one LDS atomic per wave instead of one per key on one address)
    auto tally_level = [&](bool on, uint32_t d) {
#pragma unroll
        for (uint32_t t = 0; t < 3; t++) { const uint64_t bb = __ballot(on && d == t); if (bb && lane == 0) atomicAdd(&lvl[t], (uint32_t)__popcll(bb)); }
        if (on && d >= 3) atomicAdd(&lvl[d < 255u ? d : 255u], 1u);
    };
    if (sel) {
        const uint32_t guess = ctl[4], sh0 = bits > 8 ? bits - 8 : 0;
        uint64_t mn = ~(uint64_t)0;
        for (uint32_t i0 = 0; i0 < n; i0 += NF * DT) {
            uint64_t kf[NF];
#pragma unroll
            for (int u = 0; u < NF; u++) { const uint32_t i = i0 + u * DT + threadIdx.x; kf[u] = i < n ? src[i] : ~(uint64_t)0; }
#pragma unroll
            for (int u = 0; u < NF; u++) {
                const uint64_t k = kf[u];
                const bool live = k != ~(uint64_t)0 && KCNT(k) <= dlimit;
                if (live && k < mn) mn = k;
                tally_level(live && KCNT(k) >= guess, KCNT(k) - guess);
                if (live && KCNT(k) == guess) atomicAdd(&S.hist[KID(k) >> sh0], 1u);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint64_t y = __shfl_xor(mn, o); mn = y < mn ? y : mn; }
        if (lane == 0 && mn != ~(uint64_t)0) atomicMin(smin, (unsigned long long)mn);
        __syncthreads();
        const uint64_t kmin = *smin;
        if (kmin != ~(uint64_t)0) {
            const uint32_t cmin = KCNT(kmin);
            if (cmin != guess) {                             // the guess failed (first refill of a query, a level ran out): the level tally again, from cmin
                __syncthreads();
                if (threadIdx.x < 256) lvl[threadIdx.x] = 0;
                __syncthreads();
                for (uint32_t i0 = 0; i0 < n; i0 += NF * DT) {
                    uint64_t kf[NF];
#pragma unroll
                    for (int u = 0; u < NF; u++) { const uint32_t i = i0 + u * DT + threadIdx.x; kf[u] = i < n ? src[i] : ~(uint64_t)0; }
#pragma unroll
                    for (int u = 0; u < NF; u++) { const uint64_t k = kf[u]; tally_level(k != ~(uint64_t)0 && KCNT(k) <= dlimit, KCNT(k) - cmin); }     // (nothing live lies below cmin)
                }
                __syncthreads();
            }
            if (wv == 0) {                                   // the levels that fit whole
                const uint32_t h0 = lvl[4 * lane], h1 = lvl[4 * lane + 1], h2 = lvl[4 * lane + 2], h3 = lvl[4 * lane + 3];
                const uint32_t sum = h0 + h1 + h2 + h3;
                uint32_t inc = sum;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
                const uint32_t c0 = inc - sum + h0, c1 = c0 + h1, c2 = c1 + h2, c3 = c2 + h3;
                uint32_t j = (c0 <= 64u) + (c1 <= 64u) + (c2 <= 64u) + (c3 <= 64u);
                uint32_t tk = c3 <= 64u ? c3 : (c2 <= 64u ? c2 : (c1 <= 64u ? c1 : (c0 <= 64u ? c0 : 0u)));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { j += __shfl_xor(j, o); const uint32_t y = __shfl_xor(tk, o); tk = y > tk ? y : tk; }
                if (lane == 0) { ctl[5] = j; ctl[3] = tk; }
            }
            __syncthreads();
            const uint32_t Lc = ctl[5];
            uint32_t acc = ctl[3];
            if (Lc >= 255u) cutkey = Lc == 256u ? ~(uint64_t)0 : KEY(cmin + 255u, 0);      // everything fits / everything below the clipped tail does
            else if (acc >= 48u) cutkey = KEY(cmin + Lc, 0);                                 // whole levels fill the front well enough
            else {
                // the level that straddles the quota: an id boundary inside it
                const uint32_t lc = cmin + Lc;
                bool have_hist = Lc == 0 && cmin == guess;
                uint32_t lo = 0; uint64_t cut = 0;
                for (;;) {
                    const uint32_t sh = bits > 8 ? bits - 8 : 0;
                    if (!have_hist) {
                        __syncthreads();                     // the round before has read hist / ctl
                        if (threadIdx.x < 256) S.hist[threadIdx.x] = 0;
                        __syncthreads();
                        for (uint32_t i0 = 0; i0 < n; i0 += NF * DT) {
                            uint64_t kf[NF];
#pragma unroll
                            for (int u = 0; u < NF; u++) { const uint32_t i = i0 + u * DT + threadIdx.x; kf[u] = i < n ? src[i] : ~(uint64_t)0; }
#pragma unroll
                            for (int u = 0; u < NF; u++) {
                                const uint64_t k = kf[u];
                                if (KCNT(k) == lc && KID(k) >= lo) { const uint32_t r = KID(k) - lo; if ((r >> sh) < 256u) atomicAdd(&S.hist[r >> sh], 1u); }
                            }
                        }
                        __syncthreads();
                    }
                    have_hist = false;
                    if (wv == 0) {
                        const uint32_t h0 = S.hist[4 * lane], h1 = S.hist[4 * lane + 1], h2 = S.hist[4 * lane + 2], h3 = S.hist[4 * lane + 3];
                        const uint32_t sum = h0 + h1 + h2 + h3;
                        uint32_t inc = sum;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
                        const uint32_t room = 64u - acc;
                        const uint32_t c0 = inc - sum + h0, c1 = c0 + h1, c2 = c1 + h2, c3 = c2 + h3;      // inclusive running totals: monotone, so the bins that fit are a prefix
                        uint32_t j = (c0 <= room) + (c1 <= room) + (c2 <= room) + (c3 <= room);
                        uint32_t tk = c3 <= room ? c3 : (c2 <= room ? c2 : (c1 <= room ? c1 : (c0 <= room ? c0 : 0u)));
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) { j += __shfl_xor(j, o); const uint32_t y = __shfl_xor(tk, o); tk = y > tk ? y : tk; }
                        if (lane == 0) {
                            const bool all = j == 256u;
                            const uint64_t cu = all ? (uint64_t)lo + ((uint64_t)256 << sh) : (uint64_t)lo + ((uint64_t)j << sh);
                            ctl[0] = (all || sh == 0 || acc + tk >= 48u) ? 1u : 0u;      // (a fuller front is refilled less often: worth one more pass below 48 keys)
                            ctl[1] = (uint32_t)cu; ctl[2] = (uint32_t)(cu >> 32); ctl[3] = acc + tk;
                        }
                    }
                    __syncthreads();
                    cut = (uint64_t)ctl[1] | ((uint64_t)ctl[2] << 32); acc = ctl[3];
                    if (ctl[0]) break;
                    lo = (uint32_t)cut; bits = sh; rounds++;  // inside the bin that straddles the quota
                }
                cutkey = ((uint64_t)lc << 32) + cut;          // (cut = 2^32: the whole level)
            }
        }
    }
    __syncthreads();
    for (uint32_t i0 = 0; i0 < n; i0 += NF * DT) {
        uint64_t kf[NF];
#pragma unroll
        for (int u = 0; u < NF; u++) { const uint32_t i = i0 + u * DT + threadIdx.x; kf[u] = i < n ? src[i] : ~(uint64_t)0; }
#pragma unroll
        for (int u = 0; u < NF; u++) {
            const uint64_t k = kf[u];
            const bool live = k != ~(uint64_t)0 && KCNT(k) <= dlimit;
            const bool isS = live && k < cutkey, isK = live && !isS;
            const uint64_t bS = __ballot(isS), bK = __ballot(isK);
            uint32_t baseS = 0, baseK = 0;
            if (lane == 0) { if (bS) baseS = atomicAdd(&cnt[0], (uint32_t)__popcll(bS)); if (bK) baseK = atomicAdd(&cnt[1], (uint32_t)__popcll(bK)); }
            baseS = (uint32_t)__builtin_amdgcn_readfirstlane((int)baseS); baseK = (uint32_t)__builtin_amdgcn_readfirstlane((int)baseK);
            if (isS) { const uint32_t pos = baseS + (uint32_t)__popcll(bS & ((1ull << lane) - 1)); if (pos < 64u) S.N[pos] = k; }
            if (isK) dst[baseK + (uint32_t)__popcll(bK & ((1ull << lane) - 1))] = k;
        }
    }
    __syncthreads();
    const uint32_t nsel = cnt[0] < 64u ? cnt[0] : 64u, nkeep = cnt[1];
    if (sel && wv == 0) {                                    // rank sort of the <= 64 selected keys
        const uint64_t k = lane < nsel ? S.N[lane] : ~(uint64_t)0;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nsel; j++) rank += (S.N[j] < k);
        if (lane < nsel) S.N[64 + rank] = k;
    }
    __syncthreads();
    if (sel && wv == 0) {
        pkey = lane < nsel ? S.N[64 + lane] : ~(uint64_t)0;
        nP = nsel;
        pv = (nkeep && nsel) ? readlane_u64(pkey, (int)nsel - 1) : ~(uint64_t)0;      // everything left in U is larger than everything taken
        if (lane < 3) S.W[fb + lane] = pkey;
    }
    if (threadIdx.x == 0) S.scal[4] = nkeep;
    // (work counters of this workgroup, flushed with the others at the end of the kernel: refills in bits 0..23, compactions in 24..39, selection rounds beyond the first in 40..63)
    stat += (sel ? 1ull : (1ull << 24)) + ((unsigned long long)rounds << 40);
    __syncthreads();
}

// ONEG: max_nb_conn > 128 (adjacency rows of up to 512 ids): all 512 lanes form ONE group that expands a candidate, then does the visited
// hint + lookups of the next one and fetches the row of the one after - the three stages the two 256-lane halves otherwise share out
// WLOG (insert-time pre-pass): every accepted key is also appended to a per-workgroup log; R is always "the ef smallest
// accepted keys", so at the end of the query the log, cut at dmax and sorted, IS the sorted result set W of search_layer - which the
// histogram form of R never materialises. Queries flagged in `skip` are left to the caller.
template <bool VLDS, bool PROF, int OCC, bool ONEG, bool WLOG, bool SPLIT = false>
__global__ __launch_bounds__(DT, OCC) void k_hnsw_search_dense(IndexDev ix, uint64_t nq, uint32_t knbn, uint32_t ef, const uint16_t *__restrict__ mat, uint64_t mat_ld,
                                                           uint32_t *__restrict__ scratch, uint32_t scratch_words, uint64_t *__restrict__ cbuf, uint32_t capC,
                                                           unsigned long long *__restrict__ counter, uint64_t *__restrict__ ids_out, float *__restrict__ dist_out,
                                                           uint32_t *__restrict__ count_out, uint64_t *__restrict__ evals_out, unsigned long long *__restrict__ prof,
                                                           unsigned long long *__restrict__ stats, uint64_t *__restrict__ wlog, uint32_t cap_log, uint32_t sort_cap,
                                                           uint64_t *__restrict__ w_out, uint32_t *__restrict__ w_n, const uint32_t *__restrict__ ep_in, uint32_t vis_w_arg)
{
    static_assert(!SPLIT || VLDS, "the split bitmap is a form of the LDS placement");
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    const uint32_t maxdeg = 2 * ix.M;
    const uint32_t efs = ef > knbn ? ef : knbn;
    // SPLIT: node ids [0, vis_w) are mapped by the LDS bitmap, [vis_w, n) by `visg`, behind this workgroup's fine histogram bins in the global scratch
    const uint32_t vis_w = SPLIT ? vis_w_arg : 0u;
    const uint32_t nb = dense_nblocks(ix.m), hwords = nb * (HB / 2), vis_words = SPLIT ? (vis_w >> 5) : (uint32_t)((ix.n + 31) / 32);
    const uint32_t visg_words = SPLIT && ix.n > vis_w ? (uint32_t)((ix.n - vis_w + 31) / 32) : 0u;
    unsigned long long st_pq = 0;                                        // PQ: refills / compactions / extra selection rounds (pq_rebuild)
    uint32_t st_pops = 0, st_acc = 0, st_p1 = 0, st_p2 = 0;                 // work counters (workgroup-uniform): pops / accepting pops / pops before dmax reached tau, of this workgroup (< 2^32)
    constexpr bool PHASE2 = true;       // (round 4: also with the visited bitmap in global memory - indexes beyond ~600 k nodes: test-and-set through L2 atomics)
    // PQ (late round 5): the waiting candidates are a FRONT of up to 64 keys, sorted, one per lane in the registers of wavefront 0 (insert = ballot +
    // shuffle, no LDS array to merge into, no barrier per key), over an UNSORTED overflow U in global memory that every key beyond the front's fence `pv`
    // is simply appended to (max(front) <= pv <= min(U)). The front publishes its first three keys to LDS after every change - all the other waves ever
    // need - and is refilled from U by selection (the <= 64 smallest keys of U's best count level, `pq_rebuild`) when it runs empty, every ~50-90 pops in
    // the flood regime. It replaces G (sorted, global) + its LDS window + N (sorted, LDS) and with them the rank sort of the accepted keys, the N merge
    // and the fold of N into G of every accepting pop: 6.7 k of that pop's 10.3 k cycles (NOTES.md round 4, GS_TRAV_PROFILE). -DGS_DENSE_PQ_OFF: the old form.
#ifdef GS_DENSE_PQ_OFF
    constexpr bool PQ = false;
#else
    constexpr bool PQ = true;
#endif
    long long t_a = 0, t_b = 0, t_c = 0, t_d = 0, t_e = 0, n_pop = 0, n_merge = 0;   // GS_TRAV_PROFILE: cycle stamps of workgroup 0
    long long tq1 = 0, tq2 = 0, tq3 = 0, tq4 = 0, tna = 0;
    constexpr uint32_t CN = ONEG ? 512u : (uint32_t)DCN;         // capacity of N: >= 2M (an empty N takes a whole expansion), one key per lane in its merge
    DenseLds S = carve_dense(s_raw, ix.m, knbn, maxdeg, SPLIT ? (uint64_t)vis_w : ix.n, VLDS, CN);
    // adjacency row of a (wave-uniform) candidate: the node id is pinned to an SGPR so that the row base is scalar and the load takes a
    // 32-bit lane offset instead of a 64-bit per-lane pointer
#define GS_TRAV_LOAD(p) __builtin_nontemporal_load(p)
#define GS_DROWX(K, PD, PI)                                                                                                \
    do {                                                                                                                   \
        const uint32_t rid_ = uni32(KID(K));                                                                               \
        uint32_t ho_ = hl4;                                                                                                \
        asm volatile("" : "+v"(ho_));               /* opaque lane offset: keeps the 64-bit row base scalar */            \
        PD = ix.deg0[rid_];                                                                                                \
        PI = hl < maxdeg ? GS_TRAV_LOAD((const uint32_t *)((const uint8_t *)(ix.nbr0 + (uint64_t)rid_ * maxdeg) + ho_)) : 0; \
    } while (0)
#define GS_DROW(K) GS_DROWX(K, pdeg, pid)
    // per-workgroup global scratch: the visited bitmap (VLDS = false) or the fine histogram bins (VLDS = true)
    uint32_t *vis = VLDS ? S.vis : scratch + (uint64_t)blockIdx.x * scratch_words;
    Hist3 hs; hs.Hf = VLDS ? scratch + (uint64_t)blockIdx.x * scratch_words : S.Hf; hs.H2 = S.H2; hs.H1 = S.H1;
    uint32_t *visg = SPLIT ? scratch + (uint64_t)blockIdx.x * scratch_words + hwords : nullptr;
    uint64_t *Cb[2] = {cbuf + (uint64_t)blockIdx.x * 2 * capC, cbuf + (uint64_t)blockIdx.x * 2 * capC + capC};
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t half = ONEG ? 0u : threadIdx.x >> 8, hl = ONEG ? threadIdx.x : threadIdx.x & 255, hl4 = hl * 4;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.scal[1] = atomicAdd(counter, 1ull);
        __syncthreads();
        const uint64_t qi = S.scal[1];
        if (qi >= nq) break;
        uint64_t *wl = WLOG ? wlog + (uint64_t)blockIdx.x * cap_log : nullptr;
        uint32_t nlog = 1;
        const uint16_t *matrow = mat + qi * mat_ld;
        const bool tm = PHASE2 && stats && blockIdx.x == 0 && threadIdx.x == 0;
        const long long tm0 = tm ? clock64() : 0;
        const uint32_t tau = PHASE2 ? dense_row_tau(matrow, ix.n, ix.m, efs, S.H1, S.hist, S.wsum) : 0u;
        const long long tm1 = tm ? clock64() : 0;
        for (uint32_t w = threadIdx.x; w < vis_words; w += DT) vis[w] = 0;
        if (SPLIT) for (uint32_t w = threadIdx.x; w < visg_words; w += DT) visg[w] = 0;
        for (uint32_t w = threadIdx.x; w < hwords; w += DT) hs.Hf[w] = 0;
        for (uint32_t w = threadIdx.x; w < nb * (HB / HG / 2); w += DT) S.H2[w] = 0;
        for (uint32_t w = threadIdx.x; w < nb; w += DT) S.H1[w] = 0;
        uint32_t evals = 1;                                      // <= n + upper-layer hops
        uint32_t ep = (uint32_t)ix.entry, ep_cnt = matrow[ep];
        // WLOG: the caller may hand the layer-0 entry point over (a point of level > 0 enters layer 0 where its layer-1 search ended);
        // its evaluations above layer 0 are the caller's to count
        const bool given = WLOG && ep_in && ep_in[qi] != 0xFFFFFFFFu;
        if (given) { ep = ep_in[qi]; ep_cnt = matrow[ep]; evals = 0; }
        // greedy descent on the upper layers (hnsw_rs::search outer loop)
        for (int L = given ? 0 : ix.top; L >= 1; L--) {
            for (;;) {
                const uint32_t *nbr; uint32_t deg;
                node_neighbours(ix, ep, L, nbr, deg);
                if (deg == 0) break;
                if (threadIdx.x == 0) S.scal[0] = ~(uint64_t)0;
                __syncthreads();
                evals += deg;
                if (threadIdx.x < deg) { const uint32_t c = matrow[nbr[threadIdx.x]]; if (c < ep_cnt) atomicMin((unsigned long long *)&S.scal[0], (unsigned long long)KEY(c, threadIdx.x)); }
                __syncthreads();
                const uint64_t best = S.scal[0];
                __syncthreads();
                if (best == ~(uint64_t)0) break;
                ep = nbr[KID(best)]; ep_cnt = KCNT(best);
            }
        }
        // ---- search_layer on layer 0. The next candidate is min(head of G, head of N) under the same (count,id) order, so the
        //      pop sequence is that of search_layer_block.
        int cur = 0;
        uint32_t nR = 1, nT = 1, nG = 0, headG = 0, wbase = 0, wn = 0, nN = 1, headN = 0;
        uint32_t dmax = INF_CNT, tieT = 0;                       // worst count of a full R / #keys of R tied at it
        uint64_t Tmax = 0;                                       // T[knbn-1] once T is full
        // PQ: this lane's key of the front (wavefront 0 only; ~0 = empty), the fence, the sizes of front and overflow, where the last compaction left U
        uint64_t pkey = ~(uint64_t)0, pv = ~(uint64_t)0;
        uint32_t nP = 1, nU = 0, nU_lc = 0;
        __syncthreads();
        if (threadIdx.x < 3) { S.W[threadIdx.x] = ~(uint64_t)0; S.N[1 + threadIdx.x] = ~(uint64_t)0; }
        if (PQ && threadIdx.x < 8) S.W[threadIdx.x] = threadIdx.x == 0 ? KEY(ep_cnt, ep) : ~(uint64_t)0;
        if (PQ && threadIdx.x == 0) pkey = KEY(ep_cnt, ep);
        if (threadIdx.x == 0) {
            S.T[0] = KEY(ep_cnt, ep); S.N[0] = KEY(ep_cnt, ep);
            if (WLOG) wl[0] = KEY(ep_cnt, ep);
            hist_add<VLDS>(hs, ep_cnt, 1);
            if (SPLIT && ep >= vis_w) __hip_atomic_fetch_or(&visg[(ep - vis_w) >> 5], 1u << (ep & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (VLDS) vis[ep >> 5] = 1u << (ep & 31);
            else __hip_atomic_fetch_or(&vis[ep >> 5], 1u << (ep & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (efs == 1) { dmax = ep_cnt; tieT = 1; }
        if (knbn == 1) Tmax = KEY(ep_cnt, ep);
        __syncthreads();
        // Two-stage software pipeline over the pops. The two 256-lane halves of the workgroup alternate roles: on pop i half
        // (i & 1) expands the candidate while the other half prefetches for pop i+1 - its adjacency and (VLDS) a plain read of
        // the visited words as a hint plus, where the hint says "unvisited", the count lookup; the expanding half then loads the
        // adjacency of the candidate after next. Prefetched data is keyed by the candidate it belongs to and is only a hint
        // (adjacency and counts are constant during a search, visited bits only ever get set), so a wrong prediction costs a
        // direct load, never a wrong answer.
        uint64_t pk = ~(uint64_t)0;                              // candidate this half holds data for
        uint32_t pid = 0, pdeg = 0, pcnt = 0, pst = 0, it = 0;   // pst: 1 = adjacency, 2 = + visited hint and lookups
        bool pclr = false;
        uint64_t nk = ~(uint64_t)0;                              // ONEG: row fetched for the candidate after next (no hint, no lookups yet)
        uint32_t nid = 0, ndeg = 0;
        bool phase2 = false;
        for (;;) {
            if (PHASE2 && dmax == tau && !(cap_log & 1u)) { phase2 = true; break; }       // R holds efs keys <= tau (or n < efs): the rest is order-free
            uint64_t c, c1, c2;
            const bool full = nR == efs;
            const uint32_t fb = (it & 1u) * 4u;                      // PQ: the front's first three keys are published twice over, by pop parity
            if (PQ) {
                c = S.W[fb]; c1 = S.W[fb + 1]; c2 = S.W[fb + 2];
                // the front is empty: refill it from U; U has outgrown its buffer: drop its dead keys (counts above dmax)
                const bool empty = c == ~(uint64_t)0, crowded = nU + maxdeg > capC && nU - nU_lc >= maxdeg;
                if (empty && nU == 0) break;
                if (empty || crowded) {
                    pq_rebuild(ix, S, Cb[cur], Cb[cur ^ 1], capC, nU, empty, full ? dmax : INF_CNT, pkey, pv, nP, fb, st_pq);
                    cur ^= 1; nU = (uint32_t)S.scal[4]; nU_lc = nU;
                    continue;
                }
                if (KCNT(c) > dmax) break;                           // dmax is INF_CNT until R is full; the front's head is the smallest waiting key
                if (wv == 0) {                                       // pop: the front moves down one lane; what the next pop sees if this one accepts nothing
                    pkey = wave_shl1_u64(pkey); if (lane == 63) pkey = ~(uint64_t)0;
                    nP--;
                    if (lane < 3) S.W[(fb ^ 4u) + lane] = pkey;
                }
            } else {
            if (headG < nG && headG - wbase >= wn) {                 // refill the LDS window of G
                __syncthreads();
                wbase = headG; wn = nG - headG < (uint32_t)DWIN ? nG - headG : (uint32_t)DWIN;
                if (threadIdx.x < wn + 3) S.W[threadIdx.x] = threadIdx.x < wn ? Cb[cur][headG + threadIdx.x] : ~(uint64_t)0;
                __syncthreads();
            }
            // heads of G (through its LDS window; beyond the window counts as unknown) and of N, read in one round: the candidate to
            // pop and the two that follow in the current order
            const uint32_t wo = headG - wbase;
            // (both arrays end in three ~0 sentinels; behind a pruned N there may be dead keys instead - counts above dmax, which stop
            // the search exactly like "no candidate" does)
            const uint64_t g0 = S.W[wo], g1 = S.W[wo + 1], g2 = S.W[wo + 2];
            const uint64_t n0 = S.N[headN], n1 = S.N[headN + 1], n2 = S.N[headN + 2];
            c = g0 < n0 ? g0 : n0;
            if (c == ~(uint64_t)0) break;
            if (KCNT(c) > dmax) break;                               // dmax is INF_CNT until R is full
            if (g0 < n0) { headG++; c1 = g1 < n0 ? g1 : n0; c2 = g1 < n0 ? (g2 < n0 ? g2 : n0) : (g1 < n1 ? g1 : n1); }
            else { headN++; c1 = g0 < n1 ? g0 : n1; c2 = g0 < n1 ? (g1 < n1 ? g1 : n1) : (g0 < n2 ? g0 : n2); }
            }
            const long long p0 = PROF ? clock64() : 0;
            uint32_t id = 0, cntv = 0; bool unv = false;
            if (ONEG || half == (it & 1)) {
                if (pk != c) {
                    if (ONEG && nk == c) { pid = nid; pdeg = ndeg; nk = ~(uint64_t)0; }          // its row is here, the hint stage has not seen it
                    else GS_DROW(c);
                    pst = 1;
                }
                id = pid;
                if (hl < pdeg) {
                    const uint32_t bit = 1u << (id & 31);
                    uint32_t old;
                    if (SPLIT && id >= vis_w) {
                        // a hint that saw the bit SET is never stale (bits only get set): more than half of the neighbours of a flood-regime pop are
                        // visited already, and each of them would otherwise be a memory-side atomic - phase 1 of a large index is bound by those
                        if (pst == 2 && !pclr) old = bit;
                        else old = __hip_atomic_fetch_or(&visg[(id - vis_w) >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    else if (VLDS) old = atomicOr(&vis[id >> 5], bit);
                    else old = __hip_atomic_fetch_or(&vis[id >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    unv = !(old & bit);
                    if (unv) cntv = (pst == 2 && pclr) ? pcnt : (uint32_t)matrow[id];   // every 2-byte lookup costs a full HBM sector: only for the unvisited
                }
            } else if (!ONEG && c1 != ~(uint64_t)0) {
                if (pk != c1) {
                    pk = c1; pst = 1;
                    GS_DROW(c1);
                }
                if (VLDS) {
                    pclr = false;
                    if (hl < pdeg) {
                        if (SPLIT && pid >= vis_w) pclr = !((__hip_atomic_load(&visg[(pid - vis_w) >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (pid & 31)) & 1u);
                        else pclr = !((vis[pid >> 5] >> (pid & 31)) & 1u);
                        if (pclr) pcnt = matrow[pid];
                    }
                    pst = 2;
                }
            } else if (!ONEG) pk = ~(uint64_t)0;
            it++;
            // one barrier per pop: every wave publishes how many of its lanes hit an unvisited node and how many of those are
            // below the worst count of a full R (dmax is INF_CNT until R is full, so "below" = "unvisited" then)
            const bool below = unv && cntv < dmax;
            const uint64_t bal = __ballot(unv), balb = __ballot(below);
            const long long p1 = PROF ? clock64() : 0;
            uint32_t *ws = S.wsum + ((it & 1) ? 8 : 0);              // double-buffered: the next pop may start before every wave has read
            if (lane == 0) ws[wv] = (uint32_t)__popcll(bal) | ((uint32_t)__popcll(balb) << 16);
            // the expanding half now fetches the adjacency of the candidate after next - issued only here, behind the wait for its own
            // lookups, so that wait does not include these loads
            if (!ONEG) {
                if (half != (it & 1)) {
                    pk = c2; pst = 1;
                    if (c2 != ~(uint64_t)0) GS_DROW(c2);
                }
            } else {
                pk = c1;
                if (c1 != ~(uint64_t)0) {
                    if (nk == c1) {
                        pid = nid; pdeg = ndeg; pclr = false;
                        if (VLDS) {
                            if (hl < pdeg) {
                                if (SPLIT && pid >= vis_w) pclr = !((__hip_atomic_load(&visg[(pid - vis_w) >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (pid & 31)) & 1u);
                                else pclr = !((vis[pid >> 5] >> (pid & 31)) & 1u);
                                if (pclr) pcnt = matrow[pid];
                            }
                            pst = 2;
                        } else pst = 1;
                    } else { GS_DROW(c1); pst = 1; }                 // not predicted: the row only (its lookups go direct at the expansion)
                }
                nk = c2;
                if (c2 != ~(uint64_t)0) GS_DROWX(c2, ndeg, nid);
            }
            lds_barrier();
            const long long p2 = PROF ? clock64() : 0;
            // ne <= 2M < 2^16, so the packed words add without carrying into each other; the per-wave prefixes are only needed by an
            // accepting pop and are computed there
            uint32_t tot = 0;
#pragma unroll
            for (int w = 0; w < DT / 64; w++) tot += ws[w];
            const uint32_t ne = tot & 0xFFFFu, B = tot >> 16;
            const long long p3 = PROF ? clock64() : 0;
            if (PROF && blockIdx.x == 0 && threadIdx.x == 0) { t_a += p1 - p0; t_b += p2 - p1; t_c += p3 - p2; n_pop++; }
            st_pops++;
            if (PHASE2 && dmax != tau) st_p1++;
            if (ne == 0) continue;
            evals += ne;
            // closed-form accept rule: e_i is accepted iff #{x in R: c(x) <= c_i} + #{j<i: c_j <= c_i} < ef (and c_i < dmax once R is
            // full). Shortcuts that need no ranks (and no compaction of the expansion): R not full and nR + ne <= ef -> everything
            // passes; R full with T keys tied at dmax and B <= T candidates below dmax -> the first term is <= ef - T, all B pass.
            bool slow;
            if (!full) slow = nR + ne > efs;
            else { if (B == 0) continue; slow = B > tieT; }
            uint64_t mykey = ~(uint64_t)0, ab; bool acc = false;
            uint32_t na, ci, aoff;
            uint32_t pre = 0;                                        // packed (unvisited | below << 16) counts of the waves before this one
#pragma unroll
            for (int w = 0; w < DT / 64; w++) if (w < (int)wv) pre += ws[w];
            const uint32_t off = pre & 0xFFFFu, boff = pre >> 16;
            if (!slow) { acc = below; na = B; ci = cntv; mykey = KEY(cntv, id); ab = balb; aoff = boff; }
            else {
                // the expansion E in adjacency order
                if (unv) { const uint32_t pos = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1)); S.Eid[pos] = id; S.Ecnt[pos] = cntv; }
                // exclusive prefix of the block sums, then rank = P1[block] + groups of the block + bins of the group up to c_i
                const uint32_t CH = (nb + DT - 1) / DT;
                uint32_t loc = 0;
                for (uint32_t k2 = 0; k2 < CH; k2++) { const uint32_t idx = threadIdx.x * CH + k2; if (idx < nb) loc += S.H1[idx]; }
                uint32_t inc = loc;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
                if (lane == 63) S.wsum[16 + wv] = inc;
                __syncthreads();                              // full: the histogram atomics sent to global memory have landed
                ci = threadIdx.x < ne ? S.Ecnt[threadIdx.x] : INF_CNT;
                uint32_t woff = 0;
#pragma unroll
                for (int w = 0; w < DT / 64; w++) if (w < (int)wv) woff += S.wsum[16 + w];
                uint32_t run = woff + inc - loc;
                for (uint32_t k2 = 0; k2 < CH; k2++) { const uint32_t idx = threadIdx.x * CH + k2; if (idx < nb) { S.P1[idx] = run; run += S.H1[idx]; } }
                lds_barrier();
                if (threadIdx.x < ne && ci < dmax) {
                    uint32_t le = S.P1[ci / HB];
                    for (uint32_t g = (ci / HB) * (HB / HG); g < ci / HG; g++) le += h16(S.H2, g);
                    const uint32_t w0 = (ci / HG) * (HG / 2), w1 = ci >> 1;
                    for (uint32_t w = w0; w <= w1; w++) {
                        const uint32_t x = hf_word<VLDS>(hs.Hf, w);
                        le += x & 0xFFFFu;
                        if (w < w1 || (ci & 1)) le += x >> 16;
                    }
                    if (le < efs) {
#pragma unroll 8
                        for (uint32_t j = 0; j < threadIdx.x; j++) le += (S.Ecnt[j] <= ci);
                    }
                    acc = le < efs;
                }
                if (acc) mykey = KEY(ci, S.Eid[threadIdx.x]);
                ab = __ballot(acc);
                if (lane == 0) S.wsum[24 + wv] = (uint32_t)__popcll(ab);
                lds_barrier();                                    // also: every read of Eid/Ecnt is done, As may overwrite them
                aoff = 0; na = 0;
#pragma unroll
                for (int w = 0; w < DT / 64; w++) { const uint32_t x = S.wsum[24 + w]; if (w < (int)wv) aoff += x; na += x; }
            }
            const long long p4 = PROF ? clock64() : 0;
            if (PROF && blockIdx.x == 0 && threadIdx.x == 0) t_d += p4 - p3;
            if (na == 0) continue;
            n_merge++; st_acc++;
            // accepted keys: histogram update, compaction (ballot prefix) into As, then rank-sort the na (usually 1-5) keys into A
            uint64_t minA = 0;
            const uint32_t fbn = (it & 1u) * 4u;                     // (it has moved on: the slot the next pop reads)
            if (PQ) {
                if (acc) { hist_add<VLDS>(hs, ci, 1); S.As[aoff + (uint32_t)__popcll(ab & ((1ull << lane) - 1))] = mykey; }
                lds_barrier();
                if (wv == 0) {
                    // wavefront 0 takes the accepted keys 64 at a time: beyond the fence -> appended to U as they are (one coalesced store); below
                    // it -> into the sorted front, one after the other (position = ballot of "my key is smaller", everything behind moves up one
                    // lane, a full front hands its largest key to U and the fence comes down to the new largest)
                    uint64_t tany = 0;                                 // some accepted key lies below T's largest: T has to take this pop's keys in
                    uint64_t *U = Cb[cur];
                    uint32_t nu = nU;
                    for (uint32_t b0 = 0; b0 < na; b0 += 64) {
                        const uint64_t k = b0 + lane < na ? S.As[b0 + lane] : ~(uint64_t)0;
                        tany |= __ballot(k < Tmax);
                        const bool toU = k != ~(uint64_t)0 && k >= pv;
                        const uint64_t bu = __ballot(toU);
                        if (toU) { const uint32_t pos = nu + (uint32_t)__popcll(bu & ((1ull << lane) - 1)); if (pos < capC) U[pos] = k; }
                        nu += (uint32_t)__popcll(bu);
                        uint64_t pend = __ballot(k != ~(uint64_t)0 && k < pv);
                        while (pend) {
                            const int l = __ffsll((long long)pend) - 1;
                            pend &= pend - 1;
                            const uint64_t kk = readlane_u64(k, l);
                            if (kk >= pv) { if (lane == 0 && nu < capC) U[nu] = kk; nu++; continue; }      // the fence has come down since
                            const uint32_t pos = (uint32_t)__popcll(__ballot(pkey < kk));
                            const uint64_t ev = readlane_u64(pkey, 63), up = wave_shr1_u64(pkey);
                            pkey = lane < pos ? pkey : (lane == pos ? kk : up);
                            if (ev != ~(uint64_t)0) { if (lane == 0 && nu < capC) U[nu] = ev; nu++; }
                            else nP++;
                            if (nP == 64) pv = readlane_u64(pkey, 63);
                        }
                    }
                    if (lane < 3) S.W[fbn + lane] = pkey;
                    if (lane == 0) { S.scal[4] = nu; S.scal[5] = tany ? 0 : ~(uint64_t)0; }
                }
                lds_barrier();
                nU = (uint32_t)S.scal[4]; minA = S.scal[5];
            } else {
                if (acc) { hist_add<VLDS>(hs, ci, 1); S.As[aoff + (uint32_t)__popcll(ab & ((1ull << lane) - 1))] = mykey; }
                lds_barrier();
                if (threadIdx.x < na) {
                    const uint64_t k = S.As[threadIdx.x];
                    uint32_t rank = 0;
#pragma unroll 8
                    for (uint32_t j = 0; j < na; j++) rank += (S.As[j] < k);
                    S.A[rank] = k;
                }
                lds_barrier();
            }
            if (WLOG) { if (nlog + na <= cap_log && threadIdx.x < na) wl[nlog + threadIdx.x] = PQ ? S.As[threadIdx.x] : S.A[threadIdx.x]; nlog += na; }
            {   // the candidate order has changed: start loading the adjacency of the two candidates that now come first
                // (first two of merge(A, {c1, c2}); PQ: the front as wavefront 0 has just published it) so that the loads overlap with the rest of the merge
                const uint64_t a0 = PQ ? 0 : S.A[0], a1 = PQ ? 0 : (na > 1 ? S.A[1] : ~(uint64_t)0);
                const uint64_t n1 = PQ ? S.W[fbn] : (a0 < c1 ? a0 : c1);
                const uint64_t n2 = PQ ? S.W[fbn + 1] : (a0 < c1 ? (a1 < c1 ? a1 : c1) : (a0 < c2 ? a0 : c2));
                if (!ONEG) {
                    const uint64_t want = half == (it & 1) ? n1 : n2;
                    if (want != ~(uint64_t)0 && pk != want) {
                        pk = want; pst = 1;
                        GS_DROW(want);
                    }
                } else {
                    if (n1 != ~(uint64_t)0 && pk != n1) {
                        const uint64_t ok = pk; const uint32_t oid = pid, odeg = pdeg;      // the row held so far: usually the new second
                        if (nk == n1) { pid = nid; pdeg = ndeg; }
                        else GS_DROW(n1);
                        pk = n1; pst = 1;
                        nk = ok; nid = oid; ndeg = odeg;
                    }
                    if (n2 != ~(uint64_t)0 && nk != n2) { nk = n2; GS_DROWX(n2, ndeg, nid); }
                }
            }
            const long long q1 = PROF ? clock64() : 0;
            const uint32_t dold = dmax;
            // R <- ef smallest of R u A: drop the (nR + na - ef) largest counts from the top bins
            if (nR + na >= efs) {
                const uint32_t excess = nR + na - efs;
                if (full && excess < tieT) {
                    if (threadIdx.x == 0 && excess) hist_sub<VLDS>(hs, dmax, excess);
                    tieT -= excess;
                } else {
                    __syncthreads();                              // full: wave 0 reads bins other waves have just incremented
                    if (threadIdx.x < 64) {
                        uint32_t d = full ? dmax : ix.m, ex = excess, tt = 0;
                        for (;;) {
                            d = hist_find_down<VLDS>(hs, d, lane, tt);
                            if (ex == 0 || tt == 0) break;
                            const uint32_t r = ex < tt ? ex : tt;
                            if (lane == 0) hist_sub<VLDS>(hs, d, r);
                            __threadfence_block();
                            ex -= r;
                            if (r < tt) { tt -= r; break; }
                            if (d == 0) { tt = 0; break; }
                            d -= 1;
                        }
                        if (lane == 0) { S.scal[2] = d; S.scal[3] = tt; }
                    }
                    lds_barrier();
                    dmax = (uint32_t)S.scal[2]; tieT = (uint32_t)S.scal[3];
                }
                nR = efs;
            } else nR += na;
            const uint32_t dnew = dmax;                              // INF_CNT while R is not full
            const long long q2 = PROF ? clock64() : 0;
            if (PQ) {
                // T <- knbn smallest of T u A, only when A reaches into it (1-2 % of the pops): that merge wants A sorted
                if (nT < knbn || minA < Tmax) {
                    if (threadIdx.x < na) {
                        const uint64_t k = S.As[threadIdx.x];
                        uint32_t rank = 0;
#pragma unroll 8
                        for (uint32_t j = 0; j < na; j++) rank += (S.As[j] < k);
                        S.A[rank] = k;
                    }
                    lds_barrier();
                    const SmallA sa = load_small_a(S.A, na);
                    nT = dense_merge_T(S.T, nT, S.A, na, knbn, sa);
                    if (nT == knbn) Tmax = S.T[knbn - 1];
                }
                if (PROF && blockIdx.x == 0 && threadIdx.x == 0) { const long long q4 = clock64(); t_e += q4 - p4; tq1 += q1 - p4; tq2 += q2 - q1; tq4 += q4 - q2; tna += na; }
                continue;
            }
            const SmallA sa = load_small_a(S.A, na);
            // T <- knbn smallest of T u A (only when A reaches into it)
            if (nT < knbn || S.A[0] < Tmax) {
                if (PROF && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&prof[13], 1ull);
                nT = dense_merge_T(S.T, nT, S.A, na, knbn, sa);
                if (nT == knbn) Tmax = S.T[knbn - 1];
            }
            // ---- N full? fold its live part into G first (rare): G' = live G u live N, dead tail dropped
            if (PROF && blockIdx.x == 0 && threadIdx.x == 0 && nN - headN + na > CN) atomicAdd(&prof[14], 1ull);
            if (nN - headN + na > CN) {
                const uint32_t liveN = nN - headN, liveG = nG - headG;
                const uint64_t *NL = S.N + headN;
                for (uint32_t t = threadIdx.x; t <= liveN; t += DT) S.hist[t] = 0;
                __syncthreads();
                uint64_t *src = Cb[cur] + headG, *dst = Cb[cur ^ 1];
                uint32_t alive_loc = 0;
                // (four keys of G in flight per lane: G lives in global memory and a lane's keys were fetched one round trip after the other -
                // ~10 per fold at ef = 5000, the fold's whole cost)
                for (uint32_t idx0 = 0; idx0 < liveG; idx0 += 4 * DT) {
                    uint64_t kf[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const uint32_t idx = idx0 + u * DT + threadIdx.x; kf[u] = idx < liveG ? src[idx] : ~(uint64_t)0; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t idx = idx0 + u * DT + threadIdx.x;
                        if (idx < liveG) {
                            const uint64_t k = kf[u];
                            const uint32_t lb = lower_bound_keys(NL, liveN, k);
                            if (idx + lb < capC) dst[idx + lb] = k;
                            atomicAdd(&S.hist[lb], 1u);
                            alive_loc += (KCNT(k) <= dnew);
                        }
                    }
                }
                if (threadIdx.x < liveN) alive_loc += (KCNT(NL[threadIdx.x]) <= dnew);
                __syncthreads();
                uint32_t wsumv = alive_loc;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) wsumv += __shfl_down(wsumv, o);
                if (lane == 0) S.wsum[wv] = wsumv;
                // N key t lands behind the G keys that sort before it: inclusive prefix of hist[0..t] (block scan)
                uint32_t inc = threadIdx.x <= liveN ? S.hist[threadIdx.x] : 0;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
                if (lane == 63) S.wsum[40 + wv] = inc;
                __syncthreads();
                if (threadIdx.x < liveN) {
                    uint32_t below = inc;
#pragma unroll
                    for (int w = 0; w < DT / 64; w++) if (w < (int)wv) below += S.wsum[40 + w];
                    if (threadIdx.x + below < capC) dst[threadIdx.x + below] = NL[threadIdx.x];
                }
                __syncthreads();
                uint32_t alive = 0;
#pragma unroll
                for (int w = 0; w < DT / 64; w++) alive += S.wsum[w];
                uint32_t tot = liveG + liveN; if (tot > capC) tot = capC;
                nG = (nR == efs && alive < tot) ? alive : tot;
                cur ^= 1; headG = 0; wbase = 0; wn = 0; nN = 0; headN = 0;
                if (threadIdx.x < 3) S.W[threadIdx.x] = ~(uint64_t)0;      // read as heads if G' is empty (no refill then); ordered by the barriers of the N merge below
            }
            const long long q3 = PROF ? clock64() : 0;
            // ---- N <- live N u A (both tiny, in LDS), dead tail dropped
            {
                const uint32_t liveN = nN - headN;
                uint64_t nk = 0; uint32_t npos = 0xFFFFFFFFu;
                if (threadIdx.x < liveN) { nk = S.N[headN + threadIdx.x]; npos = threadIdx.x + lb_a(sa, nk); }
                uint64_t ak = 0; uint32_t apos = 0xFFFFFFFFu;
                if (threadIdx.x < na) { ak = S.A[threadIdx.x]; apos = threadIdx.x + lower_bound_keys(S.N + headN, liveN, ak); }
                lds_barrier();
                if (npos != 0xFFFFFFFFu) S.N[npos] = nk;
                if (apos != 0xFFFFFFFFu) S.N[apos] = ak;
                if (threadIdx.x >= DT - 3) S.N[liveN + na + (threadIdx.x - (DT - 3))] = ~(uint64_t)0;
                lds_barrier();
                nN = liveN + na; headN = 0;
                if (nR == efs && dnew != dold) { const uint32_t alive = lower_bound_keys(S.N, nN, KEY(dnew, 0xFFFFFFFFu)); if (alive < nN) nN = alive; }
            }
            if (PROF && blockIdx.x == 0 && threadIdx.x == 0) { const long long q4 = clock64(); t_e += q4 - p4; tq1 += q1 - p4; tq2 += q2 - q1; tq3 += q3 - q2; tq4 += q4 - q3; tna += na; }
        }
        __syncthreads();
        const long long tm2 = tm ? clock64() : 0;
        if (PHASE2 && phase2) {
            Phase2IO io{nT, evals, 0u, tm, 0, 0, wl, cap_log, nlog};
            if (PQ) {                                                // the waiting candidates as phase 2 reads them: "G" = U (it only filters), "N" = the front
                if (wv == 0) { S.N[lane] = pkey; if (lane == 0) S.scal[6] = nP; }
                __syncthreads();
                headG = 0; nG = nU < capC ? nU : capC; headN = 0; nN = (uint32_t)S.scal[6];
            }
            dense_phase2<ONEG, WLOG, SPLIT>(ix, S, vis, matrow, tau, knbn, Cb[cur], Cb[cur ^ 1], headG, nG, headN, nN, Tmax, io, vis_w, visg);
            nT = io.nT; evals = io.evals; st_pops += io.pops; st_p2 += io.pops; nlog = io.nlog;
            if (tm) { atomicAdd(&stats[13], (unsigned long long)(io.c_build - tm2)); atomicAdd(&stats[14], (unsigned long long)(io.c_drain - io.c_build)); }
        }
        const long long tm3 = tm ? clock64() : 0;
        if (ids_out) for (uint32_t i = threadIdx.x; i < knbn; i += DT) {
            if (i < nT) { ids_out[qi * knbn + i] = KID(S.T[i]); dist_out[qi * knbn + i] = (float)KCNT(S.T[i]) / (float)ix.m; }
            else { ids_out[qi * knbn + i] = ~(uint64_t)0; dist_out[qi * knbn + i] = INFINITY; }
        }
        if (threadIdx.x == 0) { if (count_out) count_out[qi] = nT; if (evals_out) evals_out[qi] = evals; }
        if (tm) {
            atomicAdd(&stats[8], (unsigned long long)(tm1 - tm0)); atomicAdd(&stats[9], (unsigned long long)(tm2 - tm1)); atomicAdd(&stats[10], (unsigned long long)(tm3 - tm2));
            atomicAdd(&stats[11], (unsigned long long)(clock64() - tm3)); atomicAdd(&stats[12], 1ull);
        }
        if (WLOG) {
            // W = the efs smallest keys of the log: cut it at dmax (INF_CNT while R never filled), sort what is left in LDS (the search
            // arrays are dead), keep the first efs. Anything that does not fit is flagged and redone by the caller the slow way.
            uint64_t *buf = (uint64_t *)s_raw + 8;
            uint32_t *cntp = (uint32_t *)s_raw;
            if (threadIdx.x == 0) cntp[0] = 0;
            __syncthreads();
            bool bad = nlog > cap_log;
            if (!bad) for (uint32_t i = threadIdx.x; i < nlog; i += DT) {
                const uint64_t k = wl[i];
                if (KCNT(k) <= dmax) { const uint32_t pos = atomicAdd(&cntp[0], 1u); if (pos < sort_cap) buf[pos] = k; }
            }
            __syncthreads();
            const uint32_t cnt = cntp[0];
            if (cnt > sort_cap) bad = true;
            if (bad) { if (threadIdx.x == 0) w_n[qi] = 0xFFFFFFFFu; }
            else {
                uint32_t n2 = 2; while (n2 < cnt) n2 <<= 1;
                for (uint32_t i = cnt + threadIdx.x; i < n2; i += DT) buf[i] = ~(uint64_t)0;
                __syncthreads();
                for (uint32_t kk = 2; kk <= n2; kk <<= 1)
                    for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                        for (uint32_t i = threadIdx.x; i < n2; i += DT) {
                            const uint32_t l = i ^ j;
                            if (l > i) {
                                const uint64_t a = buf[i], b = buf[l];
                                if ((a > b) == ((i & kk) == 0)) { buf[i] = b; buf[l] = a; }
                            }
                        }
                        __syncthreads();
                    }
                const uint32_t nW = cnt < efs ? cnt : efs;
                for (uint32_t i = threadIdx.x; i < nW; i += DT) w_out[qi * efs + i] = buf[i];
                if (threadIdx.x == 0) w_n[qi] = nW;
            }
        }
    }
#undef GS_DROW
#undef GS_DROWX
    if (stats && threadIdx.x == 0) { atomicAdd(&stats[1], (unsigned long long)st_pops); atomicAdd(&stats[2], (unsigned long long)st_acc); atomicAdd(&stats[5], (unsigned long long)st_p1); atomicAdd(&stats[6], (unsigned long long)st_p2); if (st_pq) atomicAdd(&stats[15], st_pq); }
    if (PROF && blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&prof[0], (unsigned long long)t_a); atomicAdd(&prof[1], (unsigned long long)t_b); atomicAdd(&prof[2], (unsigned long long)t_c);
        atomicAdd(&prof[3], (unsigned long long)t_d); atomicAdd(&prof[4], (unsigned long long)t_e); atomicAdd(&prof[5], (unsigned long long)n_pop); atomicAdd(&prof[6], (unsigned long long)n_merge);
        atomicAdd(&prof[8], (unsigned long long)tq1); atomicAdd(&prof[9], (unsigned long long)tq2); atomicAdd(&prof[10], (unsigned long long)tq3); atomicAdd(&prof[11], (unsigned long long)tq4); atomicAdd(&prof[12], (unsigned long long)tna);
    }
}


__device__ __forceinline__ SearchLds carve_lds(uint8_t *base, uint32_t ef, uint32_t maxdeg)
{
    SearchLds S;
    const size_t capC = 2 * (size_t)ef + maxdeg + 64;
    S.R = (uint64_t *)base; base += 8 * (size_t)ef;
    S.C = (uint64_t *)base; base += 8 * capC;
    S.A = (uint64_t *)base; base += 8 * (size_t)maxdeg;
    S.As = (uint64_t *)base; base += 8 * (size_t)maxdeg;
    S.scal = (uint64_t *)base; base += 8 * 8;
    S.Eid = (uint32_t *)base; base += 4 * (size_t)maxdeg;
    S.Ecnt = (uint32_t *)base; base += 4 * (size_t)maxdeg;
    S.wsum = (uint32_t *)base;
    return S;
}

template <int KIND>
__global__ __launch_bounds__(ST) void k_hnsw_search(IndexDev ix, const uint8_t *__restrict__ queries, uint64_t nq, uint32_t knbn, uint32_t ef,
                                                     const uint16_t *__restrict__ mat, uint64_t mat_ld, uint32_t *__restrict__ visited, uint32_t vis_words, unsigned long long *__restrict__ counter,
                                                     uint64_t *__restrict__ ids_out, float *__restrict__ dist_out, uint32_t *__restrict__ count_out,
                                                     uint64_t *__restrict__ evals_out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    const uint32_t maxdeg = 2 * ix.M;
    const uint32_t efs = ef > knbn ? ef : knbn;
    SearchLds S = carve_lds(s_raw, efs, maxdeg);
    uint32_t *vis = visited + (uint64_t)blockIdx.x * vis_words;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.scal[1] = atomicAdd(counter, 1ull);
        __syncthreads();
        const uint64_t qi = S.scal[1];
        if (qi >= nq) break;
        const uint4 *q = (const uint4 *)(queries + qi * ix.stride);
        const uint16_t *matrow = mat ? mat + qi * mat_ld : nullptr;
        for (uint32_t w = threadIdx.x; w < vis_words; w += ST) vis[w] = 0;
        uint64_t evals = 1;
        // distance to the entry point
        if (threadIdx.x == 0) S.Eid[0] = (uint32_t)ix.entry;
        __syncthreads();
        block_distances<KIND>(ix, q, S.Eid, 1, S.Ecnt, matrow);
        uint32_t ep = (uint32_t)ix.entry, ep_cnt = S.Ecnt[0];
        __syncthreads();
        for (int L = ix.top; L >= 1; L--) greedy_layer_block<KIND>(ix, q, S, ep, ep_cnt, L, evals, matrow);
        const uint32_t nR = search_layer_block<KIND>(ix, q, S, vis, ep, ep_cnt, efs, 0, evals, matrow);
        const uint32_t nout = nR < knbn ? nR : knbn;
        for (uint32_t i = threadIdx.x; i < knbn; i += ST) {
            if (i < nout) { ids_out[qi * knbn + i] = KID(S.R[i]); dist_out[qi * knbn + i] = (float)KCNT(S.R[i]) / (float)ix.m; }
            else { ids_out[qi * knbn + i] = ~(uint64_t)0; dist_out[qi * knbn + i] = INFINITY; }
        }
        if (threadIdx.x == 0) { if (count_out) count_out[qi] = nout; if (evals_out) evals_out[qi] = evals; }
    }
}


// ======================================================================================================
// parallel_insert (SPEC 5, batch-synchronous): phase 1 = k_hnsw_plan, one workgroup per new point
// ======================================================================================================
constexpr int PLAN_CH0 = 8;      // first chunk of accepted rows checked against a candidate, then PLAN_CH1
constexpr int PLAN_CH1 = 32;

// Malkov alg. 4 as hnsw_rs::select_neighbours applies it (SPEC 5). Candidates = S.R[0..nW) ascending;
// accepted keys end up in S.A[0..na) ascending. `heuristic == false` -> take all (|W| <= deg, no extension).
// one candidate of the neighbour-selection heuristic against the na keys selected so far, by streaming signature rows (pairs the
// pair cache does not hold): true = no selected s has c(e,s) <= c(x,e)
template <int KIND>
__device__ __forceinline__ bool select_check_rows(const IndexDev &ix, const SearchLds &S, uint64_t e, uint32_t na, uint64_t &evals)
{
    const uint32_t maxdeg = 2 * ix.M;
    const uint4 *erow = (const uint4 *)(ix.data + (uint64_t)KID(e) * ix.stride);
    bool accept = true;
    uint32_t ch = PLAN_CH0;
    for (uint32_t s0 = 0; s0 < na && accept; ) {
        uint32_t nch = na - s0 < ch ? na - s0 : ch;
        if (nch > maxdeg) nch = maxdeg;
        if (threadIdx.x < nch) S.Eid[threadIdx.x] = KID(S.A[s0 + threadIdx.x]);
        __syncthreads();
        block_distances<KIND>(ix, erow, S.Eid, nch, S.Ecnt);
        evals += nch;
        bool conflict = false;
        for (uint32_t t = 0; t < nch; t++) conflict |= (S.Ecnt[t] <= KCNT(e));
        __syncthreads();
        if (conflict) accept = false;
        s0 += nch; ch = PLAN_CH1;
    }
    return accept;
}
// hnsw_rs select_neighbours (SPEC 5): walk W in ascending order, keep e iff no kept s has c(e,s) <= c(x,e), stop at deg. With the pair
// cache the walk takes the candidates SEL_CH at a time: every lane that owns a kept key looks the chunk's candidates up at once (one
// memory round trip and two barriers per chunk instead of per candidate); the chunk ends at its first accepted candidate - the ones
// behind it have to see the new member - so the result is the sequential one.
// ---- sparse pair rows: "is c(hi, lo) <= T ?" for up to NJ pairs per lane at once --------------------------------------------------------
// hi / lo / T come from callables of j (their inputs live in LDS: the lane keeps only the search state in registers). The NJ binary searches
// advance in lockstep - one dependent round trip per step for all of them (13 steps at sp_L = 4096) - with the branch-free "largest p with
// ids[p-1] <= lo" descent: ids are distinct and ascending, so a probe that reads lo itself fixes the answer and no later probe moves past it.
// yes: bit j = count <= T (exact count found); unk: bit j = not decidable from the list (no row, or not stored while T lies above the row's cut).
#define SP_VALID(mt) ((uint32_t)((mt) >> 63))
#define SP_LEN(mt) ((uint32_t)((mt) >> 16) & 0xFFFFu)
#define SP_CUT(mt) ((uint32_t)(mt) & 0xFFFFu)
template <int NJ, class FH, class FL, class FT>
__device__ __forceinline__ void sparse_pairs(const IndexDev &ix, uint32_t want, const FH &fhi, const FL &flo, const FT &fT, uint32_t &yes, uint32_t &unk)
{
    uint32_t base[NJ], len[NJ], off[NJ], eq = 0, cutok = 0;
    uint32_t top = 1; while (top < ix.sp_L) top <<= 1;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        base[j] = 0; len[j] = 0; off[j] = 0;
        if ((want >> j) & 1u) {
            const uint32_t h = fhi(j);
            const uint64_t mt = ix.sp_meta[h];
            off[j] = ix.sp_off[h];
            if (!SP_VALID(mt)) { unk |= 1u << j; want &= ~(1u << j); }
            else { len[j] = SP_LEN(mt); if (fT(j) <= SP_CUT(mt)) cutok |= 1u << j; }
        }
    }
    for (uint32_t step = top; step; step >>= 1) {
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const uint32_t pp = base[j] + step;
            if (((want >> j) & 1u) && pp <= len[j]) {
                const uint32_t v = ((const uint32_t *)(ix.sp_base + ((uint64_t)off[j] << 5)))[pp - 1], l = flo(j);
                if (v <= l) { base[j] = pp; if (v == l) eq |= 1u << j; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        if (!((want >> j) & 1u)) continue;
        if ((eq >> j) & 1u) { if ((uint32_t)((const uint16_t *)(ix.sp_base + ((uint64_t)off[j] << 5) + (uint64_t)4 * len[j]))[base[j] - 1] <= fT(j)) yes |= 1u << j; }
        else if (!((cutok >> j) & 1u)) {                        // not stored: its count is above the cut - decides only thresholds <= cut ...
            const uint32_t h = fhi(j);
            const uint32_t *bm = (fT(j) == SP_CUT(ix.sp_meta[h]) + 1 && ix.sp_bm) ? (const uint32_t *)ix.sp_bm[h] : nullptr;
            if (!bm) unk |= 1u << j;                            // ... and, through the level bitmap, the threshold cut + 1
            else { const uint32_t l = flo(j); if ((bm[l >> 5] >> (l & 31)) & 1u) yes |= 1u << j; }
        }
    }
}
constexpr int SEL_CH = 16;     // (8 -> 16: the walk is bound by its memory round trips per chunk, not by the sectors it fetches)
template <int KIND>
__device__ __forceinline__ uint32_t select_block(const IndexDev &ix, const SearchLds &S, uint32_t nW, uint32_t deg, bool heuristic, uint64_t &evals, uint32_t na0 = 0, unsigned long long *st = nullptr)
{
    if (!heuristic) {
        for (uint32_t t = threadIdx.x; t < nW; t += ST) S.A[t] = S.R[t];
        __syncthreads();
        return nW;
    }
    uint32_t na = na0, i = 0;                                           // na0 > 0: S.A[0..na0) holds what earlier windows of the candidate list kept
    uint32_t *orw = (uint32_t *)&S.scal[4];
    while (i < nW && na < deg) {
        if (na == 0) { if (threadIdx.x == 0) S.A[0] = S.R[i]; na = 1; i++; __syncthreads(); continue; }
        if (!((ix.rowptr || ix.sp_meta) && na <= ST)) {                   // no pair cache: one candidate at a time, rows streamed
            const uint64_t e = S.R[i];
            const bool accept = KCNT(e) < ix.m && select_check_rows<KIND>(ix, S, e, na, evals);
            if (accept) { if (threadIdx.x == 0) S.A[na] = e; na++; __syncthreads(); }
            i++;
            continue;
        }
        const uint32_t nc = nW - i < (uint32_t)SEL_CH ? nW - i : (uint32_t)SEL_CH;
        uint32_t fr = 0;                                                  // c(x,e) = m: c(e,s) <= m for every s, always pruned
#pragma unroll
        for (int j = 0; j < SEL_CH; j++) if ((uint32_t)j < nc && KCNT(S.R[i + j]) >= ix.m) fr |= 1u << j;
        uint32_t cm = 0, pm = 0, pbit = 0;                                // cm: bit j = conflict with my kept key, bit 16 + j = that pair is not cached
        if (threadIdx.x < na) {
            const uint32_t sid = KID(S.A[threadIdx.x]);
            const uint16_t *row[SEL_CH]; uint32_t lo[SEL_CH];
            uint32_t miss = 0;                                            // pairs the dense rows do not hold: the sparse rows' turn
#pragma unroll
            for (int j = 0; j < SEL_CH; j++) {
                row[j] = nullptr; lo[j] = 0;
                if ((uint32_t)j < nc && !((fr >> j) & 1u)) {
                    const uint32_t eid = KID(S.R[i + j]);
                    const uint32_t hi = sid > eid ? sid : eid;
                    lo[j] = sid > eid ? eid : sid;
                    row[j] = ix.rowptr ? (const uint16_t *)ix.rowptr[hi] : nullptr;
                    if (!row[j]) miss |= 1u << j;
                }
            }
#pragma unroll
            for (int j = 0; j < SEL_CH; j++) if (row[j] && (uint32_t)row[j][lo[j]] <= KCNT(S.R[i + j])) cm |= 1u << j;
            if (miss) {
                if (!ix.sp_meta) cm |= miss << 16;
                else {
                    if (st && threadIdx.x == 0) atomicAdd(&st[2], 1ull);          // chunks whose first kept key went to the sparse rows
                    uint32_t yes = 0, unk = 0;
                    sparse_pairs<SEL_CH>(ix, miss, [&](int j) { const uint32_t e = KID(S.R[i + j]); return sid > e ? sid : e; },
                                         [&](int j) { const uint32_t e = KID(S.R[i + j]); return sid > e ? e : sid; }, [&](int j) { return KCNT(S.R[i + j]); }, yes, unk);
                    cm |= yes | (unk << 16);
                }
            }
        } else if (threadIdx.x - na < (uint32_t)(SEL_CH * (SEL_CH - 1) / 2)) {
            static_assert(SEL_CH <= 16, "cm: 16 conflict bits + 16 not-cached bits; pair bits in orw[1..4]");
            // the pairs INSIDE the chunk (j > k): candidate j must also clear the candidates of the chunk that are kept before it.
            // pm: bit (j * (j - 1) / 2 + k) = c(e_j, e_k) <= c(x, e_j); a pair that is not cached counts as "e_j not cached" (bit 16 + j)
            uint32_t t = threadIdx.x - na, j = 1;
            while (t >= j) { t -= j; j++; }
            const uint32_t k = t;
            if (j < nc && !((fr >> j) & 1u) && !((fr >> k) & 1u)) {
                const uint32_t ej = KID(S.R[i + j]), ek = KID(S.R[i + k]);
                const uint32_t hi = ej > ek ? ej : ek, lo2 = ej > ek ? ek : ej;
                const uint16_t *row = ix.rowptr ? (const uint16_t *)ix.rowptr[hi] : nullptr;
                if (row) { if ((uint32_t)row[lo2] <= KCNT(S.R[i + j])) { pbit = j * (j - 1) / 2 + k; pm = 1; } }
                else if (!ix.sp_meta) cm |= 1u << (16 + j);
                else {
                    uint32_t yes = 0, unk = 0;
                    sparse_pairs<1>(ix, 1u, [&](int) { return hi; }, [&](int) { return lo2; }, [&](int) { return KCNT(S.R[i + j]); }, yes, unk);
                    if (unk) cm |= 1u << (16 + j);
                    else if (yes) { pbit = j * (j - 1) / 2 + k; pm = 1; }
                }
            }
        }
        if (threadIdx.x < 6) orw[threadIdx.x] = 0;
        __syncthreads();
        if (cm) atomicOr(&orw[0], cm);
        if (pm) atomicOr(&orw[1 + (pbit >> 5)], 1u << (pbit & 31));
        __syncthreads();
        const uint32_t v = orw[0];
        uint32_t pw[5];
#pragma unroll
        for (int t = 0; t < 5; t++) pw[t] = orw[1 + t];
        // resolve the chunk in order (every lane does the same arithmetic)
        uint32_t kept = 0, j = 0, nacc = 0;
        bool slow = false;
        for (; j < nc && na + nacc < deg; j++) {
            if ((fr >> j) & 1u) continue;
            if ((v >> (16 + j)) & 1u) { slow = true; break; }            // some pair of this candidate is not cached: check it the slow way
            evals += na + nacc;
            bool ok = !((v >> j) & 1u);
            const uint32_t pb = j * (j - 1) / 2, pwi = pb >> 5;                       // conflicts of j with the earlier candidates of the chunk: j bits from bit pb
            const uint64_t two = (uint64_t)(pwi == 0 ? pw[0] : pwi == 1 ? pw[1] : pwi == 2 ? pw[2] : pw[3]) | (uint64_t)(pwi == 0 ? pw[1] : pwi == 1 ? pw[2] : pwi == 2 ? pw[3] : pw[4]) << 32;
            const uint32_t prow = (uint32_t)(two >> (pb & 31)) & ((1u << j) - 1u);
            if (ok && (prow & kept)) ok = false;
            if (ok) { kept |= 1u << j; nacc++; }
        }
        // append the kept ones in order
        if (threadIdx.x == 0) { uint32_t w = na; for (uint32_t q = 0; q < j; q++) if ((kept >> q) & 1u) S.A[w++] = S.R[i + q]; }
        na += nacc;
        if (slow) {
            __syncthreads();
            if (st && threadIdx.x == 0) atomicAdd(&st[1], 1ull);                  // candidates checked by streaming rows (a pair no cache decides)
            const uint64_t e = S.R[i + j];
            if (select_check_rows<KIND>(ix, S, e, na, evals)) { if (threadIdx.x == 0) S.A[na] = e; na++; }
            i += j + 1;
        } else i += j;
        __syncthreads();
    }
    return na;
}

// hnsw_rs select_neighbours with extend_candidates when |W| <= deg (SPEC 5; oracle select_neighbours): the candidate set becomes
// W u { layer-0 neighbours of the members of W }, walked in ascending (c, id) order by the same heuristic. With ef_construction > 2M the
// extension adds nothing (k_hnsw_plan skips it: a result shorter than ef is the whole component); with ef_construction <= 2M - gsearch's
// default --ef 400 with -n 200..255 (gsearch.rs:219-225,268) - EVERY layer-0 selection extends, up to efc * 2M candidates. They are
// gathered per member of W (visited bitmap, distances through the same block_distances as the search), kept as keys in a per-workgroup
// global array E (a few hundred kB, L2 resident), bitonic-sorted there, and fed to select_block in LDS windows. A window that starts at
// distance 1.0 after the first kept neighbour ends the walk: every later candidate is pruned by c(e,s) <= m.
template <int KIND>
__device__ __forceinline__ uint32_t select_extended(const IndexDev &ix, const uint4 *__restrict__ q, const SearchLds &S, uint32_t *vis, uint32_t vis_words, uint32_t nW,
                                                    uint32_t deg, uint32_t win, uint64_t *__restrict__ E, uint32_t capE, const uint16_t *__restrict__ matrow, uint64_t &evals, unsigned long long *st = nullptr)
{
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t w = threadIdx.x; w < vis_words; w += ST) vis[w] = 0;
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < nW; t += ST) {
        const uint64_t kx = S.R[t];
        E[t] = kx;
        if (KID(kx) < ix.n) __hip_atomic_fetch_or(&vis[KID(kx) >> 5], 1u << (KID(kx) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    uint32_t nE = nW;
    for (uint32_t wi = 0; wi < nW; wi++) {
        const uint32_t wid = KID(S.R[wi]);
        if (wid >= ix.n) continue;                                        // a batch-mate: not linked yet, no neighbours
        const uint32_t *nbr; uint32_t dg;
        node_neighbours(ix, wid, 0, nbr, dg);
        bool unv = false; uint32_t id = 0;
        if (threadIdx.x < dg) {
            id = nbr[threadIdx.x];
            const uint32_t bit = 1u << (id & 31);
            const uint32_t old = __hip_atomic_fetch_or(&vis[id >> 5], bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            unv = !(old & bit);
        }
        const uint64_t bal = __ballot(unv);
        if (lane == 0) S.wsum[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = 0, ne = 0;
#pragma unroll
        for (int w = 0; w < ST / 64; w++) { const uint32_t x = S.wsum[w]; if (w < (int)wv) off += x; ne += x; }
        if (unv) S.Eid[off + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = id;
        __syncthreads();
        if (ne == 0) continue;
        evals += ne;
        block_distances<KIND>(ix, q, S.Eid, ne, S.Ecnt, matrow);
        if (threadIdx.x < ne && nE + threadIdx.x < capE) E[nE + threadIdx.x] = KEY(S.Ecnt[threadIdx.x], S.Eid[threadIdx.x]);
        nE += ne;
        __syncthreads();
    }
    if (nE > capE) nE = capE;                                             // (capE >= every reachable count: host sizing)
    uint32_t P = 2; while (P < nE) P <<= 1;
    for (uint32_t i = nE + threadIdx.x; i < P; i += ST) E[i] = ~(uint64_t)0;
    __syncthreads();
    for (uint32_t kk = 2; kk <= P; kk <<= 1)
        for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < P; i += ST) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t a = E[i], b = E[l];
                    if ((a > b) == ((i & kk) == 0)) { E[i] = b; E[l] = a; }
                }
            }
            __syncthreads();
        }
    uint32_t na = 0;
    for (uint32_t pos = 0; pos < nE && na < deg; pos += win) {
        const uint32_t wn = nE - pos < win ? nE - pos : win;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < wn; t += ST) S.R[t] = E[pos + t];
        __syncthreads();
        if (na >= 1 && KCNT(S.R[0]) >= ix.m) break;
        na = select_block<KIND>(ix, S, wn, deg, true, evals, na, st);
    }
    __syncthreads();
    return na;
}

template <int KIND>
__global__ __launch_bounds__(ST) void k_hnsw_plan(IndexDev ix, uint64_t b0, uint32_t nb, const uint8_t *__restrict__ blevels,
                                                   const uint32_t *__restrict__ cntmat, const uint16_t *__restrict__ mat, uint64_t mat_ld, uint32_t efc, uint32_t ef_lds, int extend,
                                                   uint32_t *__restrict__ visited, uint32_t vis_words, int vis_in_lds, uint64_t *__restrict__ plan_keys,
                                                   uint32_t *__restrict__ plan_n, unsigned long long *__restrict__ evals_total,
                                                   const uint64_t *__restrict__ w0_keys, const uint32_t *__restrict__ w0_n, const uint64_t *__restrict__ w0_evals,
                                                   int phase, uint32_t *__restrict__ ep0, uint64_t *__restrict__ ext_keys, uint32_t ext_cap)
{
    // phase 0: the whole plan of a point in one launch. With the pre-pass (plan_prepass) the work is split: phase 1 = the layers above 0
    // of the points that have any (it leaves the layer-0 entry point in ep0), then the pre-pass works W out for every point, then
    // phase 2 = layer 0 (selection on W; the sorted-array search only for the points the pre-pass flagged)
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    const uint32_t maxdeg = 2 * ix.M;
    SearchLds S = carve_lds(s_raw, ef_lds, maxdeg);
    const uint32_t i = blockIdx.x;
    const uint64_t id = b0 + i;
    const int lv = blevels[i];
    if (phase == 1 && lv == 0) return;
    const uint4 *q = (const uint4 *)(ix.data + id * ix.stride);
    // visited bitmap: in LDS behind the search arrays when it fits (a bitmap in global memory costs an L2 atomic per neighbour and
    // thrashes the L2), else this workgroup's slice of the global scratch
    uint32_t *vis = vis_in_lds ? (uint32_t *)(s_raw + ((search_lds_bytes(ef_lds, maxdeg) + 15) & ~(size_t)15)) : visited + (uint64_t)blockIdx.x * vis_words;
    uint64_t evals = 0;
    const bool have_graph = ix.n > 0;
    const uint16_t *matrow = mat ? mat + (uint64_t)i * mat_ld : nullptr;
    uint32_t ep = 0, ep_cnt = 0;
    // a level-0 point whose layer-0 result set W the pre-pass (k_hnsw_search_dense<.., WLOG>) has already worked out: entry point,
    // greedy descent and search_layer are done; only the selection remains
    const bool pre = phase == 2 && have_graph && w0_n[i] != 0xFFFFFFFFu;
    if (have_graph && phase == 2 && lv > 0 && ep0[i] != 0xFFFFFFFFu) {
        if (!pre) { ep = ep0[i]; ep_cnt = matrow[ep]; }           // flagged by the pre-pass: search layer 0 here, from where layer 1 ended
    } else if (have_graph && !pre) {
        if (threadIdx.x == 0) S.Eid[0] = (uint32_t)ix.entry;
        __syncthreads();
        block_distances<KIND>(ix, q, S.Eid, 1, S.Ecnt, matrow);
        ep = (uint32_t)ix.entry; ep_cnt = S.Ecnt[0]; evals++;
        __syncthreads();
        for (int L = ix.top; L > lv; L--) greedy_layer_block<KIND>(ix, q, S, ep, ep_cnt, L, evals, matrow);
    }
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int L = phase == 2 ? 0 : lv; L >= (phase == 1 ? 1 : 0); L--) {
        uint32_t nW = 0;
        if (pre) {
            nW = w0_n[i];
            for (uint32_t t = threadIdx.x; t < nW; t += ST) S.R[t] = w0_keys[(uint64_t)i * efc + t];
            evals += w0_evals[i];
            __syncthreads();
        } else if (have_graph && L <= ix.top) {
            for (uint32_t w = threadIdx.x; w < vis_words; w += ST) vis[w] = 0;
            __syncthreads();
            nW = search_layer_block<KIND>(ix, q, S, vis, ep, ep_cnt, efc, L, evals, matrow);
            ep = KID(S.R[0]); ep_cnt = KCNT(S.R[0]);
        }
        // batch-mates of sufficient level join the candidates (their distances come from the tile kernel)
        uint64_t mykey = ~(uint64_t)0;
        const bool ismate = threadIdx.x < nb && threadIdx.x != i && (int)blevels[threadIdx.x] >= L;
        if (ismate) mykey = KEY(cntmat[(uint64_t)i * nb + threadIdx.x], (uint32_t)(b0 + threadIdx.x));
        __syncthreads();
        if (threadIdx.x < nb) S.C[threadIdx.x] = mykey;
        const uint64_t mb = __ballot(ismate);
        if (lane == 0) S.wsum[wv] = (uint32_t)__popcll(mb);
        __syncthreads();
        uint32_t nm = 0;
#pragma unroll
        for (int w = 0; w < ST / 64; w++) nm += S.wsum[w];
        uint32_t rank = 0;
        if (ismate) for (uint32_t j = 0; j < nb; j++) rank += (S.C[j] < mykey);
        __syncthreads();
        if (ismate) S.C[nb + rank] = mykey;
        __syncthreads();
        if (nm) { const SmallA sm = load_small_a(S.C + nb, nm); nW = block_merge(S.R, 0, nW, S.C + nb, nm, efc, sm); }
        const uint32_t deg = L == 0 ? 2 * ix.M : ix.M;
        const bool ext = (L == 0) && extend;
        uint32_t na = 0;
        if (nW && ext && nW <= deg && ext_keys) {
            // R and C are one contiguous LDS region (carve_lds) and C is dead by now: the selection windows may run over both
            const uint32_t room = 3 * ef_lds + maxdeg, win = room < 512u ? room : 512u;
            na = select_extended<KIND>(ix, q, S, vis, vis_words, nW, deg, win, ext_keys + (uint64_t)i * ext_cap, ext_cap, matrow, evals, evals_total);
        } else if (nW) na = select_block<KIND>(ix, S, nW, deg, !(nW <= deg && !ext), evals, 0, evals_total);
        uint64_t *pk = plan_keys + ((uint64_t)i * ix.max_layer + (uint32_t)L) * maxdeg;
        for (uint32_t t = threadIdx.x; t < na; t += ST) pk[t] = S.A[t];
        if (threadIdx.x == 0) plan_n[(uint64_t)i * ix.max_layer + (uint32_t)L] = na;
        __syncthreads();
    }
    if (phase == 1 && threadIdx.x == 0) ep0[i] = have_graph ? ep : 0xFFFFFFFFu;      // where this point enters layer 0
    if (threadIdx.x == 0) atomicAdd(evals_total, (unsigned long long)evals);
}

// pair cache rows of the nodes that were inserted before the cache existed: rowptr[a] -> row a of an all-pairs count matrix
__global__ void k_set_rowptr(uint16_t *__restrict__ base, uint64_t ld, uint64_t n, uint64_t *__restrict__ rowptr)
{
    const uint64_t a = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a < n && rowptr[a] == 0) rowptr[a] = (uint64_t)(base + a * ld);
}
// pair cache rows of one dense batch: columns [0,b0) come from the tile kernel, [b0,b0+nb) from the mates matrix
__global__ void k_cache_rows(uint16_t *__restrict__ rowbase, uint64_t ld, uint64_t b0, uint32_t nb, const uint32_t *__restrict__ cntmat, uint64_t *__restrict__ rowptr)
{
    const uint32_t i = blockIdx.x;
    uint16_t *row = rowbase + (uint64_t)i * ld;
    for (uint32_t j = threadIdx.x; j < nb; j += blockDim.x) row[b0 + j] = (uint16_t)cntmat[(uint64_t)i * nb + j];
    if (threadIdx.x == 0 && rowptr) rowptr[b0 + i] = (uint64_t)row;          // (rowptr == nullptr: the rows are not kept - group buffer -, only the mates' columns are patched in)
}

// sparse pair row of node a = a0 + blockIdx.x from its count row (counts against every b < a; rows `ld` apart): the cut is the largest count
// level whose nodes all fit - J0 = the smallest number of matches j with #{b < a : m - c(a,b) >= j} <= L, cut = m - J0 (cut = m: the whole row) -
// and the list is written in ascending b (chunks in order, lanes in order, a lane's 8 counts in order: ordered compaction by prefix sums).
// A row whose closest L nodes cannot be separated by level (more than L nodes with >= 63 matches) gets no list (meta 0: the selection streams rows).
constexpr int SPF_T = 256;
// bump allocator of the lists and the level bitmaps (device side: only the kernel knows how long a list is and who needs a bitmap). base .. base + size is
// backed by memory (the host maps more of the arena before a launch could run out); bm_bytes / bm_limit: what the bitmaps alone may take
struct SpArena { unsigned long long base, off, size, stored, noroom, bm_bytes, bm_limit, lists_noroom; };
__global__ __launch_bounds__(SPF_T) void k_sparse_fill(const uint16_t *__restrict__ rowbase, uint64_t ld, uint64_t a0, uint32_t m, uint32_t L,
                                                       uint32_t *__restrict__ sp_off, uint64_t *__restrict__ sp_meta,
                                                       uint64_t *__restrict__ sp_bm, SpArena *__restrict__ arena, uint32_t bm_below_len)
{
    __shared__ uint32_t hist[64], wsum[SPF_T / 64], s_cut, s_base;
    __shared__ unsigned long long s_bm, s_list;
    const uint64_t a = a0 + blockIdx.x;
    const uint16_t *row = rowbase + (uint64_t)blockIdx.x * ld;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 64) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;                      // matches 0..3: ~99 % of an unrelated database - tallied in registers
    for (uint64_t b = threadIdx.x; b < a; b += SPF_T) {
        const uint32_t c = row[b], d = c <= m ? m - c : 0;
        if (d == 0) t0++; else if (d == 1) t1++; else if (d == 2) t2++; else if (d == 3) t3++; else atomicAdd(&hist[d < 63 ? d : 63], 1u);
    }
    if (t0) atomicAdd(&hist[0], t0);
    if (t1) atomicAdd(&hist[1], t1);
    if (t2) atomicAdd(&hist[2], t2);
    if (t3) atomicAdd(&hist[3], t3);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0, j0 = 64;                                // j0 = smallest j with #{d >= j} <= L
        for (int j = 63; j >= 0; j--) { if (run + hist[j] > L) break; run += hist[j]; j0 = (uint32_t)j; }
        s_cut = j0 == 64 ? 0xFFFFFFFFu : m - j0;
        if (j0 != 64 && j0 > m) s_cut = 0xFFFFFFFFu;              // (only when m < 63 and even the pairs with m matches outnumber L: no list)
        s_base = 0; s_list = 0;
        if (s_cut != 0xFFFFFFFFu) {                               // room for `run` entries: 4-byte ids, then their 2-byte counts (32-byte granules)
            const unsigned long long bytes = ((unsigned long long)run * 6 + 31) & ~31ull;
            const unsigned long long off = bytes ? atomicAdd(&arena->off, bytes) : 0ull;
            if (arena->base && off + bytes <= arena->size) s_list = arena->base + off;
            else { s_cut = 0xFFFFFFFFu; atomicAdd(&arena->lists_noroom, 1ull); }
        }
        wsum[0] = run;
    }
    __syncthreads();
    const uint32_t cut = s_cut, nlist = wsum[0];
    if (cut == 0xFFFFFFFFu) { if (threadIdx.x == 0) { sp_meta[a] = 0; sp_off[a] = 0; if (sp_bm) sp_bm[a] = 0; } return; }
    uint32_t *ids = (uint32_t *)s_list; uint16_t *cnt = (uint16_t *)(s_list + (unsigned long long)4 * nlist);
    __syncthreads();                                              // (wsum is reused by the compaction below)
    for (uint64_t c0 = 0; c0 < a; c0 += (uint64_t)SPF_T * 8) {
        const uint64_t b0 = c0 + (uint64_t)threadIdx.x * 8;
        uint32_t cc[8], nk = 0;
#pragma unroll
        for (int h = 0; h < 8; h++) { cc[h] = b0 + h < a ? row[b0 + h] : 0xFFFFFFFFu; nk += cc[h] <= cut; }
        uint32_t inc = nk;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t off = s_base + inc - nk, tot = 0;
#pragma unroll
        for (int w = 0; w < SPF_T / 64; w++) { if (w < (int)wv) off += wsum[w]; tot += wsum[w]; }
#pragma unroll
        for (int h = 0; h < 8; h++) if (cc[h] <= cut) { if (off < nlist) { ids[off] = (uint32_t)(b0 + h); cnt[off] = (uint16_t)cc[h]; } off++; }
        __syncthreads();
        if (threadIdx.x == 0) s_base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sp_meta[a] = ((uint64_t)1 << 63) | ((uint64_t)(s_base < nlist ? s_base : nlist) << 16) | (uint64_t)(cut & 0xFFFFu);
        sp_off[a] = (uint32_t)((s_list - arena->base) >> 5);
        unsigned long long ptr = 0;
        if (sp_bm && arena->bm_limit && cut < m && s_base < bm_below_len && a > 0) {        // (cut == m: the whole row is listed, nothing lies below)
            const unsigned long long bytes = ((a + 31) / 32 * 4 + 31) & ~31ull;
            if (atomicAdd(&arena->bm_bytes, bytes) + bytes <= arena->bm_limit) {
                const unsigned long long off = atomicAdd(&arena->off, bytes);
                if (arena->base && off + bytes <= arena->size) { ptr = arena->base + off; atomicAdd(&arena->stored, 1ull); }
                else atomicAdd(&arena->noroom, 1ull);
            } else atomicAdd(&arena->noroom, 1ull);
        }
        s_bm = ptr;
        if (sp_bm) sp_bm[a] = ptr;
    }
    __syncthreads();
    if (s_bm) {
        uint32_t *bm = (uint32_t *)s_bm;
        const uint32_t lvl = cut + 1;                                 // counts <= cut + 1 <=> matches >= J0 - 1
        for (uint64_t w = threadIdx.x; w < (a + 31) / 32; w += SPF_T) {
            uint32_t bits = 0;
#pragma unroll 8
            for (int h = 0; h < 32; h++) { const uint64_t b = w * 32 + h; if (b < a && (uint32_t)row[b] <= lvl) bits |= 1u << h; }
            bm[w] = bits;
        }
    }
}

// ---- phase 2: links -------------------------------------------------------------------------------
struct GraphDev {          // mutable adjacency
    uint32_t *deg0, *nbr0, *cnt0; uint32_t *degU, *nbrU, *cntU; const int32_t *upidx;
    uint32_t M, max_layer; uint64_t upper_base;
};
__device__ __forceinline__ void list_ptrs(const GraphDev &g, uint32_t list, uint32_t *&nbr, uint32_t *&cnt, uint32_t *&deg, uint32_t &cap)
{
    if (list < g.upper_base) { nbr = g.nbr0 + (uint64_t)list * 2 * g.M; cnt = g.cnt0 + (uint64_t)list * 2 * g.M; deg = g.deg0 + list; cap = 2 * g.M; }
    else { uint64_t x = list - g.upper_base; nbr = g.nbrU + x * g.M; cnt = g.cntU + x * g.M; deg = g.degU + x; cap = g.M; }
}
__device__ __forceinline__ uint32_t list_of(const GraphDev &g, uint32_t node, uint32_t L)
{
    return L == 0 ? node : (uint32_t)(g.upper_base + (uint64_t)g.upidx[node] * g.max_layer + (L - 1));
}
// own lists of the new nodes
__global__ void k_link_own(GraphDev g, uint64_t b0, uint32_t nb, const uint8_t *__restrict__ blevels, const uint64_t *__restrict__ plan_keys,
                           const uint32_t *__restrict__ plan_n)
{
    const uint32_t i = blockIdx.x, maxdeg = 2 * g.M;
    for (uint32_t L = 0; L <= blevels[i]; L++) {
        uint32_t *nbr, *cnt, *deg, cap;
        list_ptrs(g, list_of(g, (uint32_t)(b0 + i), L), nbr, cnt, deg, cap);
        const uint32_t n = plan_n[(uint64_t)i * g.max_layer + L];
        const uint64_t *pk = plan_keys + ((uint64_t)i * g.max_layer + L) * maxdeg;
        for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) { nbr[t] = KID(pk[t]); cnt[t] = KCNT(pk[t]); }
        if (threadIdx.x == 0) *deg = n;
    }
}
// reverse links -> per-list inbox
__global__ void k_link_scatter(GraphDev g, uint64_t b0, uint32_t nb, uint32_t inbox_cap, const uint8_t *__restrict__ blevels,
                               const uint64_t *__restrict__ plan_keys, const uint32_t *__restrict__ plan_n, uint32_t *__restrict__ inbox_cnt,
                               uint64_t *__restrict__ inbox, uint32_t *__restrict__ touched, uint32_t *__restrict__ ntouched)
{
    const uint32_t i = blockIdx.x, maxdeg = 2 * g.M;
    for (uint32_t L = 0; L <= blevels[i]; L++) {
        const uint32_t n = plan_n[(uint64_t)i * g.max_layer + L];
        const uint64_t *pk = plan_keys + ((uint64_t)i * g.max_layer + L) * maxdeg;
        for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
            const uint32_t list = list_of(g, KID(pk[t]), L);
            const uint32_t pos = atomicAdd(&inbox_cnt[list], 1u);
            inbox[(uint64_t)list * inbox_cap + pos] = KEY(KCNT(pk[t]), (uint32_t)(b0 + i));
            if (pos == 0) touched[atomicAdd(ntouched, 1u)] = list;
        }
    }
}
// every touched list := the `cap` smallest keys of (list  u  inbox), duplicates removed (SPEC 5: order-free)
constexpr int LM_T = 256, LM_MAX = 1024;
__global__ __launch_bounds__(LM_T) void k_link_merge(GraphDev g, uint32_t inbox_cap, uint32_t *__restrict__ inbox_cnt, const uint64_t *__restrict__ inbox,
                                                      const uint32_t *__restrict__ touched, const uint32_t *__restrict__ ntouched)
{
    __shared__ uint64_t keys[LM_MAX];
    __shared__ uint32_t flag[LM_MAX];
    const uint32_t nt = *ntouched;
    for (uint32_t b = blockIdx.x; b < nt; b += gridDim.x) {
        const uint32_t list = touched[b];
        uint32_t *nbr, *cnt, *deg, cap;
        list_ptrs(g, list, nbr, cnt, deg, cap);
        const uint32_t d = *deg, ni = inbox_cnt[list], n = d + ni;
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < n; t += LM_T)
            keys[t] = t < d ? KEY(cnt[t], nbr[t]) : inbox[(uint64_t)list * inbox_cap + (t - d)];
        // The list as it stands is sorted (this kernel and k_link_own write keys at their rank) unless it came in through import_graph:
        // then a key's rank is its position plus the inbox keys below it, and an inbox key's rank a binary search plus the inbox keys
        // below it - O(d * ni) instead of the O(n^2) of ranking everything against everything (1.4 s of a 300 k build; ni is a handful).
        bool unsorted = false;
        for (uint32_t t = threadIdx.x + 1; t < d; t += LM_T) unsorted |= !(keys[t - 1] < keys[t]);
        const bool general = __syncthreads_or(unsorted) != 0;
        uint32_t rk[LM_MAX / LM_T]; uint64_t kv[LM_MAX / LM_T];
        uint32_t ndup = 0;
        if (general) {
            // duplicates (the same undirected edge proposed from both ends inside one batch) have identical keys
            for (uint32_t t = threadIdx.x; t < n; t += LM_T) {
                const uint64_t k = keys[t]; uint32_t dup = 0;
                for (uint32_t j = 0; j < t; j++) dup |= (keys[j] == k);
                flag[t] = dup;
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < LM_MAX / LM_T; it++) {
                const uint32_t t = threadIdx.x + it * LM_T;
                rk[it] = 0xFFFFFFFFu; kv[it] = 0;
                if (t < n && !flag[t]) {
                    const uint64_t k = keys[t]; uint32_t r = 0;
                    for (uint32_t j = 0; j < n; j++) r += (!flag[j] && keys[j] < k);
                    rk[it] = r; kv[it] = k;
                }
            }
            for (uint32_t j = 0; j < n; j++) ndup += flag[j];
        } else {
            // an inbox key is a duplicate when the list holds it already or an earlier inbox entry equals it; flag[d + u] = dup | (lower bound << 1)
            for (uint32_t u = threadIdx.x; u < ni; u += LM_T) {
                const uint64_t k = keys[d + u];
                uint32_t lo = 0, hi = d;
                while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (keys[mid] < k) lo = mid + 1; else hi = mid; }
                uint32_t dup = lo < d && keys[lo] == k;
                for (uint32_t v = 0; v < u; v++) dup |= (keys[d + v] == k);
                flag[d + u] = dup | (lo << 1);
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < LM_MAX / LM_T; it++) {
                const uint32_t t = threadIdx.x + it * LM_T;
                rk[it] = 0xFFFFFFFFu; kv[it] = 0;
                if (t < d) {
                    const uint64_t k = keys[t]; uint32_t r = t;
                    for (uint32_t v = 0; v < ni; v++) r += (!(flag[d + v] & 1u) && keys[d + v] < k);
                    rk[it] = r; kv[it] = k;
                } else if (t < n && !(flag[t] & 1u)) {
                    const uint64_t k = keys[t]; uint32_t r = flag[t] >> 1;
                    for (uint32_t v = 0; v < ni; v++) r += (!(flag[d + v] & 1u) && keys[d + v] < k);
                    rk[it] = r; kv[it] = k;
                }
            }
            for (uint32_t v = 0; v < ni; v++) ndup += flag[d + v] & 1u;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < LM_MAX / LM_T; it++) if (rk[it] < cap) { nbr[rk[it]] = KID(kv[it]); cnt[rk[it]] = KCNT(kv[it]); }
        if (threadIdx.x == 0) { const uint32_t tot = n - ndup; *deg = tot < cap ? tot : cap; inbox_cnt[list] = 0; }
    }
}

}  // namespace gs

// ------------------------------------------------------------------------------------------------------
struct gs_index {
    gs_ctx *ctx = nullptr;
    gs_index_params prm{};
    size_t esz = 4, rowbytes = 0; uint64_t stride = 0; uint32_t nchunks = 0;     // INTERNAL element size / row bytes
    int ikind = GS_KIND_F32; size_t user_rowbytes = 0;      // u16 signatures (hll) are held zero-extended to u32 (gs_hamming.hip k_widen_u16)
    uint64_t n = 0, cap = 0; int64_t entry = -1; int top = -1;
    uint64_t n_upper = 0, cap_upper = 0;
    gs::DevBuf data, levels, deg0, nbr0, cnt0, upidx, degU, nbrU, cntU;
    gs::DevBuf visited, counter, cbuf;
    // insert scratch
    gs::DevBuf blevels, cntmat, plan_keys, plan_n, inbox, inbox_cnt, touched, ntouched, evals_dev;
    gs::DevBuf wlog, w0_keys, w0_n, w0_evals, ep0; // insert pre-pass (plan_prepass)
    gs::DevBuf ext_keys;                           // candidate keys of the extended selection (extend_candidates with efc <= 2M)
    hipStream_t jstream = nullptr; hipEvent_t jev = nullptr, jev_up = nullptr;      // insert: the match-join of batch i+1 runs here under plan / links of batch i
    uint64_t inbox_lists = 0;
    uint64_t insert_evals = 0;
    // dense mode: count matrix of a query / insert batch against every node, and the running
    // fraction of the graph a traversal evaluates (negative = not measured yet)
    gs::DevBuf mat;
    double search_frac = -1.0, insert_frac = -1.0;
    // pair cache: the count rows the tile kernel produces for every inserted batch are KEPT (16 bit) so that
    // the neighbour-selection heuristic looks pair distances up instead of streaming rows; 288 GB of HBM hold it up to ~500 k points
    // column-major copy of the signatures for the match-join (gs_join.hip)
    gs::DevBuf cols; uint64_t cols_cap = 0, cols_n = 0;
    gs::DevBuf join_scratch[gs::JOIN_SCRATCH];
    gs::DevBuf stats;                 // device work counters: [0] join atomics, [1] dense-traversal pops, [2] accepting pops (gs_index_search_stats)
    uint64_t stat_wg_in_flight = 0, stat_adj_row_bytes = 0;
    gs::DevBuf rowptr;
    // sparse pair rows (IndexDev::sp_*): allocated at the first dense insert batch, grown with the index; sp_L = 0: off (GS_SPARSE_ROWS=0, m > 65535, no memory)
    gs::DevBuf sp_off, sp_meta, sp_bm, sp_arena; uint32_t sp_L = 0; bool sp_tried = false;
    gs::VmArena sp_vm;                // lists + level bitmaps (grow in place; sp_arena = the device-side bump allocator over it)
    bool sp_exhausted = false;        // the arena (or the device) is full: lists are still handed out while they fit, nothing more is mapped
    uint64_t sp_reserved = 0;         // upper bound of the arena bytes handed out so far (worst case per launch; corrected from the device when it runs out)
    std::vector<gs::DevBuf *> slabs;
    uint64_t pair_cache_bytes = 0, pair_cache_budget = 0;
    bool early_cached = false;        // the nodes older than the first cached batch have all-pairs rows (insert_common)
    // hnsw_rs' DataId / PointId (answer.rs:42-57 reads Neighbour{d_id, distance, p_id}): nodes are numbered 0.. in insertion order inside the
    // library; `origin` maps them to the ids the caller inserted them under (empty = the caller's ids ARE 0.. in order, gsearch's own case,
    // dnasketch.rs:429-433), pid_rank[i] = rank of node i among the nodes of its level (PointId = (level, rank))
    std::vector<uint64_t> origin;
    std::vector<int32_t> pid_rank;
    uint64_t level_count[17] = {0};
    gs::DevBuf origin_d, pid_rank_d; uint64_t origin_d_n = 0, pid_rank_d_n = 0;
    // gs_index_sketch_and_search_dev: the padded query rows are produced batch by batch while the search is under way; dense_counts asks for
    // rows [q0, q0 + nb) of the buffer at feed_base just before it joins them
    std::function<int(uint64_t, uint64_t)> *feed = nullptr; const uint8_t *feed_base = nullptr;
    // three-stage request pipeline (round 5, GS_REQUEST_PIPELINE=3): the traversal of join batch b runs on this stream beside the count matrix of batch b + 1
    // and the sketch of batch b + 2 (search_dev, dense strategy)
    hipStream_t tstream = nullptr; hipEvent_t tev = nullptr, tev_done = nullptr; bool pipe3 = false;
    ~gs_index()
    {
        if (tstream) { (void)hipStreamSynchronize(tstream); (void)hipStreamDestroy(tstream); }
        if (tev) (void)hipEventDestroy(tev);
        if (tev_done) (void)hipEventDestroy(tev_done);
        if (jstream) { (void)hipStreamSynchronize(jstream); (void)hipStreamDestroy(jstream); }
        if (jev) (void)hipEventDestroy(jev);
        if (jev_up) (void)hipEventDestroy(jev_up);
        for (auto *b : slabs) delete b;
    }
};

namespace gs {

static void ids_append(gs_index *ix, const uint64_t *ids, const uint8_t *lv, uint64_t n);

// Gives the insert-time pair cache back. Its slabs are an optimisation (the plan kernel streams the rows of pairs that are not cached);
// the signatures, their column copy and the count matrix are not, so when one of THOSE cannot be allocated the cache goes and stays off.
static void drop_pair_cache(gs_index *ix)
{
    (void)hipGetLastError();                                       // the failed hipMalloc left its error behind
    if (!ix->slabs.empty()) {
        if (ix->jstream) (void)hipStreamSynchronize(ix->jstream);
        (void)hipStreamSynchronize(ix->ctx->stream);
        if (ix->rowptr.p) (void)hipMemsetAsync(ix->rowptr.p, 0, ix->rowptr.bytes, ix->ctx->stream);
        (void)hipStreamSynchronize(ix->ctx->stream);
        for (auto *b : ix->slabs) delete b;
        ix->slabs.clear();
    }
    ix->pair_cache_bytes = 0; ix->pair_cache_budget = 1; ix->early_cached = false;
}
// DevBuf::alloc that pays with the pair cache when the device is full
static int alloc_or_evict(gs_index *ix, DevBuf &b, size_t bytes)
{
    int rc = b.alloc(bytes);
    if (rc == GS_OK || (ix->slabs.empty() && ix->pair_cache_budget == 1)) return rc;
    drop_pair_cache(ix);
    return b.alloc(bytes);
}

static int index_reserve(gs_index *ix, uint64_t need, uint64_t need_upper)
{
    gs_ctx *c = ix->ctx;
    const uint32_t M = ix->prm.max_nb_conn, ML = ix->prm.max_layer;
    if (need > ix->cap) {
        uint64_t ncap = std::max<uint64_t>(need, std::max<uint64_t>(ix->cap + ix->cap / 2, 1024));
        // never past the capacity the caller declared (hnsw_params.capacity, 1.5 M in gsearch) while the points still fit it: the last growth step of a large index
        // is the one that decides whether it fits the device at all
        if (ix->prm.capacity >= need && ncap > ix->prm.capacity) ncap = ix->prm.capacity;
        // the column copy is rebuilt from the rows at the new capacity anyway (ensure_cols): give it back BEFORE the rows are copied - at 1 M+ genomes the old and
        // the new signature block next to it would not fit otherwise
        if (ix->cols.p && (uint64_t)ix->stride * ncap > ((uint64_t)8 << 30)) { ix->cols.release(); ix->cols_cap = 0; ix->cols_n = 0; }
        struct Arr { DevBuf *b; size_t per; };
        std::vector<Arr> arr = {
            {&ix->data, (size_t)ix->stride}, {&ix->levels, 1}, {&ix->deg0, 4}, {&ix->nbr0, (size_t)8 * M}, {&ix->cnt0, (size_t)8 * M}, {&ix->upidx, 4}, {&ix->rowptr, 8}};
        if (ix->sp_L && ix->sp_meta.p) { arr.push_back({&ix->sp_off, 4}); arr.push_back({&ix->sp_meta, 8}); arr.push_back({&ix->sp_bm, 8}); }
        for (auto &a : arr) {
            DevBuf nb;
            int rc = alloc_or_evict(ix, nb, a.per * ncap); if (rc) return rc;
            GS_HIP_CHECK(hipMemsetAsync(nb.p, 0, a.per * ncap, c->stream));
            if (ix->n) GS_HIP_CHECK(hipMemcpyAsync(nb.p, a.b->p, a.per * ix->n, hipMemcpyDeviceToDevice, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            std::swap(a.b->p, nb.p); std::swap(a.b->bytes, nb.bytes);
        }
        ix->cap = ncap;
    }
    if (need_upper > ix->cap_upper) {
        uint64_t ncap = std::max<uint64_t>(need_upper, std::max<uint64_t>(ix->cap_upper * 2, 64));
        struct { DevBuf *b; size_t per; } arr[] = {{&ix->degU, (size_t)4 * ML}, {&ix->nbrU, (size_t)4 * ML * M}, {&ix->cntU, (size_t)4 * ML * M}};
        for (auto &a : arr) {
            DevBuf nb;
            int rc = nb.alloc(a.per * ncap); if (rc) return rc;
            GS_HIP_CHECK(hipMemsetAsync(nb.p, 0, a.per * ncap, c->stream));
            if (ix->n_upper) GS_HIP_CHECK(hipMemcpyAsync(nb.p, a.b->p, a.per * ix->n_upper, hipMemcpyDeviceToDevice, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            std::swap(a.b->p, nb.p); std::swap(a.b->bytes, nb.bytes);
        }
        ix->cap_upper = ncap;
    }
    return GS_OK;
}

static IndexDev index_dev(const gs_index *ix)
{
    IndexDev d;
    d.data = ix->data.as<uint8_t>(); d.stride = ix->stride; d.nchunks = ix->nchunks; d.m = ix->prm.m;
    d.M = ix->prm.max_nb_conn; d.max_layer = ix->prm.max_layer;
    d.levels = ix->levels.as<uint8_t>(); d.deg0 = ix->deg0.as<uint32_t>(); d.nbr0 = ix->nbr0.as<uint32_t>();
    d.upidx = ix->upidx.as<int32_t>(); d.degU = ix->degU.as<uint32_t>(); d.nbrU = ix->nbrU.as<uint32_t>();
    d.rowptr = ix->rowptr.as<uint64_t>();
    const bool sp = ix->sp_L && ix->sp_meta.p;
    d.sp_base = sp ? (const uint8_t *)ix->sp_vm.va : nullptr; d.sp_off = sp ? ix->sp_off.as<uint32_t>() : nullptr; d.sp_meta = sp ? ix->sp_meta.as<uint64_t>() : nullptr; d.sp_L = sp ? ix->sp_L : 0;
    d.sp_bm = sp ? ix->sp_bm.as<uint64_t>() : nullptr;
    d.n = ix->n; d.entry = ix->entry; d.top = ix->top;
    return d;
}

// copy dense rows (host or device) into a zero-padded strided device buffer
static int upload_rows(gs_ctx *c, void *dst, uint64_t stride, const void *src, size_t rowbytes, uint64_t nrows, hipMemcpyKind kind)
{
    if (nrows == 0) return GS_OK;
    GS_HIP_CHECK(hipMemsetAsync(dst, 0, stride * nrows, c->stream));
    GS_HIP_CHECK(hipMemcpy2DAsync(dst, stride, src, rowbytes, rowbytes, nrows, kind, c->stream));
    return GS_OK;
}

// rows as the caller holds them (u16 for hll) -> internal zero-padded strided rows; nrows rows from host or device memory
static int upload_user_rows(gs_index *ix, void *dst, const void *src, uint64_t nrows, hipMemcpyKind kind)
{
    gs_ctx *c = ix->ctx;
    if (ix->prm.kind != GS_KIND_U16) return upload_rows(c, dst, ix->stride, src, ix->rowbytes, nrows, kind);
    if (nrows == 0) return GS_OK;
    GS_HIP_CHECK(hipMemsetAsync(dst, 0, ix->stride * nrows, c->stream));
    const void *dev = src;
    PoolBuf stage(c, 39);
    if (kind == hipMemcpyHostToDevice) {
        int rc = stage.alloc(ix->user_rowbytes * nrows); if (rc) return rc;
        GS_HIP_CHECK(hipMemcpyAsync(stage.p, src, ix->user_rowbytes * nrows, hipMemcpyHostToDevice, c->stream));
        dev = stage.p;
    }
    return widen_u16_rows(c, dev, nrows, ix->prm.m, dst, ix->stride);
}
// internal rows [first, first+n) -> dense rows as the caller holds them, in host memory
static int download_user_rows(gs_index *ix, uint64_t first, uint64_t n, void *out_host)
{
    gs_ctx *c = ix->ctx;
    if (n == 0) return GS_OK;
    if (ix->prm.kind != GS_KIND_U16) {
        GS_HIP_CHECK(hipMemcpy2DAsync(out_host, ix->rowbytes, ix->data.as<uint8_t>() + first * ix->stride, ix->stride, ix->rowbytes, n, hipMemcpyDeviceToHost, c->stream));
    } else {
        PoolBuf stage(c, 39);
        int rc = stage.alloc(ix->user_rowbytes * n); if (rc) return rc;
        if ((rc = narrow_u16_rows(c, ix->data.as<uint8_t>() + first * ix->stride, ix->stride, n, ix->prm.m, stage.p))) return rc;
        GS_HIP_CHECK(hipMemcpyAsync(out_host, stage.p, ix->user_rowbytes * n, hipMemcpyDeviceToHost, c->stream));
    }
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

// ---- distance-evaluation strategy -----------------------------------------------------
// gather: every evaluation streams one candidate row from HBM (72 kB at s=18000) — right when a traversal touches a
//         small part of the graph;
// dense : the counts of the whole query batch against EVERY node are produced first by the tile kernel (each row is
//         read once per 128 queries, VALU-bound), the traversal then only looks counts up — right when traversals
//         flood the graph (ties at distance 1.0 with ef in the thousands: they evaluate most of the DB anyway).
// Both give bit-identical results; GS_DIST_MODE=gather|dense|auto overrides the cost model.
enum DistMode { MODE_AUTO = 0, MODE_GATHER = 1, MODE_DENSE = 2 };
static DistMode env_mode()
{
    const char *e = getenv("GS_DIST_MODE");
    if (!e) return MODE_AUTO;
    if (!strcmp(e, "gather")) return MODE_GATHER;
    if (!strcmp(e, "dense")) return MODE_DENSE;
    return MODE_AUTO;
}
static bool use_join(const gs_index *ix);
// cost model of a batch of nq traversals that evaluate frac*n nodes each (seconds; constants measured on MI355X, profiles/):
//   gather : one row streamed per evaluation: ~5.5e12 B/s for the whole GPU, but one workgroup per query and a workgroup alone
//            pulls ~8e10 B/s, so a small batch is bound by the per-query stream
//   dense  : the count matrix of the batch first - match-join: the column store once per batch (~3.5e12 B/s for a few hundred
//            queries, ~1e12 B/s effective for thousands: the probe work per value grows) + ~3 ms of fixed cost; compare tile:
//            n*m element compares at ~1.6e13/s per query - then latency-bound traversals: ~75 ns per evaluation, 768 in flight
static bool dense_pays(const gs_index *ix, double frac, uint64_t nq)
{
    // The six rates below were measured on an MI355X (profiles/). They are scaled by what THIS device reports against that one - streaming rates by the
    // memory system (rel_hbm), the join's, the tile kernel's and the traversal's by CUs x clock (rel_compute) - instead of being taken as they are;
    // on the device they were measured on both factors are within a few percent of 1 and the decisions are the measured ones (test_cost_model_picks_..., 5 operating points).
    const double rc = ix->ctx->rel_compute, rh = ix->ctx->rel_hbm;
    const double n = (double)ix->n, q = (double)std::max<uint64_t>(nq, 1), evals = frac * n;
    const double gather = evals * (double)ix->rowbytes * std::max(q / (5.5e12 * rh), 1.0 / (8.0e10 * rh));
    double dense;
    if (use_join(ix)) dense = n * (double)ix->rowbytes / ((nq <= 512 ? 3.5e12 : 1.0e12) * rc) * std::ceil(q / (double)match_join_max_queries()) + 3e-3;
    else dense = std::ceil(q / 128.0) * 128.0 * n * (double)ix->prm.m / ((ix->ikind == GS_KIND_U64 ? 1.4e13 : 1.6e13) * rc);       // 128-query tiles
    dense += evals * 75e-9 / rc * std::max(1.0, q / 768.0);
    return dense < gather;
}
static bool use_join(const gs_index *ix)
{
    const char *e = getenv("GS_DENSE_IMPL");
    if (e && !strcmp(e, "tile")) return false;
    return ix->prm.m <= 65535;
}
// column-major copy of the signatures of nodes [0, upto): rebuilt when the capacity changed, appended otherwise
static int ensure_cols(gs_index *ix, uint64_t upto)
{
    gs_ctx *c = ix->ctx;
    int rc;
    if (ix->cols_cap != ix->cap || !ix->cols.p) {
        if ((rc = alloc_or_evict(ix, ix->cols, (size_t)ix->prm.m * ix->cap * ix->esz))) return rc;
        ix->cols_cap = ix->cap; ix->cols_n = 0;
    }
    if (ix->cols_n < upto) {
        if ((rc = rows_to_cols(c, ix->ikind, ix->prm.m, ix->data.as<uint8_t>() + ix->cols_n * ix->stride, ix->stride, upto - ix->cols_n, ix->cols.p, ix->cols_cap, ix->cols_n))) return rc;
        ix->cols_n = upto;
    }
    return GS_OK;
}
static int ensure_stats(gs_index *ix)
{
    if (ix->stats.p) return GS_OK;
    int rc = ix->stats.alloc(128);
    if (rc) return rc;
    GS_HIP_CHECK(hipMemsetAsync(ix->stats.p, 0, 128, ix->ctx->stream));
    return GS_OK;
}
// counts of nq padded query rows against nodes [0, n): out16[q * ld + e]
static int dense_counts(gs_index *ix, const uint8_t *qrows, uint64_t nq, uint64_t n, uint16_t *out16, uint64_t ld)
{
    gs_ctx *c = ix->ctx;
    int rc;
    if (!use_join(ix))
        return hamming_qxc_strided(c, ix->ikind, ix->prm.m, qrows, nq, ix->stride, ix->data.p, n, ix->stride, nullptr, nullptr, out16, ld);
    if ((rc = ensure_cols(ix, n))) return rc;
    if ((rc = ensure_stats(ix))) return rc;
    // equal batches: every join call streams all the columns, so a short last batch costs almost as much as a full one
    const uint64_t parts = (nq + match_join_max_queries() - 1) / match_join_max_queries(), jq = (nq + parts - 1) / parts;
    for (uint64_t q0 = 0; q0 < nq; q0 += jq) {
        const uint64_t nb = std::min<uint64_t>(jq, nq - q0);
        int declined = 0;
        if (ix->feed && (rc = (*ix->feed)((uint64_t)(qrows - ix->feed_base) / ix->stride + q0, nb))) return rc;
        if ((rc = match_join_counts(c, ix->ikind, ix->prm.m, qrows + q0 * ix->stride, ix->stride, nb, ix->cols.p, ix->cols_cap, n, out16 + q0 * ld, ld, ix->join_scratch,
                                    &declined, ix->stats.as<unsigned long long>(), true, 0, ix->data.p, ix->stride))) return rc;
        if (declined &&      // too many matches to record one by one (redundant queries against a redundant database): fixed-cost compare kernel
            (rc = hamming_qxc_strided(c, ix->ikind, ix->prm.m, qrows + q0 * ix->stride, nb, ix->stride, ix->data.p, n, ix->stride, nullptr, nullptr, out16 + q0 * ld, ld))) return rc;
    }
    return GS_OK;
}

// counts of nq rows against the nodes [node0, node0 + nn) only, into columns node0.. of rows whose counters were initialised (to m) by an earlier
// dense_counts of the same rows: the join over those nodes' columns (a compare-tile launch this small - 256 x <= 1792 pairs - fills 28 of 256
// CUs and takes 5 ms)
static int dense_counts_range(gs_index *ix, const uint8_t *qrows, uint64_t nq, uint64_t node0, uint64_t nn, uint16_t *out16, uint64_t ld)
{
    gs_ctx *c = ix->ctx;
    int rc;
    if ((rc = ensure_cols(ix, node0 + nn))) return rc;
    if ((rc = ensure_stats(ix))) return rc;
    GS_REQUIRE(nq <= match_join_max_queries(), GS_ERR_INVALID, "dense_counts_range: too many rows");
    return match_join_counts(c, ix->ikind, ix->prm.m, qrows, ix->stride, nq, (const uint8_t *)ix->cols.p + node0 * ix->esz, ix->cols_cap, nn, out16, ld, ix->join_scratch,
                             nullptr, ix->stats.as<unsigned long long>(), false, node0);
}

// placement for the dense traversal: visited bitmap in LDS when that still leaves >= 2 workgroups per CU (GS_DENSE_VIS=lds|global overrides)
static bool dense_vis_in_lds(const gs_index *ix, uint32_t knbn, uint32_t maxdeg)
{
    const size_t cap = 160 * 1024 - 1024;
    const size_t l = dense_lds_bytes(ix->prm.m, knbn, maxdeg, ix->n, true, maxdeg > (uint32_t)DT / 2 ? 512u : (uint32_t)DCN);
    const char *e = getenv("GS_DENSE_VIS");
    if (e && !strcmp(e, "global")) return false;
    if (e && !strcmp(e, "lds")) return l <= cap;
    // one workgroup per CU with the bitmap in LDS still beats the global bitmap by far (300 k nodes, 10 k queries: 49 ms at three per CU,
    // ~100 ms at one, 446 ms with the bitmap in global memory - every neighbour a memory-side atomic; profiles/r04_trav_placements.txt):
    // the LDS placement is kept up to ~1.08 M nodes
    return l <= cap;
}
// SPLIT placement (round 5): the LDS bitmap maps node ids [0, W), this workgroup's global scratch the rest. W = what lets `per_cu` workgroups share a
// CU (GS_SPLIT_PER_CU, default 2; GS_SPLIT_W overrides W itself - tests), a multiple of 1024; 0 = not possible / not wanted. Taken whenever the
// whole bitmap does not fit the LDS (GS_DENSE_VIS=global keeps the round-4 all-global form, GS_DENSE_VIS=split forces the split at any size).
static uint32_t dense_split_w(const gs_index *ix, uint32_t knbn, uint32_t maxdeg, uint32_t dcn, size_t min_bytes = 0, bool many_queries = false)
{
    const char *e = getenv("GS_DENSE_VIS");
    const bool forced = e && !strcmp(e, "split");
    if (e && !forced) { if (!strcmp(e, "global") || !strcmp(e, "lds")) return 0; }
    // a request of many queries also takes the split form where the whole bitmap WOULD fit the LDS but only with one workgroup per CU (~600 k - 1.2 M nodes):
    // two workgroups per CU with the upper ids behind the Bloom filter beat one with everything in LDS (1 M nodes, 10 000 queries: 101.6 against 121.4 ms,
    // profiles/r05_request_1M_placements.log). The insert pre-pass (256 points per batch: one workgroup per CU anyway) keeps the plain form.
    const bool one_per_cu = many_queries && round_up(dense_lds_bytes(ix->prm.m, knbn, maxdeg, ix->n, true, dcn), 1280) > (size_t)(160 * 1024 / 2);
    if (!forced && !one_per_cu && dense_vis_in_lds(ix, knbn, maxdeg)) return 0;
    const size_t base = std::max(dense_lds_bytes(ix->prm.m, knbn, maxdeg, 0, true, dcn), min_bytes);
    int per = getenv("GS_SPLIT_PER_CU") ? std::max(1, std::min(3, atoi(getenv("GS_SPLIT_PER_CU")))) : 2;
    uint64_t w = 0;
    for (; per >= 1 && w < 1024; per--) {
        const size_t budget = (size_t)(160 * 1024 / per) / 1280 * 1280 - 64;
        w = budget > base ? ((uint64_t)(budget - base) * 8) & ~(uint64_t)1023 : 0;
    }
    if (getenv("GS_SPLIT_W")) w = std::min<uint64_t>(w, (uint64_t)std::max(1024, atoi(getenv("GS_SPLIT_W"))) & ~(uint64_t)1023);
    if (w < 1024) return 0;
    return (uint32_t)std::min<uint64_t>(w, round_up(ix->n, 1024));
}
static int search_launch_dense(gs_index *ix, uint64_t nq, uint32_t knbn, uint32_t ef, const uint16_t *mat, uint64_t mat_ld, uint64_t *ids, float *dist,
                               uint32_t *count, uint64_t *evals)
{
    gs_ctx *c = ix->ctx;
    const uint32_t efs = std::max(ef, knbn);
    const uint32_t maxdeg = 2 * ix->prm.max_nb_conn;
    const bool oneg = maxdeg > (uint32_t)DT / 2;                       // rows of more than 256 ids: one 512-lane group instead of two halves
    const uint32_t dcn = oneg ? 512u : (uint32_t)DCN;
    const uint32_t vis_w = dense_split_w(ix, knbn, maxdeg, dcn, 0, nq >= (uint64_t)2 * c->n_cu);
    const bool split = vis_w != 0;
    const bool vlds = split || dense_vis_in_lds(ix, knbn, maxdeg);
    const size_t lds = dense_lds_bytes(ix->prm.m, knbn, maxdeg, split ? (uint64_t)vis_w : ix->n, vlds, dcn);
    // three 8-wave workgroups per CU while the LDS allows it (n <= ~300 k with the bitmap in LDS): that build is capped at 80 VGPRs
    // (18 dwords spill, none on the per-pop path); the two-per-CU build (<= 128 VGPRs) takes over for larger n
    const size_t granted = round_up(lds, 1280);                       // LDS is granted in 1280-byte granules
    uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / granted));
    if (oneg) per_cu = std::min<uint32_t>(per_cu, 2);                    // that build is not capped at 80 VGPRs
    if (getenv("GS_DENSE_PER_CU")) per_cu = std::max(1, std::min((int)per_cu, atoi(getenv("GS_DENSE_PER_CU"))));
    // per-workgroup global scratch: visited bitmap (vlds = false) or the fine histogram bins (vlds = true)
    const uint32_t scratch_words = (vlds ? dense_nblocks(ix->prm.m) * (HB / 2) : (uint32_t)((ix->n + 31) / 32)) + (split && ix->n > vis_w ? (uint32_t)((ix->n - vis_w + 31) / 32) : 0u);
    const uint32_t capC = 2 * efs + 2 * dcn + maxdeg + 64;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nq, (uint64_t)c->n_cu * per_cu);
    int rc;
    if ((rc = ensure_stats(ix))) return rc;
    ix->stat_wg_in_flight = grid; ix->stat_adj_row_bytes = (uint64_t)4 * maxdeg + 4;      // a pop loads deg0 and the full 2M-id row
    if ((rc = ix->visited.ensure((size_t)4 * scratch_words * c->n_cu * 3))) return rc;
    if ((rc = ix->cbuf.ensure((size_t)16 * capC * c->n_cu * 3))) return rc;
    GS_HIP_CHECK(hipMemsetAsync(ix->counter.p, 0, 8, c->stream));
    IndexDev d = index_dev(ix);
    // (without an accepted-key log the cap_log argument is free: its low bit switches the order-free phase 2 off, GS_DENSE_PHASE2=0, for A/B runs)
    const uint32_t p2_off = (getenv("GS_DENSE_PHASE2") && !atoi(getenv("GS_DENSE_PHASE2"))) ? 1u : 0u;
    unsigned long long *prof = nullptr;
    DevBuf profbuf;
    if (getenv("GS_TRAV_PROFILE")) { if ((rc = profbuf.alloc(128))) return rc; GS_HIP_CHECK(hipMemsetAsync(profbuf.p, 0, 128, c->stream)); prof = profbuf.as<unsigned long long>(); }
    {
    ProfScope ps(c, FAM_SEARCH);
#define GS_LAUNCH_DSEARCH(V, P, O, G) GS_LAUNCH_DSEARCH_S(V, P, O, G, false)
#define GS_LAUNCH_DSEARCH_S(V, P, O, G, SP)                                                                                  \
    do {                                                                                                                  \
        auto kern = k_hnsw_search_dense<V, P, O, G, false, SP>;                                                            \
        GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(DT), lds, c->stream, d, nq, knbn, ef, mat, mat_ld, ix->visited.as<uint32_t>(), scratch_words, \
                           ix->cbuf.as<uint64_t>(), capC, ix->counter.as<unsigned long long>(), ids, dist, count, evals, prof, ix->stats.as<unsigned long long>(),  \
                           (uint64_t *)nullptr, p2_off, 0u, (uint64_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr, vis_w);  \
    } while (0)
    if (split) { if (oneg) GS_LAUNCH_DSEARCH_S(true, false, 4, true, true); else GS_LAUNCH_DSEARCH_S(true, false, 4, false, true); }
    else if (oneg) { if (vlds) GS_LAUNCH_DSEARCH(true, false, 4, true); else GS_LAUNCH_DSEARCH(false, false, 4, true); }
    else if (prof) { if (vlds) GS_LAUNCH_DSEARCH(true, true, 4, false); else GS_LAUNCH_DSEARCH(false, true, 4, false); }
    else if (per_cu >= 3) { if (vlds) GS_LAUNCH_DSEARCH(true, false, 6, false); else GS_LAUNCH_DSEARCH(false, false, 6, false); }
    else { if (vlds) GS_LAUNCH_DSEARCH(true, false, 4, false); else GS_LAUNCH_DSEARCH(false, false, 4, false); }
#undef GS_LAUNCH_DSEARCH
#undef GS_LAUNCH_DSEARCH_S
    }
    GS_HIP_CHECK(hipGetLastError());
    if (prof) {
        unsigned long long h[16];
        GS_HIP_CHECK(hipMemcpyAsync(h, prof, 128, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        fprintf(stderr, "[GS_TRAV_PROFILE] workgroup 0: pops %llu merges %llu | cycles/pop: loads+atomics issue->ballot %.0f, sync1 %.0f, compaction+sync2 %.0f, accept rule+count %.0f | merge cycles/merge %.0f\n",
                h[5], h[6], (double)h[0] / h[5], (double)h[1] / h[5], (double)h[2] / h[5], (double)h[3] / h[5], h[6] ? (double)h[4] / h[6] : 0.0);
        if (h[6]) fprintf(stderr, "[GS_TRAV_PROFILE] of %llu accepting pops: %llu merged into T, %llu folded N into G\n", h[6], h[13], h[14]);
        if (h[6]) fprintf(stderr, "[GS_TRAV_PROFILE] per accepting pop: compaction+sort %.0f, prefetch+trim %.0f, T %.0f, fold+N merge %.0f cycles; accepted keys %.2f\n",
                          (double)h[8] / h[6], (double)h[9] / h[6], (double)h[10] / h[6], (double)h[11] / h[6], (double)h[12] / h[6]);
    }
    return GS_OK;
}

static int search_launch(gs_index *ix, const uint8_t *q_padded_dev, uint64_t nq, uint32_t knbn, uint32_t ef, const uint16_t *mat, uint64_t mat_ld,
                         uint64_t *ids, float *dist, uint32_t *count, uint64_t *evals)
{
    gs_ctx *c = ix->ctx;
    const uint32_t efs = std::max(ef, knbn);
    const uint32_t maxdeg = 2 * ix->prm.max_nb_conn;
    if (mat && maxdeg <= (uint32_t)DT && efs <= 65535u && knbn <= (uint32_t)(TMAXI * DT) && ix->prm.m <= 65535u &&
        dense_lds_bytes(ix->prm.m, knbn, maxdeg, ix->n, false, maxdeg > (uint32_t)DT / 2 ? 512u : (uint32_t)DCN) <= 160 * 1024 - 1024 && !getenv("GS_DENSE_LEGACY"))
        return search_launch_dense(ix, nq, knbn, ef, mat, mat_ld, ids, dist, count, evals);
    const size_t lds = search_lds_bytes(efs, maxdeg);
    const uint32_t vis_words = (uint32_t)((ix->n + 31) / 32);
    uint32_t grid = (uint32_t)std::min<uint64_t>(nq, (uint64_t)c->n_cu);
    GS_HIP_CHECK(hipMemsetAsync(ix->counter.p, 0, 8, c->stream));
    IndexDev d = index_dev(ix);
    ProfScope ps(c, FAM_SEARCH);
#define GS_LAUNCH_SEARCH(K)                                                                                               \
    do {                                                                                                                  \
        auto kern = k_hnsw_search<K>;                                                                                     \
        GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(ST), lds, c->stream, d, q_padded_dev, nq, knbn, ef, mat, mat_ld,       \
                           ix->visited.as<uint32_t>(), vis_words, ix->counter.as<unsigned long long>(), ids, dist, count, evals); \
    } while (0)
    if (ix->ikind == GS_KIND_F32) GS_LAUNCH_SEARCH(GS_KIND_F32);
    else if (ix->ikind == GS_KIND_U32) GS_LAUNCH_SEARCH(GS_KIND_U32);
    else GS_LAUNCH_SEARCH(GS_KIND_U64);
#undef GS_LAUNCH_SEARCH
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// true when search_dev takes the dense strategy for ALL nq queries at once (no gather probe first)
static bool search_goes_dense(const gs_index *ix, uint64_t nq, uint32_t knbn, uint32_t ef)
{
    const uint32_t efs = std::max(ef, knbn);
    const DistMode mode = env_mode();
    if (ix->prm.m > 65535) return false;
    if (mode == MODE_DENSE) return true;
    const bool eligible = mode == MODE_AUTO && ix->n >= 4096;
    return eligible && dense_pays(ix, (double)std::min<uint64_t>(ix->n, efs) / (double)ix->n, nq);
}
static int search_dev(gs_index *ix, const void *q_padded_dev, uint64_t nq, uint32_t knbn, uint32_t ef, uint64_t *ids, float *dist,
                      uint32_t *count, uint64_t *evals)
{
    gs_ctx *c = ix->ctx;
    const uint32_t efs = std::max(ef, knbn);
    const uint32_t maxdeg = 2 * ix->prm.max_nb_conn;
    const size_t lds = search_lds_bytes(efs, maxdeg);
    GS_REQUIRE(maxdeg <= ST, GS_ERR_UNSUPPORTED, "max_nb_conn too large");
    const DistMode mode = env_mode();
    // The sorted-array traversal (k_hnsw_search) keeps its ef best keys in LDS: ef <= ~6700. The dense strategy keeps a histogram instead and takes any
    // ef up to 65535 (hnsw_rs' parallel_search has no limit; gsearch itself asks for 5000, gsearch.rs:893): a larger ef goes that way whatever the cost
    // model says - when the index allows the dense traversal at all
    const bool fits_sorted = lds <= 160 * 1024 - 64 && 2 * (size_t)efs + maxdeg + 64 <= (size_t)SMAXI * ST;
    const bool dense_able = ix->prm.m <= 65535 && maxdeg <= (uint32_t)DT && efs <= 65535u && knbn <= (uint32_t)(TMAXI * DT) && mode != MODE_GATHER && !getenv("GS_DENSE_LEGACY") &&
                            dense_lds_bytes(ix->prm.m, knbn, maxdeg, ix->n, false, maxdeg > (uint32_t)DT / 2 ? 512u : (uint32_t)DCN) <= 160 * 1024 - 1024;
    GS_REQUIRE(fits_sorted || dense_able, GS_ERR_UNSUPPORTED, "ef=%u needs %zu bytes of LDS in the sorted-array traversal (max ~%u with M=%u) and this index / mode does not admit the dense one (m <= 65535, ef <= 65535, GS_DIST_MODE != gather)",
               efs, lds, (unsigned)((160 * 1024 - 64 - 32 * maxdeg - 1024) / 24), ix->prm.max_nb_conn);
    const bool force_dense = !fits_sorted;
    const uint32_t vis_words = (uint32_t)((ix->n + 31) / 32);
    int rc;
    if ((rc = ix->visited.ensure((size_t)4 * vis_words * c->n_cu))) return rc;
    if ((rc = ix->counter.ensure(64))) return rc;
    const uint8_t *q = (const uint8_t *)q_padded_dev;
    uint64_t done = 0;
    DevBuf tmp_evals;
    // a traversal evaluates at least min(n, ef) nodes (R must fill before the stop rule can fire): when that lower bound already
    // makes the dense strategy cheaper there is nothing to probe
    const bool eligible = mode == MODE_AUTO && ix->prm.m <= 65535 && ix->n >= 4096;
    const bool lb_dense = eligible && dense_pays(ix, (double)std::min<uint64_t>(ix->n, efs) / (double)ix->n, nq);
    if (eligible && !lb_dense && ix->search_frac < 0 && nq >= 256 && !force_dense) {
        // probe: the first queries go the gather way and tell which fraction of the graph a traversal evaluates
        const uint64_t np = 128;
        uint64_t *ev = evals;
        if (!ev) { if ((rc = tmp_evals.alloc(8 * np))) return rc; ev = tmp_evals.as<uint64_t>(); }
        if ((rc = search_launch(ix, q, np, knbn, ef, nullptr, 0, ids, dist, count, ev))) return rc;
        std::vector<uint64_t> h(np);
        GS_HIP_CHECK(hipMemcpyAsync(h.data(), ev, 8 * np, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        double sum = 0; for (uint64_t v : h) sum += (double)v;
        ix->search_frac = sum / (double)np / (double)ix->n;
        done = np;
    }
    const uint64_t rest = nq - done;
    // evaluated fraction: measured by an earlier search, else the one the insertions of this index measured (same graph, ef_construction)
    const double frac_known = ix->search_frac >= 0 ? ix->search_frac : ix->insert_frac;
    bool dense = ix->prm.m <= 65535 && (mode == MODE_DENSE || force_dense || lb_dense || (eligible && frac_known >= 0 && rest >= 1 && dense_pays(ix, frac_known, rest)));
    if (!dense) {
        if (rest) {
            uint64_t *ev = evals ? evals + done : nullptr;
            if (!ev && eligible && ix->search_frac < 0) { if ((rc = tmp_evals.alloc(8 * rest))) return rc; ev = tmp_evals.as<uint64_t>(); }
            if ((rc = search_launch(ix, q + done * ix->stride, rest, knbn, ef, nullptr, 0, ids + done * knbn, dist + done * knbn, count ? count + done : nullptr, ev))) return rc;
            if (eligible && ix->search_frac < 0 && ev) {       // learn the evaluated fraction from this call (small batches never probe)
                const uint64_t ns = std::min<uint64_t>(rest, 128);
                std::vector<uint64_t> h(ns);
                GS_HIP_CHECK(hipMemcpyAsync(h.data(), ev, 8 * ns, hipMemcpyDeviceToHost, c->stream));
                GS_HIP_CHECK(hipStreamSynchronize(c->stream));
                double sum = 0; for (uint64_t v : h) sum += (double)v;
                ix->search_frac = sum / (double)ns / (double)ix->n;
            }
        }
        return GS_OK;
    }
    const uint64_t ld = round_up(ix->n, 8);
    const bool join = use_join(ix);
    // one traversal launch should cover the whole request (every launch ends with a tail in which most CUs idle while the last queries
    // finish), so the count matrix is sized for all queries when HBM allows: a quarter of the free memory, at most 32 GB (10 000 queries
    // x 300 k nodes x 2 B = 6 GB); the join fills it in chunks of its own maximum inside dense_counts
    uint64_t budget = (uint64_t)4 << 30;
    {
        size_t fr = 0, tot = 0;
        // (round 5: 0.7 of what is free or already held by the matrix, not a quarter of the free memory plus what is held - at 1 M nodes that grew the buffer
        // from call to call, 10.8 -> 15.6 -> 19.2 -> 20 GB, and every growth is a hipFree + hipMalloc of ~0.45 s: profiles/r05_request_1M_matrix_realloc.log)
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget = std::max<uint64_t>(budget, std::min<uint64_t>((uint64_t)32 << 30, (uint64_t)(0.7 * (double)((uint64_t)fr + ix->mat.bytes))));
    }
    uint64_t QB = budget / (2 * ld);
    QB = std::max<uint64_t>(128, QB / 128 * 128);
    if (getenv("GS_SEARCH_QB")) QB = std::min<uint64_t>(QB, (uint64_t)std::max(1, atoi(getenv("GS_SEARCH_QB"))));
    QB = std::min<uint64_t>(QB, rest);
    const bool sv = getenv("GS_SEARCH_VERBOSE") != nullptr;
    const auto sv_t0 = std::chrono::steady_clock::now();
    if (sv) { size_t fr = 0, tot = 0; (void)hipMemGetInfo(&fr, &tot); fprintf(stderr, "[GS_SEARCH] nq %llu n %llu: QB %llu, count matrix %.2f GB wanted, %.2f GB held, device free %.2f GB\n", (unsigned long long)nq, (unsigned long long)ix->n, (unsigned long long)QB, 2.0 * QB * ld / 1e9, ix->mat.bytes / 1e9, fr / 1e9); }
    if (ix->mat.bytes < (size_t)2 * QB * ld && (rc = alloc_or_evict(ix, ix->mat, (size_t)2 * QB * ld))) return rc;
    if (sv) fprintf(stderr, "[GS_SEARCH] count matrix ready after %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sv_t0).count());
    if (join && (rc = ensure_cols(ix, ix->n))) return rc;
    if (ix->pipe3 && join && done == 0 && QB == nq) {
        // three stages, batch by batch (the fused request): count matrix of batch b on this stream, its traversal on `tstream` - beside the count matrix of
        // batch b + 1 here and the sketches the caller queued on its own stream. The traversal scratch (visited / candidate arrays / query counter) is per
        // index: launches on tstream follow one another.
        const uint64_t maxq = match_join_max_queries(), parts = (nq + maxq - 1) / maxq, jq = (nq + parts - 1) / parts;
        // (leaving this scope - also through a failing dense_counts / search_launch - waits for what is queued on tstream: traversal kernels must not go on
        // writing ids / dist / evals, nor use ix->visited / cbuf, after the call has returned an error; on success the wait is an event on the main stream below)
        struct SGuard { gs_ctx *c; hipStream_t main; hipStream_t t; bool ok = false; ~SGuard() { c->stream = main; if (!ok) (void)hipStreamSynchronize(t); } } sg{c, c->stream, ix->tstream};
        for (uint64_t q0 = 0; q0 < nq; q0 += jq) {
            const uint64_t nb = std::min(jq, nq - q0);
            if ((rc = dense_counts(ix, q + q0 * ix->stride, nb, ix->n, ix->mat.as<uint16_t>() + q0 * ld, ld))) return rc;
            GS_HIP_CHECK(hipEventRecord(ix->tev, c->stream));
            GS_HIP_CHECK(hipStreamWaitEvent(ix->tstream, ix->tev, 0));
            c->stream = ix->tstream;
            rc = search_launch(ix, q + q0 * ix->stride, nb, knbn, ef, ix->mat.as<uint16_t>() + q0 * ld, ld, ids + q0 * knbn, dist + q0 * knbn, count ? count + q0 : nullptr,
                               evals ? evals + q0 : nullptr);
            c->stream = sg.main;
            if (rc) return rc;
        }
        GS_HIP_CHECK(hipEventRecord(ix->tev_done, ix->tstream));
        GS_HIP_CHECK(hipStreamWaitEvent(c->stream, ix->tev_done, 0));
        sg.ok = true;
        return GS_OK;
    }
    for (uint64_t q0 = done; q0 < nq; q0 += QB) {
        const uint64_t nb = std::min(QB, nq - q0);
        if ((rc = dense_counts(ix, q + q0 * ix->stride, nb, ix->n, ix->mat.as<uint16_t>(), ld))) return rc;
        if (sv) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "[GS_SEARCH] count matrix of %llu queries done at %.1f ms\n", (unsigned long long)nb, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sv_t0).count()); }
        if ((rc = search_launch(ix, q + q0 * ix->stride, nb, knbn, ef, ix->mat.as<uint16_t>(), ld, ids + q0 * knbn, dist + q0 * knbn,
                                count ? count + q0 : nullptr, evals ? evals + q0 : nullptr))) return rc;
        if (sv) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "[GS_SEARCH] traversal done at %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - sv_t0).count()); }
    }
    return GS_OK;
}

}  // namespace gs

extern "C" {

int gs_index_create(gs_ctx *c, const gs_index_params *p, gs_index **out)
{
    GS_REQUIRE(c && p && out, GS_ERR_INVALID, "null argument");
    GS_REQUIRE(p->kind == GS_KIND_F32 || p->kind == GS_KIND_U32 || p->kind == GS_KIND_U64 || p->kind == GS_KIND_U16, GS_ERR_UNSUPPORTED, "signature kind %d not supported by the index", p->kind);
    GS_REQUIRE(p->m >= 1, GS_ERR_INVALID, "m must be positive");
    GS_REQUIRE(p->max_nb_conn >= 2 && p->max_nb_conn <= 255, GS_ERR_INVALID, "max_nb_conn must be in 2..255 (gsearch.rs:268)");
    GS_REQUIRE(p->max_layer >= 1 && p->max_layer <= 16, GS_ERR_INVALID, "max_layer must be in 1..16");
    GS_REQUIRE(p->ef_construction >= 1, GS_ERR_INVALID, "ef_construction must be positive");
    gs_index *ix = new gs_index();
    ix->ctx = c; ix->prm = *p;
    if (ix->prm.insert_batch == 0) ix->prm.insert_batch = 64;
    ix->ikind = p->kind == GS_KIND_U16 ? GS_KIND_U32 : p->kind;
    ix->esz = gs::kind_bytes(ix->ikind);
    ix->rowbytes = ix->esz * p->m;
    ix->user_rowbytes = gs::kind_bytes(p->kind) * (size_t)p->m;
    ix->stride = gs::round_up(ix->rowbytes, 256);
    ix->nchunks = (uint32_t)((ix->rowbytes + 15) / 16);
    *out = ix;
    return GS_OK;
}
void gs_index_destroy(gs_index *ix)
{
    if (!ix) return;
    GS_CTX_LOCK(ix->ctx);
    (void)hipSetDevice(ix->ctx->device);
    (void)hipStreamSynchronize(ix->ctx->stream);
    delete ix;
}
uint64_t gs_index_nb_point(const gs_index *ix) { return ix ? ix->n : 0; }
int gs_index_search_stats(gs_index *ix, uint64_t out[8], int reset)
{
    GS_REQUIRE(ix && out, GS_ERR_INVALID, "null argument");
    gs_ctx *c = ix->ctx;
    GS_CTX_LOCK(c);
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (ix->stats.p) {
        GS_HIP_CHECK(hipSetDevice(c->device));
        unsigned long long h[16];
        GS_HIP_CHECK(hipMemcpyAsync(h, ix->stats.p, 128, hipMemcpyDeviceToHost, c->stream));
        if (reset) GS_HIP_CHECK(hipMemsetAsync(ix->stats.p, 0, 128, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[5] = h[5]; out[6] = h[6]; out[7] = h[7];
        if (getenv("GS_TRAV_PHASES") && h[12])      // shader-clock cycles of workgroup 0 per query: tau scan, phase 1, phase 2 (list + Bloom build, drain, T merge)
            fprintf(stderr, "[GS_TRAV_PHASES] workgroup 0, %llu queries: cycles per query tau scan %.0f, phase 1 %.0f, phase 2 %.0f, output %.0f\n", h[12],
                    (double)h[8] / h[12], (double)h[9] / h[12], (double)h[10] / h[12], (double)h[11] / h[12]),
            fprintf(stderr, "[GS_TRAV_PHASES]   inside phase 2: work list + Bloom build %.0f, drain %.0f, merge into T %.0f\n", (double)h[13] / h[12], (double)h[14] / h[12],
                    ((double)h[10] - (double)h[13] - (double)h[14]) / h[12]),
            fprintf(stderr, "[GS_TRAV_PHASES]   phase 1 front, all workgroups: refills %llu, compactions %llu, selection rounds beyond the first %llu\n", h[15] & 0xFFFFFFull, (h[15] >> 24) & 0xFFFFull, h[15] >> 40);
    }
    out[3] = ix->stat_wg_in_flight; out[4] = ix->stat_adj_row_bytes;
    return GS_OK;
}
int gs_index_release_build_scratch(gs_index *ix)
{
    GS_REQUIRE(ix, GS_ERR_INVALID, "gs_index_release_build_scratch: null index");
    GS_CTX_LOCK(ix->ctx);
    const uint64_t budget = ix->pair_cache_budget;
    gs::drop_pair_cache(ix);                                       // (waits for the streams that may still read the slabs)
    ix->pair_cache_budget = budget;                                // later inserts may cache their own batches again
    return GS_OK;
}
uint64_t gs_index_insert_evals(const gs_index *ix) { return ix ? ix->insert_evals : 0; }
int gs_index_get_params(const gs_index *ix, gs_index_params *out)
{
    GS_REQUIRE(ix && out, GS_ERR_INVALID, "null argument");
    *out = ix->prm;
    return GS_OK;
}

int gs_index_import(gs_index *ix, const void *sigs, uint64_t n, const uint8_t *levels, int64_t entry, const uint32_t *deg0,
                    const uint32_t *nbr0, const uint32_t *cnt0, const int32_t *upidx, uint64_t n_upper, const uint32_t *degU,
                    const uint32_t *nbrU, const uint32_t *cntU)
{
    GS_REQUIRE(ix && ix->n == 0, GS_ERR_STATE, "import needs an empty index");
    GS_REQUIRE(n > 0 && sigs && levels && deg0 && nbr0 && cnt0 && upidx, GS_ERR_INVALID, "null argument");
    GS_REQUIRE(entry >= 0 && (uint64_t)entry < n, GS_ERR_INVALID, "entry point out of range");
    GS_REQUIRE(n_upper == 0 || (degU && nbrU && cntU), GS_ERR_INVALID, "null upper-layer arrays");
    gs_ctx *c = ix->ctx;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t M = ix->prm.max_nb_conn, ML = ix->prm.max_layer;
    int top = 0;
    for (uint64_t i = 0; i < n; i++) {
        GS_REQUIRE(levels[i] < ML, GS_ERR_INVALID, "level of node %llu out of range", (unsigned long long)i);
        GS_REQUIRE(deg0[i] <= 2 * M, GS_ERR_INVALID, "degree of node %llu out of range", (unsigned long long)i);
        for (uint32_t t = 0; t < deg0[i]; t++) GS_REQUIRE(nbr0[i * 2 * M + t] < n, GS_ERR_INVALID, "neighbour id out of range");
        GS_REQUIRE((levels[i] > 0) == (upidx[i] >= 0), GS_ERR_INVALID, "upidx inconsistent with level at node %llu", (unsigned long long)i);
        if (upidx[i] >= 0) {
            GS_REQUIRE((uint64_t)upidx[i] < n_upper, GS_ERR_INVALID, "upidx of node %llu out of range", (unsigned long long)i);
            for (uint32_t l = 1; l <= levels[i]; l++) {
                const uint64_t o = (uint64_t)upidx[i] * ML + (l - 1);
                GS_REQUIRE(degU[o] <= M, GS_ERR_INVALID, "degree of node %llu at layer %u out of range", (unsigned long long)i, l);
                for (uint32_t t = 0; t < degU[o]; t++) {
                    const uint32_t e = nbrU[o * M + t];
                    GS_REQUIRE(e < n && levels[e] >= l, GS_ERR_INVALID, "node %llu links to %u at layer %u, which does not reach that layer", (unsigned long long)i, e, l);
                }
            }
        }
        if (levels[i] > top) top = levels[i];
    }
    GS_REQUIRE(levels[entry] == top, GS_ERR_INVALID, "entry point is not on the top layer");
    int rc = gs::index_reserve(ix, n, n_upper);
    if (rc) return rc;
    if ((rc = gs::upload_user_rows(ix, ix->data.p, sigs, n, hipMemcpyHostToDevice))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(ix->levels.p, levels, n, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(ix->deg0.p, deg0, 4 * n, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(ix->nbr0.p, nbr0, (size_t)8 * M * n, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(ix->cnt0.p, cnt0, (size_t)8 * M * n, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(ix->upidx.p, upidx, 4 * n, hipMemcpyHostToDevice, c->stream));
    if (n_upper) {
        GS_HIP_CHECK(hipMemcpyAsync(ix->degU.p, degU, (size_t)4 * ML * n_upper, hipMemcpyHostToDevice, c->stream));
        GS_HIP_CHECK(hipMemcpyAsync(ix->nbrU.p, nbrU, (size_t)4 * ML * M * n_upper, hipMemcpyHostToDevice, c->stream));
        GS_HIP_CHECK(hipMemcpyAsync(ix->cntU.p, cntU, (size_t)4 * ML * M * n_upper, hipMemcpyHostToDevice, c->stream));
    }
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    ix->n = n; ix->n_upper = n_upper; ix->entry = entry; ix->top = top;
    gs::ids_append(ix, nullptr, levels, n);
    return GS_OK;
}

int gs_index_export(gs_index *ix, uint8_t *levels, int64_t *entry, uint32_t *deg0, uint32_t *nbr0, uint32_t *cnt0, int32_t *upidx,
                    uint64_t *n_upper, uint32_t *degU, uint32_t *nbrU, uint32_t *cntU)
{
    GS_REQUIRE(ix, GS_ERR_INVALID, "null index");
    gs_ctx *c = ix->ctx;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t M = ix->prm.max_nb_conn, ML = ix->prm.max_layer;
    const uint64_t n = ix->n, U = ix->n_upper;
    if (entry) *entry = ix->entry;
    if (n_upper) *n_upper = U;
    if (n) {
        if (levels) GS_HIP_CHECK(hipMemcpyAsync(levels, ix->levels.p, n, hipMemcpyDeviceToHost, c->stream));
        if (deg0) GS_HIP_CHECK(hipMemcpyAsync(deg0, ix->deg0.p, 4 * n, hipMemcpyDeviceToHost, c->stream));
        if (nbr0) GS_HIP_CHECK(hipMemcpyAsync(nbr0, ix->nbr0.p, (size_t)8 * M * n, hipMemcpyDeviceToHost, c->stream));
        if (cnt0) GS_HIP_CHECK(hipMemcpyAsync(cnt0, ix->cnt0.p, (size_t)8 * M * n, hipMemcpyDeviceToHost, c->stream));
        if (upidx) GS_HIP_CHECK(hipMemcpyAsync(upidx, ix->upidx.p, 4 * n, hipMemcpyDeviceToHost, c->stream));
    }
    if (U) {
        if (degU) GS_HIP_CHECK(hipMemcpyAsync(degU, ix->degU.p, (size_t)4 * ML * U, hipMemcpyDeviceToHost, c->stream));
        if (nbrU) GS_HIP_CHECK(hipMemcpyAsync(nbrU, ix->nbrU.p, (size_t)4 * ML * M * U, hipMemcpyDeviceToHost, c->stream));
        if (cntU) GS_HIP_CHECK(hipMemcpyAsync(cntU, ix->cntU.p, (size_t)4 * ML * M * U, hipMemcpyDeviceToHost, c->stream));
    }
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

int gs_index_get_data(gs_index *ix, uint64_t first, uint64_t n, void *out)
{
    GS_REQUIRE(ix && out && first + n <= ix->n, GS_ERR_INVALID, "bad range");
    if (n == 0) return GS_OK;
    GS_CTX_LOCK(ix->ctx);
    return gs::download_user_rows(ix, first, n, out);
}

}  // extern "C"
namespace gs {
// bookkeeping of the ids / PointIds of `n` new nodes whose levels are lv[0..n) (ids == nullptr: they continue the caller's 0.. numbering)
static void ids_append(gs_index *ix, const uint64_t *ids, const uint8_t *lv, uint64_t n)
{
    const uint64_t first = ix->pid_rank.size();
    //  ids that continue nb_point.. in order - gsearch's own case, dnasketch.rs:429-433, also when a host always goes through the `_ids`
    // entry point - ARE the node numbers: `origin` stays empty, so searches launch no k_map_ids and gs_index_save keeps the GSAMDIX1 form
    if (ids && ix->origin.empty()) {
        bool identity = true;
        for (uint64_t i = 0; i < n && identity; i++) identity = ids[i] == first + i;
        if (identity) ids = nullptr;
    }
    if (ids && ix->origin.empty()) { ix->origin.resize(first); for (uint64_t i = 0; i < first; i++) ix->origin[i] = i; }
    for (uint64_t i = 0; i < n; i++) {
        ix->pid_rank.push_back((int32_t)ix->level_count[lv[i] <= 16 ? lv[i] : 16]++);
        if (ids) ix->origin.push_back(ids[i]);
        else if (!ix->origin.empty()) ix->origin.push_back(first + i);
    }
}
__global__ void k_pid_of(const uint64_t *__restrict__ ids, uint64_t total, const uint8_t *__restrict__ levels, const int32_t *__restrict__ rank, uint8_t *__restrict__ layer_out,
                         int32_t *__restrict__ rank_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t id = ids[i];
    const bool ok = id != ~(uint64_t)0;
    layer_out[i] = ok ? levels[id] : (uint8_t)0xFF;
    rank_out[i] = ok ? rank[id] : -1;
}
__global__ void k_map_ids(uint64_t *__restrict__ ids, uint64_t total, const uint64_t *__restrict__ origin)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total && ids[i] != ~(uint64_t)0) ids[i] = origin[ids[i]];
}
// after a search on the device: PointIds of the answers (optional), then internal node numbers -> the caller's ids
static int finish_ids(gs_index *ix, uint64_t *ids_dev, uint64_t total, uint8_t *pid_layer_dev, int32_t *pid_rank_dev)
{
    gs_ctx *c = ix->ctx;
    int rc;
    if (pid_layer_dev && pid_rank_dev) {
        if (ix->pid_rank_d_n != ix->n) {
            if ((rc = ix->pid_rank_d.ensure(4 * (size_t)ix->cap))) return rc;
            GS_HIP_CHECK(hipMemcpyAsync(ix->pid_rank_d.p, ix->pid_rank.data(), 4 * (size_t)ix->n, hipMemcpyHostToDevice, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            ix->pid_rank_d_n = ix->n;
        }
        hipLaunchKernelGGL(k_pid_of, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, c->stream, ids_dev, total, ix->levels.as<uint8_t>(), ix->pid_rank_d.as<int32_t>(), pid_layer_dev, pid_rank_dev);
        GS_HIP_CHECK(hipGetLastError());
    }
    if (!ix->origin.empty()) {
        if (ix->origin_d_n != ix->n) {
            if ((rc = ix->origin_d.ensure(8 * (size_t)ix->cap))) return rc;
            GS_HIP_CHECK(hipMemcpyAsync(ix->origin_d.p, ix->origin.data(), 8 * (size_t)ix->n, hipMemcpyHostToDevice, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            ix->origin_d_n = ix->n;
        }
        hipLaunchKernelGGL(k_map_ids, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, c->stream, ids_dev, total, ix->origin_d.as<uint64_t>());
        GS_HIP_CHECK(hipGetLastError());
    }
    return GS_OK;
}
}  // namespace gs
extern "C" {

static int search_common(gs_index *ix, const void *queries, bool on_dev, uint64_t nq, uint32_t knbn, uint32_t ef, uint64_t *ids, float *dist,
                         uint32_t *count, uint64_t *evals, uint8_t *pid_layer = nullptr, int32_t *pid_rank = nullptr)
{
    GS_REQUIRE((pid_layer == nullptr) == (pid_rank == nullptr), GS_ERR_INVALID, "pid_layer_out and pid_rank_out go together");
    GS_REQUIRE(ix, GS_ERR_INVALID, "null index");
    GS_REQUIRE(knbn >= 1 && ef >= 1, GS_ERR_INVALID, "knbn and ef must be positive");
    if (nq == 0) return GS_OK;
    GS_REQUIRE(queries && ids && dist, GS_ERR_INVALID, "null argument");
    GS_REQUIRE(ix->n > 0, GS_ERR_STATE, "search on an empty index");
    gs_ctx *c = ix->ctx;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    gs::PoolBuf dq(c, 32), dids(c, 33), ddist(c, 34), dcount(c, 35), devals(c, 36);
    int rc;
    if ((rc = dq.alloc(ix->stride * nq))) return rc;
    if ((rc = gs::upload_user_rows(ix, dq.p, queries, nq, on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice))) return rc;
    if (on_dev) {
        if ((rc = gs::search_dev(ix, dq.p, nq, knbn, ef, ids, dist, count, evals))) return rc;
        if ((rc = gs::finish_ids(ix, ids, nq * knbn, pid_layer, pid_rank))) return rc;
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        return GS_OK;
    }
    gs::PoolBuf dpl(c, 40), dpr(c, 41);
    if ((rc = dids.alloc(8 * nq * knbn))) return rc;
    if ((rc = ddist.alloc(4 * nq * knbn))) return rc;
    if ((rc = dcount.alloc(4 * nq))) return rc;
    if ((rc = devals.alloc(8 * nq))) return rc;
    if (pid_layer && ((rc = dpl.alloc(nq * knbn)) || (rc = dpr.alloc(4 * nq * knbn)))) return rc;
    if ((rc = gs::search_dev(ix, dq.p, nq, knbn, ef, dids.as<uint64_t>(), ddist.as<float>(), dcount.as<uint32_t>(), devals.as<uint64_t>()))) return rc;
    if ((rc = gs::finish_ids(ix, dids.as<uint64_t>(), nq * knbn, pid_layer ? dpl.as<uint8_t>() : nullptr, pid_layer ? dpr.as<int32_t>() : nullptr))) return rc;
    if (pid_layer) {
        GS_HIP_CHECK(hipMemcpyAsync(pid_layer, dpl.p, nq * knbn, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipMemcpyAsync(pid_rank, dpr.p, 4 * nq * knbn, hipMemcpyDeviceToHost, c->stream));
    }
    GS_HIP_CHECK(hipMemcpyAsync(ids, dids.p, 8 * nq * knbn, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dist, ddist.p, 4 * nq * knbn, hipMemcpyDeviceToHost, c->stream));
    if (count) GS_HIP_CHECK(hipMemcpyAsync(count, dcount.p, 4 * nq, hipMemcpyDeviceToHost, c->stream));
    if (evals) GS_HIP_CHECK(hipMemcpyAsync(evals, devals.p, 8 * nq, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

int gs_index_count_matrix(gs_index *ix, const void *queries, uint64_t nq, uint16_t *counts_out)
{
    GS_REQUIRE(ix, GS_ERR_INVALID, "null index");
    if (nq == 0) return GS_OK;
    GS_REQUIRE(queries && counts_out, GS_ERR_INVALID, "null argument");
    GS_REQUIRE(ix->n > 0, GS_ERR_STATE, "count matrix of an empty index");
    GS_REQUIRE(ix->prm.m <= 65535, GS_ERR_UNSUPPORTED, "16-bit counts need m <= 65535");
    gs_ctx *c = ix->ctx;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    gs::PoolBuf dq(c, 32);
    int rc;
    if ((rc = dq.alloc(ix->stride * nq))) return rc;
    if ((rc = gs::upload_user_rows(ix, dq.p, queries, nq, hipMemcpyHostToDevice))) return rc;
    const uint64_t ld = gs::round_up(ix->n, 8);
    if (ix->mat.bytes < (size_t)2 * nq * ld && (rc = gs::alloc_or_evict(ix, ix->mat, (size_t)2 * nq * ld))) return rc;
    if ((rc = gs::dense_counts(ix, dq.as<uint8_t>(), nq, ix->n, ix->mat.as<uint16_t>(), ld))) return rc;
    GS_HIP_CHECK(hipMemcpy2DAsync(counts_out, 2 * ix->n, ix->mat.p, 2 * ld, 2 * ix->n, nq, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

int gs_index_parallel_search(gs_index *ix, const void *queries, uint64_t nq, uint32_t knbn, uint32_t ef, uint64_t *ids, float *dist,
                             uint32_t *count, uint64_t *evals)
{
    return search_common(ix, queries, false, nq, knbn, ef, ids, dist, count, evals);
}
int gs_index_parallel_search_dev(gs_index *ix, const void *queries_dev, uint64_t nq, uint32_t knbn, uint32_t ef, uint64_t *ids, float *dist,
                                 uint32_t *count, uint64_t *evals)
{
    return search_common(ix, queries_dev, true, nq, knbn, ef, ids, dist, count, evals);
}
int gs_index_parallel_search_pid(gs_index *ix, const void *queries, uint64_t nq, uint32_t knbn, uint32_t ef, uint64_t *ids, float *dist, uint32_t *count, uint64_t *evals,
                                 uint8_t *pid_layer, int32_t *pid_rank)
{
    return search_common(ix, queries, false, nq, knbn, ef, ids, dist, count, evals, pid_layer, pid_rank);
}
int gs_index_parallel_search_pid_dev(gs_index *ix, const void *queries_dev, uint64_t nq, uint32_t knbn, uint32_t ef, uint64_t *ids, float *dist, uint32_t *count,
                                     uint64_t *evals, uint8_t *pid_layer, int32_t *pid_rank)
{
    return search_common(ix, queries_dev, true, nq, knbn, ef, ids, dist, count, evals, pid_layer, pid_rank);
}
/* sketch the request genomes, then ONE parallel_search (sketch_and_request_dir_compressedkmer, /root/reference/src/dna/dnarequest.rs:240-360) as one call:
 * with the dense strategy and more than one join batch the sketch of batch b + 1 runs on a second stream beside the count matrix of batch b. */
int gs_index_sketch_and_search_dev(gs_index *ix, const gs_sketch_params *p, const void *seq_dev, uint64_t seq_bytes, const uint64_t *rec_start_dev, const uint64_t *rec_len_dev,
                                   uint64_t n_rec, const uint64_t *genome_rec_off_dev, uint64_t n_genomes, void *sig_out_dev, uint32_t knbn, uint32_t ef, uint64_t *ids, float *dist,
                                   uint32_t *count, uint64_t *evals)
{
    GS_REQUIRE(ix && p, GS_ERR_INVALID, "null argument");
    GS_REQUIRE(knbn >= 1 && ef >= 1, GS_ERR_INVALID, "knbn and ef must be positive");
    if (n_genomes == 0) return GS_OK;
    GS_REQUIRE(seq_dev && rec_start_dev && rec_len_dev && genome_rec_off_dev && ids && dist, GS_ERR_INVALID, "null argument");
    GS_REQUIRE(ix->n > 0, GS_ERR_STATE, "search on an empty index");
    int rc = gs_check_params(p);
    if (rc) return rc;
    GS_REQUIRE(gs_sig_kind(p) == ix->prm.kind && p->sketch_size == ix->prm.m, GS_ERR_INVALID, "the sketcher's signatures (kind %d, length %u) are not the index's (kind %d, length %u)",
               gs_sig_kind(p), p->sketch_size, ix->prm.kind, ix->prm.m);
    gs_ctx *c = ix->ctx;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    const uint64_t nq = n_genomes;
    gs::PoolBuf dq(c, 32), sigbuf(c, 60);
    if ((rc = dq.alloc(ix->stride * nq))) return rc;
    uint8_t *sig = (uint8_t *)sig_out_dev;
    if (!sig) { if ((rc = sigbuf.alloc(ix->user_rowbytes * nq))) return rc; sig = (uint8_t *)sigbuf.p; }
    const uint64_t maxq = gs::match_join_max_queries(), parts = (nq + maxq - 1) / maxq, jq = (nq + parts - 1) / parts;
    const char *pe = getenv("GS_REQUEST_PIPELINE");
    const bool pipe = !(pe && !atoi(pe)) && parts >= 2 && (p->algo == GS_ALGO_OPTDENS || p->algo == GS_ALGO_REVOPTDENS) && gs::use_join(ix) &&
                      gs::search_goes_dense(ix, nq, knbn, ef);
    const bool pipe3 = pipe && pe && atoi(pe) == 3;
    if (!pipe) {
        if ((rc = gs::sketch_dev_impl(c, p, seq_dev, seq_bytes, rec_start_dev, rec_len_dev, n_rec, genome_rec_off_dev, nq, sig, true))) return rc;
        if ((rc = gs::upload_user_rows(ix, dq.p, sig, nq, hipMemcpyDeviceToDevice))) return rc;
        if ((rc = gs::search_dev(ix, dq.p, nq, knbn, ef, ids, dist, count, evals))) return rc;
        if ((rc = gs::finish_ids(ix, ids, nq * knbn, nullptr, nullptr))) return rc;
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        return GS_OK;
    }
    if (!c->child && (rc = gs_ctx_create(&c->child, c->device, nullptr))) return rc;
    gs_ctx *w = c->child;
    std::vector<hipEvent_t> ev(parts, nullptr);
    std::vector<uint8_t> waited(parts, 0);
    hipEvent_t ev_in = nullptr;
    // every exit goes through here: events destroyed, the worker's pending profile events handed to this context's accounts, its profile flag cleared
    auto cleanup = [&]() {
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
        if (ev_in) (void)hipEventDestroy(ev_in);
        for (int f = 0; f < gs::FAM_COUNT; f++) {
            auto &src = w->prof[f].pending;
            c->prof[f].pending.insert(c->prof[f].pending.end(), src.begin(), src.end());
            src.clear();
        }
        w->profile = false;
    };
#define GS_FUSED_FAIL(code) do { const int rc_ = (code); (void)hipStreamSynchronize(w->stream); (void)hipStreamSynchronize(c->stream); ix->feed = nullptr; cleanup(); return rc_; } while (0)
    // the sketches read what the caller produced on this context's stream
    if (hipEventCreateWithFlags(&ev_in, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev_in, c->stream) != hipSuccess || hipStreamWaitEvent(w->stream, ev_in, 0) != hipSuccess) {
        gs::set_error("event setup failed"); GS_FUSED_FAIL(GS_ERR_HIP);
    }
    w->profile = c->profile;
    if (pipe3) {
        // launches shaped to SHARE a compute unit: a sketch workgroup that asks for more than half of the LDS stays alone of its kind on its CU and leaves
        // room for a traversal workgroup (53.7 kB at 300 k nodes) beside it; the traversal stream gets the highest priority - its workgroups are the
        // latency-bound ones -, the sketch stream the lowest
        if (!ix->tstream) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (hipStreamCreateWithPriority(&ix->tstream, hipStreamNonBlocking, hi) != hipSuccess || hipEventCreateWithFlags(&ix->tev, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ix->tev_done, hipEventDisableTiming) != hipSuccess) { gs::set_error("stream setup failed"); GS_FUSED_FAIL(GS_ERR_HIP); }
        }
        const char *sl = getenv("GS_PIPE_SKETCH_LDS");
        w->sketch_min_lds = sl ? (uint32_t)atoi(sl) : 100u * 1024u;
        ix->pipe3 = true;
    }
    struct P3Guard { gs_index *ix; gs_ctx *w; ~P3Guard() { ix->pipe3 = false; w->sketch_min_lds = 0; } } p3guard{ix, w};
    for (uint64_t b = 0; b < parts; b++) {
        const uint64_t g0 = b * jq, nb = std::min<uint64_t>(jq, nq - g0);
        // (seq_bytes only feeds the launch heuristics: this batch's share of it)
        if ((rc = gs::sketch_dev_impl(w, p, seq_dev, std::max<uint64_t>(seq_bytes / nq * nb, 64), rec_start_dev, rec_len_dev, n_rec, genome_rec_off_dev + g0, nb, sig + g0 * ix->user_rowbytes, false))) GS_FUSED_FAIL(rc);
        if (hipEventCreateWithFlags(&ev[b], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[b], w->stream) != hipSuccess) { gs::set_error("event setup failed"); GS_FUSED_FAIL(GS_ERR_HIP); }
    }
    std::function<int(uint64_t, uint64_t)> feed = [&](uint64_t q0, uint64_t nb) -> int {
        for (uint64_t b = q0 / jq; b < parts && b * jq < q0 + nb; b++)
            if (!waited[b]) { GS_HIP_CHECK(hipStreamWaitEvent(c->stream, ev[b], 0)); waited[b] = 1; }
        return gs::upload_user_rows(ix, dq.as<uint8_t>() + q0 * ix->stride, sig + q0 * ix->user_rowbytes, nb, hipMemcpyDeviceToDevice);
    };
    ix->feed = &feed; ix->feed_base = dq.as<uint8_t>();
    rc = gs::search_dev(ix, dq.p, nq, knbn, ef, ids, dist, count, evals);
    ix->feed = nullptr;
    if (rc) GS_FUSED_FAIL(rc);
    if ((rc = gs::finish_ids(ix, ids, nq * knbn, nullptr, nullptr))) GS_FUSED_FAIL(rc);
    if (hipStreamSynchronize(w->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) { gs::set_error("stream synchronisation failed after the fused request"); GS_FUSED_FAIL(GS_ERR_HIP); }
    cleanup();
#undef GS_FUSED_FAIL
    return GS_OK;
}
int gs_index_set_ids(gs_index *ix, const uint64_t *ids, uint64_t n)
{
    GS_REQUIRE(ix && ids, GS_ERR_INVALID, "null argument");
    GS_REQUIRE(n == ix->n, GS_ERR_INVALID, "gs_index_set_ids: %llu ids for %llu points", (unsigned long long)n, (unsigned long long)ix->n);
    GS_CTX_LOCK(ix->ctx);
    bool identity = true;
    for (uint64_t i = 0; i < n && identity; i++) identity = ids[i] == i;
    if (identity) ix->origin.clear(); else ix->origin.assign(ids, ids + n);
    ix->origin_d_n = 0;
    return GS_OK;
}
int gs_index_get_ids(gs_index *ix, uint64_t first, uint64_t n, uint64_t *ids_out)
{
    GS_REQUIRE(ix && (ids_out || n == 0) && first + n <= ix->n, GS_ERR_INVALID, "bad range");
    GS_CTX_LOCK(ix->ctx);
    for (uint64_t i = 0; i < n; i++) ids_out[i] = ix->origin.empty() ? first + i : ix->origin[first + i];
    return GS_OK;
}

static int gen_level_host(const gs_index *ix, uint64_t id)
{
    // SPEC 5 level generator (hnsw_rs LayerGenerator restated; level scale = f / ln(M), dnasketch.rs:141)
    const double scale = ix->prm.scale_modify / log((double)ix->prm.max_nb_conn);
    gs::Rng g; g.seed(ix->prm.seed ^ gs::fx64(id));
    const double u = g.u64f();
    const uint64_t ML = ix->prm.max_layer, zone = gs::uint_zone(ML);
    if (u == 0.0) return (int)gs::rng_uint(g, ML, zone);
    int l = (int)floor(-log(u) * scale);
    if (l >= (int)ML) l = (int)gs::rng_uint(g, ML, zone);
    return l;
}

// Layer-0 search of an insert batch through the dense traversal kernel: queries = the batch's rows (their counts
// against every node are in `mat`), ef = ef_construction, one accepted-key log per resident workgroup; points of level > 0 (their
// entry point comes out of an ef-search on layer 1, not of the greedy descent) are skipped and keep the sorted-array search inside
// k_hnsw_plan. Leaves the device pointers of W (efc keys per point, sorted), |W| (0xFFFFFFFF = not done) and the evaluation counts.
namespace gs {
// sparse pair rows of `nrows` nodes a0.. from their count rows (k_sparse_fill). Lists and level bitmaps come from a bump allocator the KERNEL
// runs over the index's arena (only it knows how long a list is and which rows need a bitmap). The host keeps the worst case of every launch
// backed by memory - every row a full list and a bitmap - and looks at the real fill level (one 64-byte read back) only when that bound runs past
// what is mapped; then more of the arena is mapped (1 GB steps, nothing moves). When the arena (GS_SPARSE_ARENA_GB, default: what the device
// keeps free next to the signatures and their column copy at the declared capacity) or the device is full the rows get no list: the selection
// streams signature rows for their pairs, as it did before round 5.
static int sparse_fill(gs_index *ix, const uint16_t *rows, uint64_t ld, uint64_t a0, uint32_t nrows)
{
    gs_ctx *c = ix->ctx;
    if (!ix->sp_L || nrows == 0) return GS_OK;
    const uint64_t need = (uint64_t)nrows * ((((uint64_t)ix->sp_L * 6 + 31) & ~(uint64_t)31) + (((a0 + nrows + 31) / 32 * 4 + 31) & ~(uint64_t)31));
    if (ix->sp_reserved + need > ix->sp_vm.mapped && !ix->sp_exhausted) {
        SpArena h{};
        GS_HIP_CHECK(hipMemcpyAsync(&h, ix->sp_arena.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        const uint64_t used = std::min<uint64_t>(h.off, h.size);
        if (used + need > ix->sp_vm.mapped) {
            // map ahead: the worst case of this launch + 1 GB, so that the read back above (a stream synchronisation) stays rare
            const uint64_t want = std::min<uint64_t>(ix->sp_vm.va_bytes, used + need + ix->sp_vm.chunk);
            // (a failure leaves what is mapped: the kernel turns rows away when it runs out, and the host stops asking - no more read backs)
            if (!ix->sp_vm.map_to(want) || used + need > ix->sp_vm.mapped) ix->sp_exhausted = true;
            h.size = ix->sp_vm.mapped; h.base = (unsigned long long)ix->sp_vm.va; h.off = used;
            GS_HIP_CHECK(hipMemcpyAsync(ix->sp_arena.p, &h, 24, hipMemcpyHostToDevice, c->stream));      // base, off (unchanged), size
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));                                                    // (h is a local)
        }
        ix->sp_reserved = used + need;
    } else ix->sp_reserved += need;
    const uint32_t below = (uint32_t)std::min<uint64_t>(ix->sp_L, (uint64_t)ix->prm.ef_construction + ix->prm.ef_construction / 4);
    hipLaunchKernelGGL(k_sparse_fill, dim3(nrows), dim3(SPF_T), 0, c->stream, rows, ld, a0, ix->prm.m, ix->sp_L, ix->sp_off.as<uint32_t>(),
                       ix->sp_meta.as<uint64_t>(), ix->sp_bm.as<uint64_t>(), ix->sp_arena.as<SpArena>(), below);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}
static bool prepass_ok(const gs_index *ix, uint32_t efc)
{
    const uint32_t maxdeg = 2 * ix->prm.max_nb_conn, knbn = 1;
    if (getenv("GS_PLAN_PREPASS") && !atoi(getenv("GS_PLAN_PREPASS"))) return false;
    if (ix->n < 4096 || ix->entry < 0 || maxdeg > (uint32_t)DT / 2 || efc > 65535u || efc < 2 || ix->prm.m > 65535u) return false;
    uint32_t sort_cap = 2; while (sort_cap < 4 * efc) sort_cap <<= 1;
    if (dense_split_w(ix, knbn, maxdeg, (uint32_t)DCN, (size_t)8 * sort_cap + 64)) return true;     // (round 5) beyond the LDS: the split bitmap
    if (!dense_vis_in_lds(ix, knbn, maxdeg)) return false;                       // WLOG is instantiated for the LDS bitmap and its split form
    return std::max<size_t>(dense_lds_bytes(ix->prm.m, knbn, maxdeg, ix->n, true), (size_t)8 * sort_cap + 64) <= 160 * 1024 - 1024;
}
static int plan_prepass(gs_index *ix, uint32_t nb, uint32_t efc, const uint16_t *mat, uint64_t mat_ld, const uint64_t **w0k, const uint32_t **w0n, const uint64_t **w0e)
{
    gs_ctx *c = ix->ctx;
    *w0k = nullptr; *w0n = nullptr; *w0e = nullptr;
    const uint32_t maxdeg = 2 * ix->prm.max_nb_conn, knbn = 1;
    uint32_t sort_cap = 2; while (sort_cap < 4 * efc) sort_cap <<= 1;             // keys the epilogue can sort: ties at dmax ride along
    const uint32_t vis_w = dense_split_w(ix, knbn, maxdeg, (uint32_t)DCN, (size_t)8 * sort_cap + 64);
    size_t lds = dense_lds_bytes(ix->prm.m, knbn, maxdeg, vis_w ? (uint64_t)vis_w : ix->n, true);
    lds = std::max<size_t>(lds, (size_t)8 * sort_cap + 64);
    const size_t granted = round_up(lds, 1280);
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / granted));
    const uint32_t scratch_words = dense_nblocks(ix->prm.m) * (HB / 2) + (vis_w && ix->n > vis_w ? (uint32_t)((ix->n - vis_w + 31) / 32) : 0u);
    const uint32_t capC = 2 * efc + 2 * (uint32_t)DCN + maxdeg + 64,
                   cap_log = 16 * efc + ((getenv("GS_DENSE_PHASE2") && !atoi(getenv("GS_DENSE_PHASE2"))) ? 1u : 0u);      // (odd = order-free phase 2 off, A/B)
    const uint32_t grid = (uint32_t)std::min<uint64_t>(nb, (uint64_t)c->n_cu * per_cu);
    int rc;
    if ((rc = ensure_stats(ix))) return rc;
    if ((rc = ix->visited.ensure((size_t)4 * scratch_words * c->n_cu * 3))) return rc;
    if ((rc = ix->cbuf.ensure((size_t)16 * capC * c->n_cu * 3))) return rc;
    if ((rc = ix->counter.ensure(64))) return rc;
    if ((rc = ix->wlog.ensure((size_t)8 * cap_log * grid))) return rc;
    if ((rc = ix->w0_keys.ensure((size_t)8 * efc * nb))) return rc;
    if ((rc = ix->w0_n.ensure((size_t)4 * nb))) return rc;
    if ((rc = ix->w0_evals.ensure((size_t)8 * nb))) return rc;
    GS_HIP_CHECK(hipMemsetAsync(ix->counter.p, 0, 8, c->stream));
    IndexDev d = index_dev(ix);
    {   // (inside the caller's FAM_INSERT profiling scope)
#define GS_LAUNCH_WL(O) GS_LAUNCH_WL_S(O, false)
#define GS_LAUNCH_WL_S(O, SP)                                                                                             \
    do {                                                                                                                  \
        auto kern = k_hnsw_search_dense<true, false, O, false, true, SP>;                                                 \
        GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));      \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(DT), lds, c->stream, d, (uint64_t)nb, knbn, efc, mat, mat_ld, ix->visited.as<uint32_t>(), scratch_words, \
                           ix->cbuf.as<uint64_t>(), capC, ix->counter.as<unsigned long long>(), (uint64_t *)nullptr, (float *)nullptr, (uint32_t *)nullptr, \
                           ix->w0_evals.as<uint64_t>(), (unsigned long long *)nullptr, (unsigned long long *)nullptr, ix->wlog.as<uint64_t>(), cap_log, sort_cap, \
                           ix->w0_keys.as<uint64_t>(), ix->w0_n.as<uint32_t>(), ix->ep0.as<uint32_t>(), vis_w);               \
    } while (0)
        if (vis_w) GS_LAUNCH_WL_S(4, true); else if (per_cu >= 3) GS_LAUNCH_WL(6); else GS_LAUNCH_WL(4);
#undef GS_LAUNCH_WL
#undef GS_LAUNCH_WL_S
    }
    GS_HIP_CHECK(hipGetLastError());
    *w0k = ix->w0_keys.as<uint64_t>(); *w0n = ix->w0_n.as<uint32_t>(); *w0e = ix->w0_evals.as<uint64_t>();
    return GS_OK;
}
}  // namespace gs

static int insert_common(gs_index *ix, const void *sigs, bool on_dev, uint64_t n, const uint64_t *ids = nullptr)
{
    GS_REQUIRE(ix, GS_ERR_INVALID, "null index");
    if (n == 0) return GS_OK;
    GS_REQUIRE(sigs, GS_ERR_INVALID, "null signatures");
    gs_ctx *c = ix->ctx;
    GS_CTX_LOCK(c);
    const uint32_t M = ix->prm.max_nb_conn, ML = ix->prm.max_layer, maxdeg = 2 * M, efc = ix->prm.ef_construction;
    const uint32_t B = std::min<uint32_t>(std::min<uint32_t>(ix->prm.insert_batch, 256u), gs::ST);
    GS_REQUIRE(!ix->prm.keep_pruned, GS_ERR_UNSUPPORTED, "keep_pruned=true is not implemented on the device (gsearch sets false, dnasketch.rs:160)");
    GS_REQUIRE(ix->n + n < ((uint64_t)1 << 32) - 1, GS_ERR_INVALID, "too many points");
    GS_HIP_CHECK(hipSetDevice(c->device));
    // levels and upper-layer slots of the new points (host: needs libm log, like the oracle)
    std::vector<uint8_t> lv(n);
    std::vector<int32_t> up(n);
    uint64_t nup = ix->n_upper;
    for (uint64_t i = 0; i < n; i++) { lv[i] = (uint8_t)gen_level_host(ix, ix->n + i); up[i] = lv[i] > 0 ? (int32_t)nup++ : -1; }
    int rc = gs::index_reserve(ix, ix->n + n, nup);
    if (rc) return rc;
    const uint64_t first = ix->n;
    gs::ids_append(ix, ids, lv.data(), n);
    struct IdGuard {          // an insert that fails half-way leaves ix->n behind first + n: drop the ids / ranks of the points that did not make it
        gs_index *ix; const std::vector<uint8_t> &lv; uint64_t first;
        ~IdGuard()
        {
            for (uint64_t i = ix->n > first ? ix->n - first : 0; i < lv.size() && first + i < ix->pid_rank.size(); i++) ix->level_count[lv[i] <= 16 ? lv[i] : 16]--;
            if (ix->pid_rank.size() > ix->n) ix->pid_rank.resize(ix->n);
            if (ix->origin.size() > ix->n) ix->origin.resize(ix->n);
        }
    } id_guard{ix, lv, first};
    if ((rc = gs::upload_user_rows(ix, ix->data.as<uint8_t>() + first * ix->stride, sigs, n, on_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(ix->levels.as<uint8_t>() + first, lv.data(), n, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(ix->upidx.as<int32_t>() + first, up.data(), 4 * n, hipMemcpyHostToDevice, c->stream));
    // Overlap (round 3, GS_INSERT_OVERLAP=1; OFF by default): the match-join of batch i+1 only needs signatures - not batch i's links - so it
    // can run on a second stream under the pre-pass / plan / link kernels of batch i. Measured on the 300 k-genome build: 18.0 s with and
    // without - the join's two 72 kB workgroups per CU and the plan kernel's 85 kB do not share a CU, the streams just take turns.
    // Its rows were queued on the main stream just above.
    const bool overlap = getenv("GS_INSERT_OVERLAP") && atoi(getenv("GS_INSERT_OVERLAP"));
    if (overlap) {
        if (!ix->jstream) {
            GS_HIP_CHECK(hipStreamCreateWithFlags(&ix->jstream, hipStreamNonBlocking));
            GS_HIP_CHECK(hipEventCreateWithFlags(&ix->jev, hipEventDisableTiming));
            GS_HIP_CHECK(hipEventCreateWithFlags(&ix->jev_up, hipEventDisableTiming));
        }
        GS_HIP_CHECK(hipEventRecord(ix->jev_up, c->stream));
        GS_HIP_CHECK(hipStreamWaitEvent(ix->jstream, ix->jev_up, 0));
    }
    struct JGuard {          // whatever way this function is left: nothing of it is still running on the second stream, and the context has its own stream back
        gs_index *ix; gs_ctx *c; hipStream_t main;
        ~JGuard() { c->stream = main; if (ix->jstream) (void)hipStreamSynchronize(ix->jstream); }
    } jguard{ix, c, c->stream};
    bool pf_have = false; uint64_t pf_b0 = 0;
    // scratch
    const uint32_t ef_lds = std::max(efc, B);
    const size_t lds = gs::search_lds_bytes(ef_lds, maxdeg);
    GS_REQUIRE(lds <= 160 * 1024 - 64, GS_ERR_UNSUPPORTED, "ef_construction=%u needs %zu bytes of LDS", efc, lds);
    GS_REQUIRE(2 * (size_t)ef_lds + maxdeg + 64 <= (size_t)gs::SMAXI * gs::ST && maxdeg <= (uint32_t)gs::ST, GS_ERR_UNSUPPORTED, "ef_construction too large");
    const uint32_t vis_words = (uint32_t)((first + n + 31) / 32);
    if ((rc = ix->visited.ensure((size_t)4 * vis_words * std::max<uint32_t>(B, c->n_cu)))) return rc;
    if ((rc = ix->blevels.ensure(B))) return rc;
    if ((rc = ix->cntmat.ensure((size_t)4 * B * B))) return rc;
    if ((rc = ix->plan_keys.ensure((size_t)8 * B * ML * maxdeg))) return rc;
    if ((rc = ix->plan_n.ensure((size_t)4 * B * ML))) return rc;
    if ((rc = ix->touched.ensure((size_t)4 * B * (maxdeg + (size_t)ML * M)))) return rc;
    if ((rc = ix->ntouched.ensure(64))) return rc;
    // extend_candidates with ef_construction <= 2M: every layer-0 selection extends W by the neighbours of its members (select_extended);
    // one key array per point of a batch, sized for every node the extension can reach (a power of two: it is bitonic-sorted in place)
    uint32_t ext_cap = 0;
    if (ix->prm.extend_candidates && efc <= maxdeg) {
        const uint64_t reach = std::min<uint64_t>(ix->n + n, (uint64_t)efc * maxdeg) + efc;
        ext_cap = 2; while (ext_cap < reach) ext_cap <<= 1;
        if ((rc = ix->ext_keys.ensure((size_t)8 * ext_cap * B))) return rc;
    }
    if (!ix->evals_dev.p) { if ((rc = ix->evals_dev.alloc(64))) return rc; GS_HIP_CHECK(hipMemsetAsync(ix->evals_dev.p, 0, 64, c->stream)); }
    const uint64_t nlists = ix->cap + ix->cap_upper * ML;
    if (nlists != ix->inbox_lists) {
        if ((rc = ix->inbox.alloc((size_t)8 * nlists * B))) return rc;
        if ((rc = ix->inbox_cnt.alloc((size_t)4 * nlists))) return rc;
        GS_HIP_CHECK(hipMemsetAsync(ix->inbox_cnt.p, 0, (size_t)4 * nlists, c->stream));
        ix->inbox_lists = nlists;
    }
    gs::GraphDev g;
    g.deg0 = ix->deg0.as<uint32_t>(); g.nbr0 = ix->nbr0.as<uint32_t>(); g.cnt0 = ix->cnt0.as<uint32_t>();
    g.degU = ix->degU.as<uint32_t>(); g.nbrU = ix->nbrU.as<uint32_t>(); g.cntU = ix->cntU.as<uint32_t>();
    g.upidx = ix->upidx.as<int32_t>(); g.M = M; g.max_layer = ML; g.upper_base = ix->cap;
    ix->n_upper = nup;
    const gs::DistMode mode = gs::env_mode();
    unsigned long long seg_ev0 = ix->insert_evals; double seg_den = 0; uint32_t seg_batches = 0;
    const bool cnt16 = ix->prm.m <= 65535;
    if (ix->pair_cache_budget == 0) {
        const char *e = getenv("GS_PAIR_CACHE_GB");
        if (e) ix->pair_cache_budget = (uint64_t)(atof(e) * 1e9);
        else {
            // default: 55 % of the device, but never more than what is FREE now minus what this index still has to allocate next to it
            // (column store, count matrix, query scratch): other indexes and processes may hold memory already
            size_t fr = 0, tot = 0;
            uint64_t budget = (uint64_t)(0.55 * (double)c->hbm_bytes);
            if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
                const uint64_t reserve = (uint64_t)ix->prm.m * ix->cap * ix->esz + ((uint64_t)8 << 30);
                budget = std::min<uint64_t>(budget, fr > reserve ? (uint64_t)fr - reserve : 0);
            }
            ix->pair_cache_budget = std::max<uint64_t>(budget, 1);
        }
    }
    const uint64_t slab_ld = gs::round_up(first + n, 8);
    // batches joined together (GS_INSERT_GROUP, default 8; 1 = every batch its own join): rows [grp_b0, grp_end) hold their counts against nodes [0, grp_b0)
    uint32_t grp_n = overlap ? 1u : 8u;
    if (const char *e = getenv("GS_INSERT_GROUP")) grp_n = overlap ? 1u : (uint32_t)std::max(1, std::min(12, atoi(e)));
    uint64_t grp_b0 = 0, grp_end = 0;
    gs::DevBuf *slab = nullptr; uint64_t slab_first = 0;      // rows of this call's points, allocated at the first dense batch
    bool slab_tried = false;
    for (uint64_t b0 = first; b0 < first + n; b0 += B) {
        const uint32_t nb = (uint32_t)std::min<uint64_t>(B, first + n - b0);
        const uint8_t *blv = lv.data() + (b0 - first);
        GS_HIP_CHECK(hipMemcpyAsync(ix->blevels.p, blv, nb, hipMemcpyHostToDevice, c->stream));
        GS_HIP_CHECK(hipMemsetAsync(ix->plan_n.p, 0, (size_t)4 * nb * ML, c->stream));
        GS_HIP_CHECK(hipMemsetAsync(ix->ntouched.p, 0, 4, c->stream));
        const uint8_t *rows = ix->data.as<uint8_t>() + b0 * ix->stride;
        if (nb > 1) { if ((rc = gs::hamming_qxc_strided(c, ix->ikind, ix->prm.m, rows, nb, ix->stride, rows, nb, ix->stride, nullptr, ix->cntmat.as<uint32_t>(), nullptr, nb))) return rc; }
        gs::IndexDev d = gs::index_dev(ix);
        d.n = b0; d.entry = ix->entry; d.top = ix->top;                  // the graph frozen at batch start
        const uint32_t vw = (uint32_t)((b0 + 31) / 32);
        // feedback for the cost model: every 8 batches measure which fraction of the graph an insertion evaluates
        if (mode == gs::MODE_AUTO && b0 >= 4096 && (ix->insert_frac < 0 ? seg_batches >= 2 : seg_batches >= 32)) {
            unsigned long long ev = 0;
            GS_HIP_CHECK(hipMemcpyAsync(&ev, ix->evals_dev.p, 8, hipMemcpyDeviceToHost, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (seg_den > 0) ix->insert_frac = (double)(ev - seg_ev0) / seg_den;
            seg_ev0 = ev; seg_den = 0; seg_batches = 0;
        }
        const bool dense = (mode == gs::MODE_DENSE && b0 > 0) || (mode == gs::MODE_AUTO && b0 >= 4096 && nb >= 64 && ix->insert_frac >= 0 && gs::dense_pays(ix, ix->insert_frac, nb));
        const uint16_t *matp = nullptr; uint64_t mat_ld = 0;
        if (dense && cnt16) {
            // the column copy is (re)allocated here, BEFORE a slab is taken or used: ensure_cols may have to evict the pair cache to find
            // room, and a slab pointer held across that would dangle
            if (gs::use_join(ix) && (rc = gs::ensure_cols(ix, b0))) return rc;
            if (slab && ix->slabs.empty()) slab = nullptr;           // the cache was given back (drop_pair_cache): rows go to ix->mat from here on
            if (!slab_tried) {
                slab_tried = true;
                const uint64_t need = (first + n - b0) * slab_ld * 2;
                // a slab is only taken while the device keeps room for the next growth step of the index itself: a new signature block
                // next to the old one (x1.5) and the column copy that follows it, ~2x the signatures held now, + 16 GB of working space
                bool room = true;
                {
                    size_t fr = 0, tot = 0;
                    if (hipMemGetInfo(&fr, &tot) == hipSuccess) room = (uint64_t)fr >= need + 2 * (uint64_t)ix->stride * ix->cap + ((uint64_t)16 << 30);
                }
                if (room && ix->pair_cache_bytes + need <= ix->pair_cache_budget) {
                    slab = new gs::DevBuf();
                    if (slab->alloc(need) != GS_OK) { delete slab; slab = nullptr; }
                    else { ix->slabs.push_back(slab); ix->pair_cache_bytes += need; slab_first = b0; }
                }
                // the nodes inserted before the first cached batch (the first 4096 in auto mode) have no rows of their own: a pair of two
                // of them would send the selection heuristic back to streaming 2M signature rows per candidate - 5 % of the candidates,
                // most of its time. One all-pairs tile pass (b0^2 counts, 32 MB at 4096) closes the hole.
                if (slab && !ix->early_cached && b0 > 0 && b0 <= 32768) {
                    const uint64_t eld = gs::round_up(b0, 8), ebytes = b0 * eld * 2;
                    gs::DevBuf *early = new gs::DevBuf();
                    if (ix->pair_cache_bytes + ebytes <= ix->pair_cache_budget && early->alloc(ebytes) == GS_OK) {
                        if ((rc = gs::hamming_qxc_strided(c, ix->ikind, ix->prm.m, ix->data.p, b0, ix->stride, ix->data.p, b0, ix->stride, nullptr, nullptr, early->as<uint16_t>(), eld))) { delete early; return rc; }
                        hipLaunchKernelGGL(gs::k_set_rowptr, dim3((uint32_t)((b0 + 255) / 256)), dim3(256), 0, c->stream, early->as<uint16_t>(), eld, b0, ix->rowptr.as<uint64_t>());
                        GS_HIP_CHECK(hipGetLastError());
                        ix->slabs.push_back(early); ix->pair_cache_bytes += ebytes; ix->early_cached = true;
                    } else delete early;
                }
            }
            // sparse pair rows (round 5): allocated here, at the first dense batch of the index (they cost sp_L x 6 bytes per node of capacity)
            if (!ix->sp_tried) {
                ix->sp_tried = true;
                const char *eo = getenv("GS_SPARSE_ROWS"), *el = getenv("GS_SPARSE_L"), *ea = getenv("GS_SPARSE_ARENA_GB"), *eb = getenv("GS_SPARSE_BITMAP_GB");
                // what the lists and bitmaps may take: what the device keeps free next to the signatures and their column copy AT THE DECLARED CAPACITY
                // (hnsw_params.capacity: 1.5 M in gsearch) and 40 GB of everything else (adjacency, count matrices, join scratch, the caller's own buffers).
                // List length: up to 8192 entries - a level of chance matches must fit whole -, less when even half-full lists would not fit
                const uint64_t capd = std::max<uint64_t>(ix->prm.capacity, ix->cap);
                const uint64_t fixed = 2 * (uint64_t)ix->stride * capd + ((uint64_t)40 << 30);
                uint64_t room = c->hbm_bytes > fixed ? c->hbm_bytes - fixed : 0;
                if (ea) room = (uint64_t)(atof(ea) * 1e9);
                room = std::min<uint64_t>(room, (uint64_t)120 << 30);                       // (32-bit offsets in 32-byte units reach 128 GB)
                const uint32_t autoL = (uint32_t)std::max<uint64_t>(2048, std::min<uint64_t>(8192, room / (3 * capd) / 512 * 512));
                ix->sp_L = (eo && !atoi(eo)) ? 0u : (uint32_t)std::max(64, std::min(32768, el ? atoi(el) : (int)autoL));
                if (ix->sp_L && !ea && room < ((uint64_t)2 << 30)) ix->sp_L = 0;
                // (GS_SPARSE_ARENA_CHUNK_MB: the mapping step, 1 GB; tests make it small to fill an arena of a few MB and watch it grow in place)
                const size_t vm_chunk = getenv("GS_SPARSE_ARENA_CHUNK_MB") ? (size_t)(atof(getenv("GS_SPARSE_ARENA_CHUNK_MB")) * 1048576.0) : ((size_t)1 << 30);
                if (ix->sp_L) {
                    if (!ix->sp_vm.reserve(c->device, std::max<uint64_t>(room, 65536), vm_chunk) || ix->sp_off.alloc((size_t)4 * ix->cap) != GS_OK || ix->sp_meta.alloc((size_t)8 * ix->cap) != GS_OK ||
                        ix->sp_bm.alloc((size_t)8 * ix->cap) != GS_OK || ix->sp_arena.alloc(sizeof(gs::SpArena)) != GS_OK) {
                        (void)hipGetLastError(); ix->sp_vm.release(); ix->sp_off.release(); ix->sp_meta.release(); ix->sp_bm.release(); ix->sp_arena.release(); ix->sp_L = 0;
                    } else {
                        GS_HIP_CHECK(hipMemsetAsync(ix->sp_off.p, 0, (size_t)4 * ix->cap, c->stream));
                        GS_HIP_CHECK(hipMemsetAsync(ix->sp_meta.p, 0, (size_t)8 * ix->cap, c->stream));
                        GS_HIP_CHECK(hipMemsetAsync(ix->sp_bm.p, 0, (size_t)8 * ix->cap, c->stream));
                        gs::SpArena h{};
                        h.base = (unsigned long long)ix->sp_vm.va;
                        h.bm_limit = (unsigned long long)((eb ? atof(eb) : 24.0) * 1e9);
                        if ((double)h.bm_limit > 0.6 * (double)room) h.bm_limit = (unsigned long long)(0.6 * (double)room);
                        GS_HIP_CHECK(hipMemcpyAsync(ix->sp_arena.p, &h, sizeof(h), hipMemcpyHostToDevice, c->stream));
                        GS_HIP_CHECK(hipStreamSynchronize(c->stream));                  // (h is a local)
                        ix->sp_reserved = 0;
                    }
                }
                // the nodes inserted before this batch have no count rows to take their lists from: one all-pairs tile pass over them (like the dense
                // cache's early rows, which it reuses when they exist), lists written, matrix given back
                if (ix->sp_L && b0 > 0 && b0 <= 32768) {
                    const uint64_t eld = gs::round_up(b0, 8);
                    gs::DevBuf tmp; const uint16_t *em = nullptr;
                    if (ix->early_cached && !ix->slabs.empty()) em = ix->slabs.back()->as<uint16_t>();      // (pushed just above, this call)
                    else if (tmp.alloc(b0 * eld * 2) == GS_OK) {
                        if ((rc = gs::hamming_qxc_strided(c, ix->ikind, ix->prm.m, ix->data.p, b0, ix->stride, ix->data.p, b0, ix->stride, nullptr, nullptr, tmp.as<uint16_t>(), eld))) return rc;
                        em = tmp.as<uint16_t>();
                    } else (void)hipGetLastError();
                    if (em) {
                        if ((rc = gs::sparse_fill(ix, em, eld, 0, (uint32_t)b0))) return rc;
                        GS_HIP_CHECK(hipStreamSynchronize(c->stream));          // tmp goes out of scope
                    }
                }
                d = gs::index_dev(ix); d.n = b0; d.entry = ix->entry; d.top = ix->top;
            }
            // where this batch's count rows live: the call's slab of the dense pair cache (kept), or a rolling buffer of one GROUP of batches (ix->mat;
            // round 5: without a slab the batches were joined one by one - the columns streamed once per batch instead of once per group)
            const bool can_group = grp_n > 1 && gs::use_join(ix);
            uint16_t *out16; mat_ld = slab_ld;
            if (slab) out16 = slab->as<uint16_t>() + (b0 - slab_first) * slab_ld;
            else {
                const size_t rows_wanted = can_group ? (size_t)grp_n * B : (size_t)B;
                if (!(b0 >= grp_b0 && b0 < grp_end)) {                                    // a new group (or single batch) starts: the buffer may grow now
                    // (with 25 % of headroom, up to the declared capacity: the row length follows the index, and a buffer that is a few rows short at every
                    // insert call costs a multi-GB hipFree + hipMalloc per call - 183 of them in a 1.5 M-genome build fed 8192 genomes at a time)
                    const size_t want = 2 * rows_wanted * slab_ld;
                    if (ix->mat.bytes < want) {
                        const size_t at_cap = 2 * rows_wanted * gs::round_up(std::max<uint64_t>(ix->prm.capacity, ix->cap), 8);
                        const size_t ask = std::max(want, std::min(want + want / 4, at_cap));
                        rc = alloc_or_evict(ix, ix->mat, ask);
                        if (rc && ask > want) { (void)hipGetLastError(); rc = alloc_or_evict(ix, ix->mat, want); }
                        if (rc) return rc;
                    }
                }
                out16 = (b0 >= grp_b0 && b0 < grp_end) ? ix->mat.as<uint16_t>() + (b0 - grp_b0) * slab_ld : ix->mat.as<uint16_t>();
            }
            if (can_group && b0 >= grp_b0 && b0 < grp_end) {
                // inside a group: this batch's counts against the nodes of the group's start are in its rows already; the nodes the earlier
                // batches of the group added since (at most (grp_n - 1) * B of them) take a small join of their own (their columns only)
                if (b0 > grp_b0 && (rc = gs::dense_counts_range(ix, rows, nb, grp_b0, b0 - grp_b0, out16, mat_ld))) return rc;
            } else if (can_group) {
                // a group of batches: ONE join of all their points against the nodes present now - the columns (21.6 GB at 300 k nodes) are
                // streamed once per group instead of once per batch
                grp_b0 = b0; grp_end = std::min<uint64_t>(first + n, b0 + (uint64_t)grp_n * B);
                if ((rc = gs::dense_counts(ix, rows, grp_end - grp_b0, b0, out16, mat_ld))) return rc;
            } else if (slab && pf_have && pf_b0 == b0) {                    // its counts were produced on the second stream while the previous batch was planned
                GS_HIP_CHECK(hipStreamWaitEvent(c->stream, ix->jev, 0));
                pf_have = false;
            } else {
                if (pf_have) { GS_HIP_CHECK(hipStreamSynchronize(ix->jstream)); pf_have = false; }
                grp_b0 = grp_end = 0;
                if ((rc = gs::dense_counts(ix, rows, nb, b0, out16, mat_ld))) return rc;
                // a join produced on the main stream shares the query-column scratch and the column store with the next one on the second
                // stream: that one must not start before this one is done
                if (overlap) { GS_HIP_CHECK(hipEventRecord(ix->jev_up, c->stream)); GS_HIP_CHECK(hipStreamWaitEvent(ix->jstream, ix->jev_up, 0)); }
            }
            if (slab && ix->slabs.empty()) {                         // evicted under our feet after all: this batch's counts again, into ix->mat
                slab = nullptr; grp_b0 = grp_end = 0;
                if (ix->mat.bytes < (size_t)2 * B * slab_ld && (rc = ix->mat.alloc((size_t)2 * B * slab_ld))) return rc;
                out16 = ix->mat.as<uint16_t>();
                if ((rc = gs::dense_counts(ix, rows, nb, b0, out16, mat_ld))) return rc;
            }
            // the mates' columns from the tile matrix; with a slab the rows stay where they are as the dense pair cache (rowptr)
            hipLaunchKernelGGL(gs::k_cache_rows, dim3(nb), dim3(256), 0, c->stream, out16, mat_ld, b0, nb, ix->cntmat.as<uint32_t>(), slab ? ix->rowptr.as<uint64_t>() : (uint64_t *)nullptr);
            GS_HIP_CHECK(hipGetLastError());
            if ((rc = gs::sparse_fill(ix, out16, mat_ld, b0, nb))) return rc;
            matp = out16;
        }
        if (b0 >= 4096) { seg_den += (double)nb * (double)b0; seg_batches++; }
        // pre-pass: the layer-0 search of the batch's level-0 points through the dense traversal kernel (accepted-key log -> W)
        const uint64_t *w0k = nullptr, *w0e = nullptr; const uint32_t *w0n = nullptr;
        const bool pp = matp && gs::prepass_ok(ix, efc);
        if (pp) {
            if ((rc = ix->ep0.ensure((size_t)4 * B))) return rc;
            GS_HIP_CHECK(hipMemsetAsync(ix->ep0.p, 0xFF, (size_t)4 * nb, c->stream));
        }
        const size_t lds_vis = ((lds + 15) & ~(size_t)15) + (size_t)4 * vw;
        const int vis_in_lds = lds_vis <= 160 * 1024 - 1024 && !getenv("GS_PLAN_VIS_GLOBAL");
        const size_t lds_plan = vis_in_lds ? lds_vis : lds;
        {
            gs::ProfScope ps(c, gs::FAM_INSERT);
#define GS_LAUNCH_PLAN(K)                                                                                                  \
    do {                                                                                                                   \
        auto kern = gs::k_hnsw_plan<K>;                                                                                    \
        GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_plan));  \
        hipLaunchKernelGGL(kern, dim3(nb), dim3(gs::ST), lds_plan, c->stream, d, b0, nb, ix->blevels.as<uint8_t>(), ix->cntmat.as<uint32_t>(), matp, mat_ld, efc, ef_lds, \
                           ix->prm.extend_candidates, ix->visited.as<uint32_t>(), vw, vis_in_lds, ix->plan_keys.as<uint64_t>(), ix->plan_n.as<uint32_t>(), \
                           ix->evals_dev.as<unsigned long long>(), w0k, w0n, w0e, phase, ix->ep0.as<uint32_t>(),           \
                           ext_cap ? ix->ext_keys.as<uint64_t>() : (uint64_t *)nullptr, ext_cap);                          \
    } while (0)
#define GS_LAUNCH_PLAN_KIND()                                                                                              \
    do { if (ix->ikind == GS_KIND_F32) GS_LAUNCH_PLAN(GS_KIND_F32); else if (ix->ikind == GS_KIND_U32) GS_LAUNCH_PLAN(GS_KIND_U32); else GS_LAUNCH_PLAN(GS_KIND_U64); } while (0)
            int phase = 0;
            if (!pp) GS_LAUNCH_PLAN_KIND();
            else {
                phase = 1; GS_LAUNCH_PLAN_KIND();                  // layers above 0 of the few points that have any
                GS_HIP_CHECK(hipGetLastError());
                if ((rc = gs::plan_prepass(ix, nb, efc, matp, mat_ld, &w0k, &w0n, &w0e))) return rc;
                phase = 2; GS_LAUNCH_PLAN_KIND();                  // selection on the W the pre-pass worked out
            }
#undef GS_LAUNCH_PLAN_KIND
#undef GS_LAUNCH_PLAN
        }
        GS_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL(gs::k_link_own, dim3(nb), dim3(256), 0, c->stream, g, b0, nb, ix->blevels.as<uint8_t>(), ix->plan_keys.as<uint64_t>(), ix->plan_n.as<uint32_t>());
        hipLaunchKernelGGL(gs::k_link_scatter, dim3(nb), dim3(256), 0, c->stream, g, b0, nb, B, ix->blevels.as<uint8_t>(), ix->plan_keys.as<uint64_t>(),
                           ix->plan_n.as<uint32_t>(), ix->inbox_cnt.as<uint32_t>(), ix->inbox.as<uint64_t>(), ix->touched.as<uint32_t>(), ix->ntouched.as<uint32_t>());
        hipLaunchKernelGGL(gs::k_link_merge, dim3(std::min<uint32_t>(nb * 64u, (uint32_t)c->n_cu * 8u)), dim3(gs::LM_T), 0, c->stream, g, B, ix->inbox_cnt.as<uint32_t>(),
                           ix->inbox.as<uint64_t>(), ix->touched.as<uint32_t>(), ix->ntouched.as<uint32_t>());
        GS_HIP_CHECK(hipGetLastError());
        // the next batch's match-join, on the second stream, while this batch's kernels run (same decision as the loop head will take)
        if (overlap && slab && matp && gs::use_join(ix) && b0 + nb < first + n) {
            const uint64_t b0n = b0 + nb;
            const uint32_t nbn = (uint32_t)std::min<uint64_t>(B, first + n - b0n);
            const bool dn = (mode == gs::MODE_DENSE) || (mode == gs::MODE_AUTO && b0n >= 4096 && nbn >= 64 && ix->insert_frac >= 0 && gs::dense_pays(ix, ix->insert_frac, nbn));
            if (dn && !ix->slabs.empty()) {
                c->stream = ix->jstream;
                rc = gs::dense_counts(ix, ix->data.as<uint8_t>() + b0n * ix->stride, nbn, b0n, slab->as<uint16_t>() + (b0n - slab_first) * slab_ld, slab_ld);
                c->stream = jguard.main;
                if (rc) return rc;
                GS_HIP_CHECK(hipEventRecord(ix->jev, ix->jstream));
                pf_have = true; pf_b0 = b0n;
            }
        }
        // entry point: the first id of the highest new level (SPEC 5)
        for (uint32_t i = 0; i < nb; i++) if ((int)blv[i] > ix->top) { ix->top = blv[i]; ix->entry = (int64_t)(b0 + i); }
        ix->n = b0 + nb;
        // the level / batch buffers are reused by the next batch: the stream orders the copies after the kernels
    }
    unsigned long long ev[3] = {0, 0, 0};
    GS_HIP_CHECK(hipMemcpyAsync(ev, ix->evals_dev.p, 24, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    ix->insert_evals = ev[0];
    if (getenv("GS_SPARSE_VERBOSE")) {
        uint64_t with = 0, full = 0;
        if (ix->sp_L && ix->sp_meta.p) {
            std::vector<uint64_t> mt(ix->n);
            GS_HIP_CHECK(hipMemcpy(mt.data(), ix->sp_meta.p, 8 * ix->n, hipMemcpyDeviceToHost));
            for (uint64_t v : mt) { with += v >> 63; full += (v >> 63) && ((v >> 16) & 0xFFFFu) == ix->sp_L; }
        }
        gs::SpArena ar{};
        if (ix->sp_arena.p) GS_HIP_CHECK(hipMemcpy(&ar, ix->sp_arena.p, sizeof(ar), hipMemcpyDeviceToHost));
        fprintf(stderr, "[GS_SPARSE] n %llu L %u: nodes with a list %llu (at capacity %llu, turned away %llu), dense-cache bytes %llu; selection chunks through the lists %llu, candidates checked by streaming rows %llu (both since the index was made); arena %.4f GB used of %.4f mapped (%.2f reserved), level bitmaps %llu in %.2f GB, turned away %llu\n",
                (unsigned long long)ix->n, ix->sp_L, (unsigned long long)with, (unsigned long long)full, ar.lists_noroom, (unsigned long long)ix->pair_cache_bytes, ev[2], ev[1],
                std::min(ar.off, ar.size) / 1e9, ix->sp_vm.mapped / 1e9, ix->sp_vm.va_bytes / 1e9, ar.stored, std::min(ar.bm_bytes, ar.bm_limit) / 1e9, ar.noroom);
    }
    return GS_OK;
}

int gs_index_parallel_insert(gs_index *ix, const void *sigs, uint64_t n) { return insert_common(ix, sigs, false, n); }
int gs_index_parallel_insert_dev(gs_index *ix, const void *sigs_dev, uint64_t n) { return insert_common(ix, sigs_dev, true, n); }
int gs_index_parallel_insert_ids(gs_index *ix, const void *sigs, const uint64_t *ids, uint64_t n)
{
    GS_REQUIRE(ids || n == 0, GS_ERR_INVALID, "null ids");
    return insert_common(ix, sigs, false, n, ids);
}
int gs_index_parallel_insert_ids_dev(gs_index *ix, const void *sigs_dev, const uint64_t *ids, uint64_t n)
{
    GS_REQUIRE(ids || n == 0, GS_ERR_INVALID, "null ids");
    return insert_common(ix, sigs_dev, true, n, ids);
}

int gs_index_bruteforce_search(gs_index *ix, const void *queries, uint64_t nq, uint32_t knbn, uint64_t *ids, float *dist)
{
    GS_REQUIRE(ix && knbn >= 1, GS_ERR_INVALID, "bad argument");
    if (nq == 0) return GS_OK;
    GS_REQUIRE(queries && ids && dist && ix->n > 0, GS_ERR_INVALID, "bad argument");
    gs_ctx *c = ix->ctx;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    // dense copy of the data rows (the tile kernel takes unpadded rows), query blocks of 256
    const uint64_t n = ix->n;
    gs::DevBuf dense, dq, dd;
    int rc;
    if ((rc = dense.alloc(ix->rowbytes * n))) return rc;
    GS_HIP_CHECK(hipMemcpy2DAsync(dense.p, ix->rowbytes, ix->data.p, ix->stride, ix->rowbytes, n, hipMemcpyDeviceToDevice, c->stream));
    const uint64_t QB = 256;
    if ((rc = dq.alloc(ix->rowbytes * QB))) return rc;
    if ((rc = dd.alloc(4 * QB * n))) return rc;
    std::vector<float> h(QB * n);
    std::vector<uint64_t> keys(n);
    for (uint64_t q0 = 0; q0 < nq; q0 += QB) {
        const uint64_t nb = std::min(QB, nq - q0);
        if (ix->prm.kind == GS_KIND_U16) {                       // u16 rows from the host: stage, then zero-extend into the dense u32 block
            gs::PoolBuf stage(c, 39);
            if ((rc = stage.alloc(ix->user_rowbytes * nb))) return rc;
            GS_HIP_CHECK(hipMemcpyAsync(stage.p, (const uint8_t *)queries + ix->user_rowbytes * q0, ix->user_rowbytes * nb, hipMemcpyHostToDevice, c->stream));
            if ((rc = gs::widen_u16_rows(c, stage.p, nb, ix->prm.m, dq.p, ix->rowbytes))) return rc;
        } else
        GS_HIP_CHECK(hipMemcpyAsync(dq.p, (const uint8_t *)queries + ix->rowbytes * q0, ix->rowbytes * nb, hipMemcpyHostToDevice, c->stream));
        if ((rc = gs_hamming_qxc_dev(c, ix->ikind, ix->prm.m, dq.p, nb, dense.p, n, dd.as<float>()))) return rc;
        GS_HIP_CHECK(hipMemcpyAsync(h.data(), dd.p, 4 * nb * n, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        for (uint64_t i = 0; i < nb; i++) {
            const float *row = h.data() + i * n;
            // distances are exact multiples count/m: recover the integer count for the (count,id) order
            for (uint64_t j = 0; j < n; j++) keys[j] = ((uint64_t)(uint32_t)llrintf(row[j] * (float)ix->prm.m) << 32) | j;
            const uint64_t kk = std::min<uint64_t>(knbn, n);
            std::partial_sort(keys.begin(), keys.begin() + kk, keys.end());
            for (uint32_t t = 0; t < knbn; t++) {
                if (t < kk) { const uint32_t e = (uint32_t)keys[t]; ids[(q0 + i) * knbn + t] = ix->origin.empty() ? e : ix->origin[e]; dist[(q0 + i) * knbn + t] = row[e]; }
                else { ids[(q0 + i) * knbn + t] = ~(uint64_t)0; dist[(q0 + i) * knbn + t] = INFINITY; }
            }
        }
    }
    return GS_OK;
}

/* own binary dump (role of Hnsw::file_dump / HnswIo::load_hnsw, dumpload.rs:31, reloadhnsw.rs:41-51):
 *   "GSAMDIX1" | gs_index_params | n, n_upper, entry, top (u64,u64,i64,i64) | signatures (dense rows) | levels | deg0 | nbr0 | cnt0 |
 *   upidx | degU | nbrU | cntU   (the export layout of gs_index_export) */
int gs_index_save(gs_index *ix, const char *path)
{
    GS_REQUIRE(ix && path, GS_ERR_INVALID, "null argument");
    GS_REQUIRE(ix->n > 0, GS_ERR_STATE, "nothing to save");
    GS_CTX_LOCK(ix->ctx);
    FILE *f = fopen(path, "wb");
    GS_REQUIRE(f, GS_ERR_IO, "cannot open %s for writing", path);
    const uint32_t M = ix->prm.max_nb_conn, ML = ix->prm.max_layer;
    const uint64_t n = ix->n, U = ix->n_upper;
    int rc = GS_OK;
    // 'GSAMDIX1': ids are 0..n-1; 'GSAMDIX2': the caller's ids (n x u64) follow the graph
    const char magic[8] = {'G', 'S', 'A', 'M', 'D', 'I', 'X', ix->origin.empty() ? '1' : '2'};
    uint64_t hdr[4] = {n, U, (uint64_t)ix->entry, (uint64_t)(int64_t)ix->top};
    bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(&ix->prm, sizeof(ix->prm), 1, f) == 1 && fwrite(hdr, 8, 4, f) == 4;
    const uint64_t CH = 4096;
    std::vector<uint8_t> buf(ix->user_rowbytes * CH);
    for (uint64_t r0 = 0; ok && r0 < n; r0 += CH) {
        const uint64_t nr = std::min(CH, n - r0);
        if ((rc = gs_index_get_data(ix, r0, nr, buf.data()))) break;
        ok = fwrite(buf.data(), ix->user_rowbytes, nr, f) == nr;
    }
    if (ok && rc == GS_OK) {
        std::vector<uint8_t> lv(n); std::vector<uint32_t> d0(n), n0(n * 2 * M), c0(n * 2 * M); std::vector<int32_t> up(n);
        std::vector<uint32_t> dU(std::max<uint64_t>(U, 1) * ML), nU(std::max<uint64_t>(U, 1) * ML * M), cU(std::max<uint64_t>(U, 1) * ML * M);
        rc = gs_index_export(ix, lv.data(), nullptr, d0.data(), n0.data(), c0.data(), up.data(), nullptr, dU.data(), nU.data(), cU.data());
        if (rc == GS_OK)
            ok = fwrite(lv.data(), 1, n, f) == n && fwrite(d0.data(), 4, n, f) == n && fwrite(n0.data(), 4, n * 2 * M, f) == n * 2 * M &&
                 fwrite(c0.data(), 4, n * 2 * M, f) == n * 2 * M && fwrite(up.data(), 4, n, f) == n && fwrite(dU.data(), 4, U * ML, f) == U * ML &&
                 fwrite(nU.data(), 4, U * ML * M, f) == U * ML * M && fwrite(cU.data(), 4, U * ML * M, f) == U * ML * M;
        if (ok && rc == GS_OK && !ix->origin.empty()) ok = fwrite(ix->origin.data(), 8, n, f) == n;
    }
    fclose(f);
    if (rc) return rc;
    GS_REQUIRE(ok, GS_ERR_IO, "short write to %s", path);
    return GS_OK;
}
int gs_index_load(gs_ctx *c, const char *path, gs_index **out)
{
    GS_REQUIRE(c && path && out, GS_ERR_INVALID, "null argument");
    FILE *f = fopen(path, "rb");
    GS_REQUIRE(f, GS_ERR_IO, "cannot open %s", path);
    char magic[8]; gs_index_params prm; uint64_t hdr[4];
    memset(&prm, 0, sizeof prm);
    bool ok = fread(magic, 1, 8, f) == 8 && !memcmp(magic, "GSAMDIX", 7) && (magic[7] == '1' || magic[7] == '2') && fread(&prm, sizeof(prm), 1, f) == 1 && fread(hdr, 8, 4, f) == 4;
    const bool with_ids = ok && magic[7] == '2';
    if (!ok) { fclose(f); GS_REQUIRE(false, GS_ERR_IO, "%s is not a gsearch_amd index dump", path); }
    const uint64_t n = hdr[0], U = hdr[1];
    const uint32_t M = prm.max_nb_conn, ML = prm.max_layer;
    // the header is untrusted: validate it like gs_index_create would, and against the file size, BEFORE sizing any allocation from it
    struct stat st;
    const bool sane = (prm.kind == GS_KIND_F32 || prm.kind == GS_KIND_U32 || prm.kind == GS_KIND_U64 || prm.kind == GS_KIND_U16) && prm.m >= 1 && M >= 2 && M <= 255 &&
                      ML >= 1 && ML <= 16 && prm.ef_construction >= 1 && n >= 1 && n < ((uint64_t)1 << 31) && U <= n && fstat(fileno(f), &st) == 0;
    if (!sane) { fclose(f); GS_REQUIRE(false, GS_ERR_IO, "%s: corrupt header", path); }
    const size_t rowbytes = gs::kind_bytes(prm.kind) * (size_t)prm.m;
    const unsigned __int128 expect = (unsigned __int128)8 + sizeof(prm) + 32 + (unsigned __int128)n * (rowbytes + 1 + 4 + (size_t)16 * M + 4 + (with_ids ? 8 : 0)) + (unsigned __int128)U * ML * (4 + (size_t)8 * M);
    if (expect != (unsigned __int128)(uint64_t)st.st_size) { fclose(f); GS_REQUIRE(false, GS_ERR_IO, "%s: size %llu does not match its header (truncated or corrupt)", path, (unsigned long long)st.st_size); }
    std::vector<uint8_t> sigs, lv; std::vector<uint32_t> d0, n0, c0, dU, nU, cU; std::vector<int32_t> up; std::vector<uint64_t> oid;
    try {
        if (with_ids) oid.resize(n);
        sigs.resize(rowbytes * n); lv.resize(n); d0.resize(n); n0.resize(n * 2 * M); c0.resize(n * 2 * M); up.resize(n);
        dU.resize(std::max<uint64_t>(U, 1) * ML); nU.resize(std::max<uint64_t>(U, 1) * ML * M); cU.resize(std::max<uint64_t>(U, 1) * ML * M);
    } catch (const std::exception &) { fclose(f); GS_REQUIRE(false, GS_ERR_IO, "%s: not enough host memory for %llu points", path, (unsigned long long)n); }
    ok = fread(sigs.data(), rowbytes, n, f) == n && fread(lv.data(), 1, n, f) == n && fread(d0.data(), 4, n, f) == n &&
         fread(n0.data(), 4, n * 2 * M, f) == n * 2 * M && fread(c0.data(), 4, n * 2 * M, f) == n * 2 * M && fread(up.data(), 4, n, f) == n &&
         fread(dU.data(), 4, U * ML, f) == U * ML && fread(nU.data(), 4, U * ML * M, f) == U * ML * M && fread(cU.data(), 4, U * ML * M, f) == U * ML * M &&
         (!with_ids || fread(oid.data(), 8, n, f) == n);
    fclose(f);
    GS_REQUIRE(ok, GS_ERR_IO, "%s is truncated", path);
    gs_index *ix = nullptr;
    int rc = gs_index_create(c, &prm, &ix);
    if (rc) return rc;
    rc = gs_index_import(ix, sigs.data(), n, lv.data(), (int64_t)hdr[2], d0.data(), n0.data(), c0.data(), up.data(), U, dU.data(), nU.data(), cU.data());
    if (rc == GS_OK && with_ids) rc = gs_index_set_ids(ix, oid.data(), n);
    if (rc) { gs_index_destroy(ix); return rc; }
    *out = ix;
    return GS_OK;
}

}  // extern "C"
