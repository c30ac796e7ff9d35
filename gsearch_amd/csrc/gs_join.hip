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
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>
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

// rows (row-major, strided) -> columns: cols[s * colcap + first + i] = rows[i][s]   (32 x 32 LDS transpose). The column copy is private to the join, so f32
// values are stored canonical (-0 as +0); a NaN stays what it is - no table ever holds one, so it cannot match
template <typename T>
__global__ __launch_bounds__(256) void k_rows_to_cols(const uint8_t *__restrict__ rows, uint64_t stride, uint64_t nrows, uint32_t m, T *__restrict__ cols,
                                                       uint64_t colcap, uint64_t first, int f32)
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
        if (s < m && r < nrows) { const T v = tile[tx][j]; cols[s * colcap + first + r] = (f32 && v == (T)0x80000000u) ? (T)0 : v; }      // f32: -0 stored as +0 (canonical keys)
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
constexpr uint32_t HTILE = 128;   // edge of the compare tile kernel's tiles (gs_hamming.hip HT)

__device__ __forceinline__ uint32_t join_hash(uint32_t k) { return k * 0x9E3779B1u; }
__device__ __forceinline__ uint32_t join_hash(uint64_t k) { return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 32); }
#define GS_SEL4(a, i) (((i) & 2) ? (((i) & 1) ? a[3] : a[2]) : (((i) & 1) ? a[1] : a[0]))

#ifndef GS_JOIN_COUNT_KIND
#define GS_JOIN_COUNT_KIND 0      // what `natom` counts: 0 the memory-side atomics (the product); 1..6: tools/join_atomics_kinds.sh
#endif
constexpr int JN = 8;             // nodes per lane: a workgroup owns JT * JN nodes for a whole block of slots
// Barrier that only orders LDS traffic (the hash table of a slot): a __syncthreads() also waits for vmcnt(0), i.e. for every count atomic
// and column prefetch the wave has in flight - three times per slot
__device__ __forceinline__ void join_lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// orders the LDS traffic of ONE wavefront (its survivor queue): the LDS serves a wavefront's instructions in order, the compiler must too
__device__ __forceinline__ void join_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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
// CL (request batches, DESIGN.md 3.8 "heavy blocks"): queries and nodes carry the id of the heavy block ("cluster") they belong to, if any.
//   * the tag word of a table entry is query + 1 (12 bits) | JTAG_MULTI | cluster << 13;
//   * a hit of a node on an entry of ITS OWN cluster is dropped: the counters of a cluster's (query, node) pairs are written afterwards by
//     the compare tile kernel over the block (hamming_blocks) - thousands of matches per pair that would otherwise be atomics;
//   * 4-byte keys: key and tag word share one 8-byte LDS entry, so an insert sees complete entries and a query whose cluster already holds
//     this key in this slot is NOT inserted again (the entry gets JTAG_MULTI): 83 isolates of one species put one entry where they agree
//     instead of a run of 83 every probe has to walk. A hit of a FOREIGN node on a MULTI entry (a chance match) stands for a match with
//     every member that holds the key: the wavefront expands it together, lanes over the cluster's members in the cluster-sorted key copy
//     `qs` (cl_lo / qlist: member ranges and query numbers), one atomic per member found. (A first version queued these hits for a kernel
//     of its own: 1.4e7 same-address queue atomics per batch and 3e8 count atomics no longer hidden under the probes - 35 % slower.)
constexpr uint32_t JTAG_MASK = 0xFFFu, JTAG_MULTI = 0x1000u, JTAG_MORE = 0x80000000u;
constexpr int JTAG_CL_SHIFT = 13;
constexpr uint32_t JCL_MAX = 2047;      // cluster ids 1..2047
constexpr int JWALK = 4;                // table entries the in-place own-cluster test of a node's value looks at (k_match_join, CL)

template <int KIND, typename T, int SR, bool CL>
__global__ __launch_bounds__(JT, (sizeof(T) == 8 && SR == 1) ? 4 : 8) void k_match_join(const T *__restrict__ qkey, uint32_t nq, uint32_t log2p, const T *__restrict__ cols, uint64_t colcap, uint64_t n,
                                                    uint32_t slot_lo, uint32_t slot_hi, uint32_t slots_per_wg, uint32_t *__restrict__ mm32, uint64_t ld,
                                                    unsigned long long *__restrict__ stats, int chunk_major, uint64_t col0, const uint16_t *__restrict__ qcl,
                                                    const uint16_t *__restrict__ nodelab, const T *__restrict__ qs, uint32_t nh, const uint32_t *__restrict__ cl_lo,
                                                    const uint32_t *__restrict__ qlist, uint32_t dedup_flags)
{
    const uint32_t dedup_below = dedup_flags & 0x7FFFFFFFu;      // bit 31: the in-place own-cluster test (below) is on for this launch
    static_assert(JU == 4 && JN % JU == 0, "GS_SEL4 / pending mask are written for JU = 4");
    static_assert(!CL || SR == 1, "clusters: request batches only");
    constexpr bool E8 = sizeof(T) == 4 && SR == 1;               // 4-byte keys, one slot per round: key + tag word in ONE 8-byte entry (a probe step is one LDS read; an
                                                                 // insert sees complete entries, so it can flag the entries behind which ANOTHER entry holds the same key:
                                                                 // a hit on an entry without JTAG_MORE ends the probe instead of walking on to the next empty slot)
    constexpr bool DEDUP = CL && E8;                             // ... and a cluster's equal keys can share one entry
    constexpr bool D64 = sizeof(T) == 8 && SR == 1 && !CL;      // 8-byte keys (round 6): the same survivor queue; key and tag stay in separate LDS arrays (a CAS holds 8 bytes), so no
                                                                 // JTAG_MORE - a probe chain ends at the first empty entry - and a queue entry is two 8-byte words {value}, {owner | accumulator}.
                                                                 // The round-3 form it replaces for request batches ran 312 VALU lane-instructions per (slot, node) element against 64 for 4-byte
                                                                 // keys (profiles/r06_join_u64_pmc.txt): every lane of a wavefront walked the flattened state machine for one lane's survivor
    constexpr bool DENSE = E8 || D64;                            // survivors of the bitmap are compacted per wavefront and probed one per lane (below)
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
    const uint32_t P = 1u << log2p, mask = P - 1, sh = 32 - log2p;
    constexpr uint32_t BMW = (1u << JB_LOG2) / 32;                // bitmap words per slot
    uint32_t *tag = (uint32_t *)s_raw;                            // [SR][P]            (DEDUP: ent[P] of {key, tag word} first, then the bitmap)
    uint32_t *bm = E8 ? (uint32_t *)(s_raw + 8 * (size_t)P) : (uint32_t *)(s_raw + 4 * (size_t)P * SR);      // [SR][BMW]
    T *key = (T *)(s_raw + (4 * (size_t)P + (size_t)BMW * 4) * SR);   // [SR][P]
    unsigned long long *ent = (unsigned long long *)s_raw;
    // chunk_major: consecutive workgroups sweep the slot blocks of ONE node chunk, so the counters being updated at any time are
    // those of a few chunks (a window of the count matrix that fits the 256 MB Infinity Cache) instead of all of them
    const uint32_t bchunk = chunk_major ? blockIdx.y : blockIdx.x, bslot = chunk_major ? blockIdx.x : blockIdx.y;
    const uint64_t e0 = (uint64_t)bchunk * (JT * JN) + threadIdx.x;      // the lane's nodes: e0 + i * JT, i < JN
    const uint32_t s0 = slot_lo + bslot * slots_per_wg, s1 = s0 + slots_per_wg < slot_hi ? s0 + slots_per_wg : slot_hi;
    // CL: the cluster ids of the lane's nodes come from a copy of the node labels laid out like the lanes' nodes ([chunk][it][lane][JU] of 16 bits:
    // one 8-byte load per lane and group of four nodes), fetched again for every slot beside the column values - all eight labels resident would
    // cost four more registers than the 64 this kernel may use, and a look-up at hit time stalled the wavefront on every own-cluster hit (round 4:
    // 201 ms instead of 126 per request when a fifth of the nodes belong to clusters)
    const uint2 *labT = CL ? (const uint2 *)nodelab + (uint64_t)bchunk * (JN / JU) * JT : nullptr;      // (workgroup-uniform: + the lane at each use)
    uint32_t sticky[JN];
    uint32_t natom = 0;                                           // memory-side atomics this lane sends (work counter for the bench's roofline)
    uint32_t nexp = 0;                                            // CL: chance matches on shared entries this wavefront expanded
#pragma unroll
    for (int i = 0; i < JN; i++) sticky[i] = 0;
    // DENSE: the lane's node validity as a mask, its first node's matrix column, and the wavefront's survivor queue
    const uint32_t lane = threadIdx.x & 63;
    uint32_t vmask = 0;
#pragma unroll
    for (int u = 0; u < JN; u++) vmask |= (uint32_t)(e0 + (uint64_t)u * JT < n) << u;
    // (the matrix column of a node = a workgroup-uniform base + a 32-bit lane part, added where an atomic is sent: a per-lane 64-bit base lived through the whole slot loop)
    const uint64_t ebase_u = col0 + (uint64_t)bchunk * (JT * JN);
    const uint32_t tid = threadIdx.x;
    uint32_t labmask = 0;                                         // CL: which of the lane's nodes belong to a cluster at all (labels do not change from slot to slot)
    if (CL && DENSE) {
#pragma unroll
        for (int it = 0; it < JN / JU; it++) {
            const uint2 l4 = labT[it * JT + threadIdx.x];
            labmask |= ((uint32_t)((l4.x & 0xFFFFu) != 0u) | (uint32_t)((l4.x >> 16) != 0u) << 1 | (uint32_t)((l4.y & 0xFFFFu) != 0u) << 2 | (uint32_t)((l4.y >> 16) != 0u) << 3) << (it * JU);
        }
    }
    const bool wave_lab = CL && DENSE && (dedup_flags >> 31) && __any(labmask != 0u);   // (wavefront-uniform) some node of this wavefront belongs to a cluster
    uint2 *wq = (uint2 *)(s_raw + (D64 ? 12 : 8) * (size_t)P + (size_t)BMW * 4) + (threadIdx.x >> 6) * (D64 ? 128 : 64);      // D64: entry i = words 2i (value), 2i + 1 (owner | accumulator)
    T vn[DENSE ? JN : JU];
#pragma unroll
    for (int u = 0; u < (DENSE ? JN : JU); u++) { const uint64_t e = e0 + (uint64_t)u * JT; vn[u] = (e < n && s0 < s1) ? cols[(uint64_t)s0 * colcap + e] : (T)0; }
    for (uint32_t sr = s0; sr < s1; sr += SR) {
        const uint32_t nr = s1 - sr < (uint32_t)SR ? s1 - sr : (uint32_t)SR;       // slots of this round
        // one-slot rounds: the lane's query keys of this slot (at most JQ: nq <= 0.4 * 2^JP_MAX_LOG2) are requested together and before the barriers -
        // loaded one by one inside the insert loop, each cost the workgroup a trip to memory with nothing else to do
        // (scalar bases + an offset the compiler cannot hoist: kept across the slot loop, the four per-lane offsets were spilled and every reload waited)
        constexpr int JQ = DENSE ? (int)((((1u << JP_MAX_LOG2) * 2) / 5 + JT - 1) / JT) : 1;
        T kq[JQ];
        uint32_t tq[(JQ + 1) / 2] = {};
        if (DENSE) {
            typedef const T __attribute__((address_space(1))) *gptr;
            const uint64_t qb = (uint64_t)(qkey + (uint64_t)sr * nq);
            const gptr qk = (gptr)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(qb >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)qb));
            uint32_t o = threadIdx.x;
            asm volatile("" : "+v"(o));
#pragma unroll
            for (int j = 0; j < JQ; j++) { const gptr qj = qk + j * JT; kq[j] = o + (uint32_t)j * JT < nq ? qj[o] : (T)0; }
            // CL: the cluster ids of the lane's queries, requested with the keys (two per register). Read inside insert_key they were four global loads per slot, each
            // waited for where it stood: ~15 VMEM wave-instructions per wavefront and slot more than the plain kernel (profiles/r06_join_cl_pmc_bench.txt)
            if constexpr (CL) {
#pragma unroll
                for (int j = 0; j < JQ; j++) { const uint32_t i = o + (uint32_t)j * JT; tq[j >> 1] |= (i < nq ? (uint32_t)qcl[i] : 0u) << ((j & 1) * 16); }
            }
        }
        join_lds_barrier();                                       // the previous round's probes are done
        if (E8) { for (uint32_t i = threadIdx.x; i < P; i += JT) ent[i] = 0ull; }
        else for (uint32_t i = threadIdx.x; i < P * SR; i += JT) tag[i] = 0;
        for (uint32_t i = threadIdx.x; i < BMW * SR; i += JT) bm[i] = 0;
        join_lds_barrier();
        auto insert_key = [&](const uint32_t r, const uint32_t q, T k, const uint32_t clq) {
            const uint32_t tw = CL ? (q + 1) | (clq << JTAG_CL_SHIFT) : q + 1;
            if (never_equal<KIND, T>(k)) return;
            k = canon<KIND, T>(k);
            const uint32_t hq = join_hash(k);
            atomicOr(&bm[r * BMW + (hq >> (32 - JB_LOG2 + 5))], (1u << ((hq >> (32 - JB_LOG2)) & 31)) | (1u << ((hq >> (27 - JB_LOG2)) & 31)));      // two bits of ONE word
            uint32_t h = hq >> sh;
            if (E8) {
                const unsigned long long nw = ((unsigned long long)tw << 32) | (uint32_t)k;
                for (;;) {
                    const unsigned long long old = atomicCAS(&ent[h], 0ull, nw);
                    if (old == 0ull) break;
                    // same key already entered by a query of the same cluster: that entry stands for this query too
                    // (only clusters numbered below dedup_below share entries - all of them by default, JDEDUP_MINQ)
                    if (DEDUP && (tw >> JTAG_CL_SHIFT) - 1u < dedup_below - 1u && (uint32_t)old == (uint32_t)k &&
                        (((uint32_t)(old >> 32) >> JTAG_CL_SHIFT) & JCL_MAX) == (tw >> JTAG_CL_SHIFT)) {
                        atomicOr((uint32_t *)&ent[h] + 1, JTAG_MULTI);
                        break;
                    }
                    // this key again, further down the run: the entry passed here is not the last one with it
                    if ((uint32_t)old == (uint32_t)k && !((uint32_t)(old >> 32) & JTAG_MORE)) atomicOr((uint32_t *)&ent[h] + 1, JTAG_MORE);
                    h = (h + 1) & mask;
                }
            } else {
                while (atomicCAS(&tag[r * P + h], 0u, tw) != 0u) h = (h + 1) & mask;
                key[r * P + h] = k;
            }
        };
        if (DENSE) {
#pragma unroll
            for (int j = 0; j < JQ; j++) { const uint32_t i = threadIdx.x + (uint32_t)j * JT; if (i < nq) insert_key(0u, i, kq[j], (tq[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu); }
        } else {
            for (uint32_t i = threadIdx.x; i < nq * nr; i += JT) { const uint32_t r = i / nq, q = i - r * nq; insert_key(r, q, qkey[(uint64_t)(sr + r) * nq + q], CL ? (uint32_t)qcl[q] : 0u); }
        }
        // CL: the cluster ids of the lane's eight nodes, fetched again for every slot (16 kB per workgroup: L2) so that they occupy registers only between the
        // inserts and the push - the request is in flight across the barrier (opaque offsets: hoisted out of the slot loop they would live, and spill, through it)
        uint2 lb0 = make_uint2(0u, 0u), lb1 = make_uint2(0u, 0u);
        if constexpr (CL && DENSE) {
            if (wave_lab) { uint32_t o0 = threadIdx.x, o1 = JT + threadIdx.x; asm volatile("" : "+v"(o0), "+v"(o1)); lb0 = labT[o0]; lb1 = labT[o1]; }
        }
        join_lds_barrier();
#pragma unroll 1
        for (uint32_t r = 0; r < nr; r++) {
            const uint32_t s = sr + r, tb = r * P, bb = r * BMW;
            const T *col = cols + (uint64_t)s * colcap;
            if constexpr (DENSE) {
                // The survivors of the bitmap are ~7 % of the values (6 % chance matches of unrelated genomes, 1 % false positives): probed in place, a
                // wavefront runs the probe loop for its lane with the longest chain while a tenth of its lanes work. Instead every survivor is pushed
                // to a 64-entry queue of the wavefront ({value, owner lane | value index | the tag its node's accumulator holds}), lanes 0..count-1
                // probe ONE value each (static registers, no selects), send every hit but one straight to memory and hand one tag - the accumulator's
                // own if it is among the hits, otherwise the first - back through the queue slot to the owner, whose accumulator logic then runs on a
                // statically indexed register. Values are queued whole (all lanes that want value u, or none), so a burst - a related query makes 64
                // consecutive nodes match in one slot - takes another trip of the loop.
                uint32_t pend = 0;
#pragma unroll
                for (int u0 = 0; u0 < JN; u0 += 4) {                 // (four bitmap words in flight at a time; the columns hold canonical values)
                    uint32_t hsd[4], bwd[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { hsd[u] = join_hash(vn[u0 + u]); bwd[u] = bm[hsd[u] >> (32 - JB_LOG2 + 5)]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t pass = (vmask >> (u0 + u)) & (bwd[u] >> ((hsd[u] >> (32 - JB_LOG2)) & 31)) & (bwd[u] >> ((hsd[u] >> (27 - JB_LOG2)) & 31)) & 1u;
                        pend |= pass << (u0 + u);
                    }
                }
                if constexpr (CL) {
                    // Own-cluster hits end HERE when the table says so in one read (round 6, skewed databases: 72 % of the nodes belong to a cluster and two thirds of their
                    // values pass the bitmap - their own cluster's queries hold them - so the survivor path ran for 47 % of all values instead of 7 %): the entry at the
                    // value's home position holds this key, belongs to the node's cluster and is the ONLY entry with the key (no JTAG_MORE; a cluster's equal keys share
                    // one entry) - every hit of this value is one the block compare writes, exactly what the probe below would find and drop.
                    if (wave_lab) {
#pragma unroll
                        for (int u = 0; u < JN; u++) {
                            const uint32_t l2 = u < 4 ? ((u & 2) ? lb0.y : lb0.x) : ((u & 2) ? lb1.y : lb1.x);
                            const uint32_t lab = (l2 >> ((u & 1) * 16)) & 0xFFFFu;
                            if (((pend >> u) & 1u) && lab != 0u) {
                                // (up to JWALK entries: the key may sit behind its home position - a collision - and clusters below GS_JOIN_DEDUP_MINQ queries, if that is
                                // raised, enter their equal keys one by one; a foreign entry with the key, or a longer walk, leaves the value to the probe)
                                uint32_t hh = join_hash(vn[u]) >> sh;
#pragma unroll
                                for (int j = 0; j < JWALK; j++) {
                                    const unsigned long long en = ent[hh];
                                    const uint32_t t = (uint32_t)(en >> 32);
                                    if (t == 0u) { pend &= ~(1u << u); natom += GS_JOIN_COUNT_KIND == 8; break; }      // the end of the run: nobody else holds this value
                                    if ((uint32_t)en == (uint32_t)vn[u]) {
                                        if (((t >> JTAG_CL_SHIFT) & JCL_MAX) != lab) break;                            // a foreign (or unclustered) query holds it too
                                        if (!(t & JTAG_MORE)) { pend &= ~(1u << u); natom += GS_JOIN_COUNT_KIND == 8; break; }
                                    }
                                    hh = (hh + 1) & mask;
                                }
                            }
                        }
                    }
                }
                while (__any(pend != 0u)) {
                    uint32_t qn = 0;
                    const uint32_t pend0 = pend;
#pragma unroll
                    for (int u = 0; u < JN; u++) {
                        const bool want = (pend >> u) & 1u;
                        const unsigned long long mw = __ballot(want);
                        const uint32_t cw = (uint32_t)__popcll(mw);
                        if (cw != 0u && qn + cw <= 64u) {
                            if (want) {
                                const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(mw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mw, 0u));
                                const uint32_t who = lane | ((uint32_t)u << 6) | (sticky[u] << 9) | (CL ? ((labmask >> u) & 1u) << 29 : 0u);
                                if constexpr (D64) { wq[2 * at] = make_uint2((uint32_t)vn[u], (uint32_t)((uint64_t)vn[u] >> 32)); wq[2 * at + 1].x = who; }
                                else wq[at] = make_uint2((uint32_t)vn[u], who);
                            }
                            qn += cw; pend &= ~(1u << u);
                        }
                    }
                    join_wave_sync();
                    if constexpr (CL) {
                        if (GS_JOIN_COUNT_KIND == 7 && lane < qn) natom++;                              // (instrumented builds: survivors that reach the probe)
                        // clusters: the same probe, but the wavefront stays together until its last lane is done, so that all 64 lanes can expand a
                        // chance match on a shared entry. The node's own cluster is looked up (L2) only when the node has one AND the entry hit
                        // belongs to a cluster: a percent of the nodes.
                        bool have = lane < qn;
                        uint2 qe = make_uint2(0u, 0u);
                        if (have) qe = wq[lane];
                        uint32_t st = (qe.y >> 9) & 0xFFFFFu, flp = 0;
                        const uint32_t eloc = (qe.y & 63u) + ((qe.y >> 6) & 7u) * JT;                   // the owner's node, relative to the wavefront's first
                        uint32_t hh = join_hash((T)qe.x) >> sh;
                        for (;;) {
                            uint32_t mitem = 0;                                                         // cluster of this lane's pending expansion
                            if (have) {
                                const unsigned long long en = ent[hh];
                                const uint32_t t = (uint32_t)(en >> 32);
                                if (t == 0u) have = false;
                                else {
                                    if ((uint32_t)en == qe.x) {
                                        if (!(t & JTAG_MORE)) have = false;                             // the last entry with this key
                                        bool count_it = true;
                                        if ((t >> 12) & 0x7FFFFu) {                                     // entry of a cluster
                                            const uint32_t cl = (t >> JTAG_CL_SHIFT) & JCL_MAX;
                                            uint32_t nl = 0;
                                            if ((qe.y >> 29) & 1u) {
                                                const uint32_t uo = (qe.y >> 6) & 7u, to = threadIdx.x - lane + (qe.y & 63u);
                                                nl = nodelab[(((uint64_t)bchunk * (JN / JU) + (uo >> 2)) * JT + to) * JU + (uo & 3u)];
                                            }
                                            if (nl == cl) { count_it = false; natom += GS_JOIN_COUNT_KIND == 9; }      // own cluster: the block compare writes this pair's counter
                                            else if (t & JTAG_MULTI) { mitem = cl; count_it = false; }
                                        }
                                        if (count_it) {
                                            const uint32_t tg = t & JTAG_MASK, one = tg | 0x1000u;
                                            const bool same = (st & 0xFFFu) == tg, weak = st < 0x2000u, full = st >= 0xFF000u;
                                            const uint32_t fl = same ? (full ? st : 0u) : (weak ? st : one);
                                            st = same ? (full ? one : st + 0x1000u) : (weak ? one : st);
                                            if (fl) {
                                                if (flp) { join_flush(mm32, ld, flp, ebase_u + (tid - lane + eloc)); natom += GS_JOIN_COUNT_KIND == 0; }
                                                flp = fl;
                                            }
                                        }
                                    }
                                    hh = (hh + 1) & mask;
                                }
                            }
                            unsigned long long pm = __ballot(mitem != 0u);
                            while (pm) {
                                const int src = __ffsll((long long)pm) - 1;
                                pm &= pm - 1;
                                nexp++;
                                const uint32_t cl = __builtin_amdgcn_readlane(mitem, src), vsrc = __builtin_amdgcn_readlane(qe.x, src);
                                const uint64_t e = ebase_u + (tid - lane + __builtin_amdgcn_readlane(eloc, src));
                                const uint32_t lo = cl_lo[cl], hi = cl_lo[cl + 1];
                                for (uint32_t pos = lo + lane; pos < hi; pos += 64) {
                                    const T k2 = qs[(uint64_t)s * nh + pos];
                                    if (!never_equal<KIND, T>(k2) && (uint32_t)canon<KIND, T>(k2) == vsrc) {
                                        const uint64_t idx = (uint64_t)qlist[pos] * ld + e;
                                        atomicSub(&mm32[idx >> 1], 1u << ((idx & 1) * 16));
                                        natom += GS_JOIN_COUNT_KIND == 0;
                                    }
                                }
                            }
                            if (!__any(have)) break;
                        }
                        if (lane < qn) {
                            if (flp) { join_flush(mm32, ld, flp, ebase_u + (tid - lane + eloc)); natom += GS_JOIN_COUNT_KIND == 0; }
                            wq[lane].x = st;
                        }
                    } else if constexpr (D64) {
                        if (lane < qn) {
                            const uint2 q0 = wq[2 * lane];
                            const uint32_t who = wq[2 * lane + 1].x;
                            const T val = (T)((uint64_t)q0.x | ((uint64_t)q0.y << 32));
                            uint32_t st = who >> 9, flp = 0;
                            const uint32_t eo = (who & 63u) + ((who >> 6) & 7u) * JT - lane;
                            uint32_t hh = join_hash(val) >> sh;
                            for (;;) {
                                const uint32_t t = tag[hh];
                                if (t == 0u) break;
                                if (key[hh] == val) {
                                    const uint32_t tg = t & JTAG_MASK, one = tg | 0x1000u;
                                    const bool same = (st & 0xFFFu) == tg, weak = st < 0x2000u, full = st >= 0xFF000u;
                                    const uint32_t fl = same ? (full ? st : 0u) : (weak ? st : one);
                                    st = same ? (full ? one : st + 0x1000u) : (weak ? one : st);
                                    if (fl) {
                                        if (flp) { join_flush(mm32, ld, flp, ebase_u + (tid + eo)); natom++; }
                                        flp = fl;
                                    }
                                }
                                hh = (hh + 1) & mask;
                            }
                            if (flp) { join_flush(mm32, ld, flp, ebase_u + (tid + eo)); natom++; }
                            wq[2 * lane + 1].x = st;
                        }
                    } else
                    if (lane < qn) {
                        const uint2 qe = wq[lane];
                        uint32_t st = qe.y >> 9, flp = 0;                                   // the node's accumulator: tag | run << 12, run <= 255
                        const uint32_t eo = (qe.y & 63u) + ((qe.y >> 6) & 7u) * JT - lane;  // the owner's node, relative to this lane's first
                        uint32_t hh = join_hash((T)qe.x) >> sh;
                        for (;;) {
                            const unsigned long long en = ent[hh];
                            const uint32_t t = (uint32_t)(en >> 32);
                            if (t == 0u) break;
                            if ((uint32_t)en == qe.x) {
                                // the accumulator rule of GS_JOIN_HIT as selects; what it sends to memory waits in flp - nearly every chain sends at
                                // most one, so the address arithmetic and the atomic sit once behind the loop
                                const uint32_t tg = t & JTAG_MASK, one = tg | 0x1000u;
                                const bool same = (st & 0xFFFu) == tg, weak = st < 0x2000u, full = st >= 0xFF000u;
                                const uint32_t fl = same ? (full ? st : 0u) : (weak ? st : one);
                                if (GS_JOIN_COUNT_KIND == 1 && !same && !weak) natom++;       // (instrumented builds only: what the atomics are made of)
                                if (GS_JOIN_COUNT_KIND == 2 && !same && weak && st) natom++;
                                if (GS_JOIN_COUNT_KIND == 5) natom++;
                                if (GS_JOIN_COUNT_KIND == 6 && same) natom++;
                                st = same ? (full ? one : st + 0x1000u) : (weak ? one : st);
                                if (fl) {
                                    if (flp) { join_flush(mm32, ld, flp, ebase_u + (tid + eo)); natom += GS_JOIN_COUNT_KIND == 0; }
                                    flp = fl;
                                }
                                if (!(t & JTAG_MORE)) break;                                // the last entry with this key
                            }
                            hh = (hh + 1) & mask;
                        }
                        if (flp) { join_flush(mm32, ld, flp, ebase_u + (tid + eo)); natom += GS_JOIN_COUNT_KIND == 0; }
                        wq[lane].x = st;
                    }
                    join_wave_sync();
                    // the owners take their accumulators back: queue positions are those of the push (the same ballots in the same order)
                    const uint32_t queued = pend0 & ~pend;
                    uint32_t qb = 0;
#pragma unroll
                    for (int u = 0; u < JN; u++) {
                        const bool mine = (queued >> u) & 1u;
                        const unsigned long long mw = __ballot(mine);
                        if (mw != 0ull) {
                            if (mine) { const uint32_t at = qb + __builtin_amdgcn_mbcnt_hi((uint32_t)(mw >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mw, 0u)); sticky[u] = D64 ? wq[2 * at + 1].x : wq[at].x; }
                            qb += (uint32_t)__popcll(mw);
                        }
                    }
                    join_wave_sync();
                }
                // the next slot's values arrive while its table is built
                if (s + 1 < s1) {
                    // (the base through readfirstlane: the compiler then addresses with a scalar base + the lane's 32-bit offset instead of keeping a
                    // per-lane 64-bit pointer alive across the slot loop - which it spilled, and whose reload waits for every atomic in flight)
                    const uint64_t cb = (uint64_t)(col + colcap + (uint64_t)bchunk * (JT * JN));
                    typedef const T __attribute__((address_space(1))) *gptr;       // (an integer cast alone would leave a flat pointer)
                    const gptr cn = (gptr)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(cb >> 32)) << 32) | (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)cb));      // (the builtin returns int)
#pragma unroll
                    // (nontemporal: the column store - 21.6 GB per batch at 300 k nodes - is read once; as ordinary loads it flushes the count-matrix window the
                    // atomics work on out of the 256 MB Infinity Cache. With the chunk-major launch order below, whose live window is a few node chunks' counters,
                    // 19.3 -> 18.6 ms per 2500-query batch; either alone 19.1 / 19.2: profiles/r06_join_nt_chunk_major.txt)
                    for (int u = 0; u < JN; u++) { const gptr cu = cn + u * JT; vn[u] = ((vmask >> u) & 1u) ? __builtin_nontemporal_load(cu + threadIdx.x) : (T)0; }
                }
            } else {
#pragma unroll
            for (int it = 0; it < JN / JU; it++) {
                uint2 lab4 = make_uint2(0u, 0u);
                if (CL) { uint32_t o = it * JT + threadIdx.x; asm volatile("" : "+v"(o)); lab4 = labT[o]; }      // (opaque offset: reloaded per slot, not kept live)
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
                    const uint32_t pass = (uint32_t)(e < n) & (uint32_t)!never_equal<KIND, T>(vn[u]) & (bw[u] >> (bit & 31)) & (bw[u] >> ((hs[u] >> (27 - JB_LOG2)) & 31)) & 1u;
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
                {
                for (;;) {
                    if (!have && pend) { uu = (uint32_t)__ffs((int)pend) - 1; pend &= pend - 1; vv = GS_SEL4(v, uu); hh = GS_SEL4(hs, uu) >> sh; have = true; }
                    if (!have) break;
                    uint32_t t; T k;
                    if (E8) { const unsigned long long en = ent[hh]; t = (uint32_t)(en >> 32); k = (T)(uint32_t)en; }
                    else { t = tag[tb + hh]; k = key[tb + hh]; }
                    if (t == 0u) { have = false; continue; }
                    if (k == vv) {
                        if (E8 && !(t & JTAG_MORE)) have = false;                            // the last entry with this key
                        const uint32_t ni = it * JU + uu;
                        const uint64_t el = e0 + (uint64_t)ni * JT;                          // node within this column range
                        bool count_it = true;
                        if (CL && ((t >> 12) & 0x7FFFFu))                                    // (8-byte keys: no shared entries, only the own-cluster rule)
                            count_it = (((uu & 2u ? lab4.y : lab4.x) >> ((uu & 1u) * 16)) & 0xFFFFu) != ((t >> JTAG_CL_SHIFT) & JCL_MAX);
                        if (count_it) {
                            const uint64_t e = col0 + el;                                    // column of the count matrix (col0: the node range starts there)
                            uint32_t st = GS_SEL4((sticky + it * JU), uu);
                            GS_JOIN_HIT(st, t & JTAG_MASK, e);
#pragma unroll
                            for (int u = 0; u < JU; u++) if (uu == (uint32_t)u) sticky[it * JU + u] = st;
                        }
                    }
                    hh = (hh + 1) & mask;
                }
                }
            }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < JN; i++) if (sticky[i]) { join_flush(mm32, ld, sticky[i], ebase_u + (tid + (uint32_t)i * JT)); natom += GS_JOIN_COUNT_KIND == 0 || (GS_JOIN_COUNT_KIND == 3 && sticky[i] >= 0x2000u) || (GS_JOIN_COUNT_KIND == 4 && sticky[i] < 0x2000u); }
    if (stats) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) natom += __shfl_down(natom, o);
        if ((threadIdx.x & 63) == 0 && natom) atomicAdd(stats, (unsigned long long)natom);
        if (CL && (threadIdx.x & 63) == 0 && nexp) atomicAdd(stats + 7, (unsigned long long)nexp);
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

// ---- heavy blocks ("clusters") of a request batch ---------------------------------------------------------------------------------
// gsearch's stated databases (GTDB / NCBI prokaryotes, /root/reference/README.md:134) hold thousands of near-identical genomes per
// species, and a request often is many isolates of a few species: every such query matches every database genome of its species in
// thousands of slots. Recording those matches one atomic at a time (and walking, per probe, the run of equal keys 83 isolates leave in
// the table) is what made such batches 4x slower than unrelated ones. Instead:
//   phase 0  the join proper over the first JS0 slots only;
//   scan     the (query, node) pairs whose counter already lost >= JHEAVY matches are the heavy pairs (chance: < 1e-6 per pair);
//   labels   connected components of the heavy pairs by min-label propagation (a few rounds; ANY labelling is correct, see below);
//   host     components with >= 2 queries become clusters: cluster id per query / per node, member lists, 128 x 128 tiles;
//   main     the join over the remaining slots with CL = true (own-cluster hits dropped, equal keys of a cluster entered once, chance
//            matches on such shared entries expanded over the cluster's members by the wavefront);
//   blocks   the compare tile kernel over every cluster's queries x nodes block WRITES those counters (all m slots).
// Correctness does not depend on how the labels were found: a pair is either counted match by match (phase 0 + main) or its
// counter is overwritten by the exact compare - "own cluster" in the main pass and "member of the block" in the last are the same test.
constexpr uint32_t JS0 = 48, JHEAVY = 3;
constexpr uint64_t JPAIR_CAP = (uint64_t)4 << 20, JPAIR_MAX = (uint64_t)256 << 20;      // heavy pairs listed by default (32 MB) / at most (2 GB)
constexpr uint32_t JTILE_CAP = 24576;
constexpr uint32_t JDEDUP_MINQ = 2;      // clusters of at least this many queries enter equal keys once (shared entries): all of them since round 6 (GS_JOIN_DEDUP_MINQ).
                                         // Round 4 kept the small ones apart (12): a chance match on a shared entry is expanded by the whole wavefront. Measured on a skewed
                                         // database (tools/skew_probe.py, 2500 x 100 k, 197 clusters of 8.8 queries on average): such expansions are rare - 35 189 per batch at
                                         // 2 against 3 471 at 12 - while the runs of equal keys the small clusters left behind were walked by every own-cluster survivor:
                                         // cluster-aware join 19.3 ms at 12, 16.5 at 8, 11.4 at 4, 9.4 at 2 (with the in-place own-cluster test of k_match_join)

// tiles of 8 consecutive counters per lane x 256 lanes; a workgroup walks many tiles, stages the heavy pairs it finds in LDS and sends them to
// the list in blocks (a first version paid one same-address global atomic per wavefront and tile: 1.1e6 of them, 10 ms per batch)
constexpr uint32_t HS_STAGE = 3072;
__global__ __launch_bounds__(256) void k_heavy_scan(const uint16_t *__restrict__ mat, uint64_t ld, uint32_t nq, uint64_t n, uint32_t m, uint2 *__restrict__ pairs,
                                                    unsigned long long *__restrict__ ctr /* [0] pairs, [1] matches so far */, uint64_t cap)
{
    __shared__ uint2 stage[HS_STAGE];
    __shared__ uint32_t sn;
    __shared__ unsigned long long sbase;
    if (threadIdx.x == 0) sn = 0;
    __syncthreads();
    const uint64_t tiles_per_row = (n + 2047) / 2048, ntiles = tiles_per_row * nq;
    uint32_t tot = 0;
    auto flush = [&]() {                       // all lanes; sn pairs -> the global list
        __syncthreads();
        const uint32_t cnt = sn;
        if (cnt) {
            if (threadIdx.x == 0) sbase = atomicAdd(&ctr[0], (unsigned long long)cnt);
            __syncthreads();
            const unsigned long long base = sbase;
            for (uint32_t i = threadIdx.x; i < cnt; i += 256) if (base + i < cap) pairs[base + i] = stage[i];
            __syncthreads();
            if (threadIdx.x == 0) sn = 0;
        }
        __syncthreads();
    };
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t q = (uint32_t)(t / tiles_per_row);
        const uint64_t e8 = (t - (uint64_t)q * tiles_per_row) * 2048 + (uint64_t)threadIdx.x * 8;
        if (e8 < n) {
            const uint4 w = *(const uint4 *)(mat + (uint64_t)q * ld + e8);        // ld is a multiple of 8 and the padding columns hold m
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t c = (ww[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
                const uint32_t def = (e8 + i < n) ? m - c : 0u;
                tot += def;
                if (def >= JHEAVY) stage[atomicAdd(&sn, 1u)] = make_uint2((uint32_t)(e8 + i), q | (def << 16));
            }
        }
        __syncthreads();
        if (sn >= HS_STAGE - 2048) flush();                                       // (uniform: sn is read after the barrier)
    }
    flush();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o);
    if ((threadIdx.x & 63) == 0 && tot) atomicAdd(&ctr[1], (unsigned long long)tot);
}
__global__ void k_label_init(uint32_t *__restrict__ labq, uint32_t nq, uint32_t *__restrict__ labe, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) labq[i] = (uint32_t)i;
    if (i < n) labe[i] = 0xFFFFFFFFu;
}
__global__ void k_label_prop(const uint2 *__restrict__ pairs, const unsigned long long *__restrict__ ctr, uint64_t cap, uint32_t *__restrict__ labq, uint32_t *__restrict__ labe)
{
    const uint64_t np = ctr[0] < cap ? ctr[0] : cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 p = pairs[i];
        const uint32_t q = p.y & 0xFFFFu, a = labq[q], b = labe[p.x], l = a < b ? a : b;
        if (l < a) atomicMin(&labq[q], l);
        if (l < b) atomicMin(&labe[p.x], l);
    }
}
// nodes per component label (labels are query numbers): what the host needs to size the components without the 1.2 MB of node labels
__global__ void k_label_count(const uint32_t *__restrict__ labe, uint64_t n, uint32_t nq, uint32_t *__restrict__ cnt)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) { const uint32_t l = labe[e]; if (l < nq) atomicAdd(&cnt[l], 1u); }
}
// cluster-sorted copy of the keys of the clustered queries: qs[s * nh + pos] = row qlist[pos], slot s
template <typename T>
__global__ __launch_bounds__(256) void k_query_cols_list(const uint8_t *__restrict__ rows, uint64_t stride, const uint32_t *__restrict__ qlist, uint32_t nh, uint32_t m, T *__restrict__ qs)
{
    __shared__ T tile[32][33];
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const uint64_t r0 = (uint64_t)blockIdx.x * 32, s0 = (uint64_t)blockIdx.y * 32;
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint64_t r = r0 + j, s = s0 + tx;
        tile[j][tx] = (r < nh && s < m) ? ((const T *)(rows + (uint64_t)qlist[r] * stride))[s] : (T)0;
    }
    __syncthreads();
    for (uint32_t j = ty; j < 32; j += 8) {
        const uint64_t s = s0 + j, r = r0 + tx;
        if (s < m && r < nh) qs[s * nh + r] = tile[tx][j];
    }
}
struct JoinGeom { uint32_t chunks, log2p; size_t lds1, ldsq; };

template <int KIND, typename T, bool CL>
static int join_launch(gs_ctx *c, const JoinGeom &g, const T *qkey, uint32_t nq, const void *cols, uint64_t colcap, uint64_t n, uint32_t slot_lo, uint32_t slot_hi,
                       uint16_t *out16, uint64_t ld, unsigned long long *stats, uint64_t col0, bool few_blocks, const uint16_t *qcl, const uint16_t *nodelab, const T *qs,
                       uint32_t nh, const uint32_t *cl_lo, const uint32_t *qlist, uint32_t dedup_below = 0)
{
    // one workgroup = JT * JN nodes x a block of slots; blocks sized so that the grid is about eight rounds of 2 workgroups per CU
    // (one round leaves the slowest workgroup's tail exposed: 145 -> 125 ms per 10 k-query request), at least 32 slots each
    const uint32_t ms = slot_hi - slot_lo, chunks = g.chunks;
    uint32_t blocks = std::max<uint32_t>(1, std::min<uint32_t>((16 * c->n_cu) / chunks, std::max<uint32_t>(ms / 32, (2 * c->n_cu) / chunks)));
    if (few_blocks) blocks = std::max<uint32_t>(1, std::min<uint32_t>(ms / 4, (2 * c->n_cu + chunks - 1) / chunks));        // phase 0: one round of short blocks
    if (getenv("GS_JOIN_BLOCKS") && !few_blocks) blocks = (uint32_t)atoi(getenv("GS_JOIN_BLOCKS"));
    blocks = std::min<uint32_t>(std::max<uint32_t>(blocks, 1), ms);
    const uint32_t slots_per_wg = (ms + blocks - 1) / blocks;
    // chunk-major: consecutive workgroups sweep the slot blocks of ONE node chunk, so the counters under update at any time are those of a few chunks - with the
    // nontemporal column loads of the kernel that window stays in the Infinity Cache (request batches; an insert batch's matrix is small either way)
    // (not for a handful of chunks: the workgroups of one chunk then are a few rounds of the device with a tail each - 50 000 nodes, 7 chunks, 8-byte keys: 47.8 -> 56.1 ms)
    const int chunk_major = getenv("GS_JOIN_CHUNK_MAJOR") ? atoi(getenv("GS_JOIN_CHUNK_MAJOR")) : ((nq >= 1024 && chunks >= 16) ? 1 : 0);
    const uint32_t nblk = (ms + slots_per_wg - 1) / slots_per_wg;
    dim3 jg(chunk_major ? nblk : chunks, chunk_major ? chunks : nblk);
    // small tables (an insert batch): several slots per barrier round while two workgroups still fit a CU
    int sr = 1;
    if (!CL) {
        if (4 * g.lds1 <= 80 * 1024 && slots_per_wg >= 8) sr = 4; else if (2 * g.lds1 <= 80 * 1024 && slots_per_wg >= 4) sr = 2;
        if (getenv("GS_JOIN_SLOTS_PER_ROUND")) { const int e = atoi(getenv("GS_JOIN_SLOTS_PER_ROUND")); if (e == 1 || (e == 2 && 2 * g.lds1 <= 80 * 1024) || (e == 4 && 4 * g.lds1 <= 80 * 1024)) sr = e; }
    }
    const size_t lds = g.lds1 * sr + (sr == 1 ? g.ldsq : 0);
    ProfScope ps(c, FAM_HAMMING);
#define GS_JOIN_GO(KERN)                                                                                                                              \
    do {                                                                                                                                              \
        auto kern = KERN;                                                                                                                             \
        if (lds > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));             \
        hipLaunchKernelGGL(kern, jg, dim3(JT), lds, c->stream, qkey, nq, g.log2p, (const T *)cols, colcap, n, slot_lo, slot_hi, slots_per_wg, (uint32_t *)out16, ld, \
                           stats, chunk_major, col0, qcl, nodelab, qs, nh, cl_lo, qlist, dedup_below);                                                             \
    } while (0)
    if (CL) GS_JOIN_GO((k_match_join<KIND, T, 1, CL>));
    else if (sr == 4) GS_JOIN_GO((k_match_join<KIND, T, 4, false>));
    else if (sr == 2) GS_JOIN_GO((k_match_join<KIND, T, 2, false>));
    else GS_JOIN_GO((k_match_join<KIND, T, 1, false>));
#undef GS_JOIN_GO
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// the sampled decision of the unclustered join: 1 = hand the batch to the compare tile kernel
template <int KIND, typename T>
static int join_sample_declines(gs_ctx *c, const JoinGeom &g, uint32_t m, const T *qkey, uint32_t nq, const void *cols, uint64_t colcap, uint64_t n, DevBuf *scratch, int *decline_out)
{
    // sampled match density -> estimated join time (column stream + one memory-side atomic per match the accumulator does not
    // absorb, 1.6e10/s: profiles/r01_match_join_pmc.txt) against the fixed cost of the compare tile kernel
    int rc;
    const uint32_t nsamp = 24;
    DevBuf &cnt = scratch[1];
    if ((rc = cnt.ensure(16))) return rc;
    GS_HIP_CHECK(hipMemsetAsync(cnt.p, 0, 16, c->stream));
    const size_t lds_s = (sizeof(T) + 4) * ((size_t)1 << g.log2p);
    auto ks = k_match_sample<KIND, T>;
    if (lds_s > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
    hipLaunchKernelGGL(ks, dim3(g.chunks, nsamp), dim3(JT), lds_s, c->stream, qkey, nq, g.log2p, (const T *)cols, colcap, n, m, nsamp, cnt.as<unsigned long long>());
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
    *decline_out = decline ? 1 : 0;
    return GS_OK;
}

template <int KIND, typename T>
static int join_impl(gs_ctx *c, uint32_t m, const uint8_t *qrows, uint64_t qstride, uint32_t nq, const void *cols, uint64_t colcap, uint64_t n, uint16_t *out16,
                     uint64_t ld, DevBuf *scratch /* [JOIN_SCRATCH] reusable */, int *declined, unsigned long long *stats, bool init, uint64_t col0,
                     const void *rows, uint64_t rstride)
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
    JoinGeom g;
    g.log2p = 6;
    while (g.log2p < (uint32_t)JP_MAX_LOG2 && (double)(1u << g.log2p) * 0.4 < (double)nq) g.log2p++;
    g.chunks = (uint32_t)((n + (uint64_t)JT * JN - 1) / ((uint64_t)JT * JN));
    g.lds1 = (sizeof(T) + 4) * ((size_t)1 << g.log2p) + ((size_t)1 << JB_LOG2) / 8;
    g.ldsq = (size_t)(JT / 64) * 64 * (sizeof(T) == 4 ? 8 : 16);         // the wavefronts' survivor queues (one-slot rounds; 8-byte keys: two words per entry; their cluster variant does not use it)
    const bool may_decline = declined && m >= 64 && n >= 4096;
    const char *ce = getenv("GS_JOIN_CLUSTER");
    // heavy blocks: request batches (the insert path passes no `declined`) large enough for phase 0 to be a small part of the work
    bool cluster = declined && init && col0 == 0 && rows && nq >= 256 && n >= 8192 && m >= 16 * JS0 && (ld % 8) == 0 && !(ce && !atoi(ce)) && !getenv("GS_JOIN_DECLINE");
    if (ce && atoi(ce) == 2) cluster = declined && init && col0 == 0 && rows && m >= 2 * JS0 && (ld % 8) == 0;       // tests: small shapes too
    if (!cluster) {
        if (may_decline) {
            int dec = 0;
            if ((rc = join_sample_declines<KIND, T>(c, g, m, k0.as<T>(), nq, cols, colcap, n, scratch, &dec))) return rc;
            if (dec) { *declined = 1; return GS_OK; }
        }
        return join_launch<KIND, T, false>(c, g, k0.as<T>(), nq, cols, colcap, n, 0, m, out16, ld, stats, col0, false, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
    }
    const bool verbose = getenv("GS_JOIN_VERBOSE") != nullptr;
    // GS_JOIN_TIMES=1: wall time of every stage (a sync after each: diagnostic only)
    const bool times = getenv("GS_JOIN_TIMES") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!times) return;
        (void)hipStreamSynchronize(c->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[GS_JOIN_TIMES] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    lap("query columns + memset");
    // ---- phase 0 + heavy pairs + labels
    DevBuf &pairs = scratch[2], &ctr = scratch[3], &labq = scratch[4], &labe = scratch[5];
    uint64_t pair_cap = JPAIR_CAP;
    DevBuf &lcnt = scratch[14];
    if ((rc = pairs.ensure(pair_cap * 8)) || (rc = ctr.ensure(64)) || (rc = labq.ensure(4 * (size_t)nq)) || (rc = labe.ensure(4 * (size_t)n)) || (rc = lcnt.ensure(4 * (size_t)nq))) return rc;
    pair_cap = std::max<uint64_t>(pair_cap, pairs.bytes / 8);       // (a list grown by an earlier batch stays)
    if ((rc = join_launch<KIND, T, false>(c, g, k0.as<T>(), nq, cols, colcap, n, 0, JS0, out16, ld, stats, 0, true, nullptr, nullptr, nullptr, 0, nullptr, nullptr))) return rc;
    lap("phase 0 (first slots)");
    // (pinned staging: a pageable destination makes the runtime bounce 1.2 MB through its own buffer, synchronously)
    uint8_t *pin = (uint8_t *)pinned_pool(c)->ensure(34, 64 + 4 * (2 * (size_t)nq + n) + 2 * ((size_t)nq + 8 + (size_t)g.chunks * JT * JN) + 64);
    GS_REQUIRE(pin, GS_ERR_HIP, "match-join: pinned staging buffer");
    unsigned long long *hc = (unsigned long long *)pin;
    uint32_t *hlq = (uint32_t *)(pin + 64), *hcn = hlq + nq, *hle = hcn + nq;
    // The pair list holds (query, node) pairs with >= JHEAVY matches in the first slots. A database with a species of 30 000 genomes (skewed family sizes:
    // NCBI / GTDB, /root/reference/README.md:134) puts queries-of-that-species x 30 000 pairs on it - 7.6 M per 2500-query batch at 10 % of 300 k nodes -
    // where 100-member families leave ~0.3 M. A list that overflows is sized to what the scan counted and the scan repeated (once: the count is exact);
    // round 5 gave such a batch to the compare tile kernel (3 s per 10 000 queries instead of 0.1).
    for (int attempt = 0;; attempt++) {
        GS_HIP_CHECK(hipMemsetAsync(ctr.p, 0, 64, c->stream));
        hipLaunchKernelGGL(k_heavy_scan, dim3(c->n_cu * 8), dim3(256), 0, c->stream, out16, ld, nq, n, m, pairs.as<uint2>(), ctr.as<unsigned long long>(), pair_cap);
        lap("heavy scan");
        hipLaunchKernelGGL(k_label_init, dim3((uint32_t)((std::max<uint64_t>(n, nq) + 255) / 256)), dim3(256), 0, c->stream, labq.as<uint32_t>(), nq, labe.as<uint32_t>(), n);
        for (int it = 0; it < 6; it++)
            hipLaunchKernelGGL(k_label_prop, dim3(c->n_cu * 4), dim3(256), 0, c->stream, pairs.as<uint2>(), ctr.as<unsigned long long>(), pair_cap, labq.as<uint32_t>(), labe.as<uint32_t>());
        GS_HIP_CHECK(hipMemsetAsync(lcnt.p, 0, 4 * (size_t)nq, c->stream));
        hipLaunchKernelGGL(k_label_count, dim3(c->n_cu * 2), dim3(256), 0, c->stream, labe.as<uint32_t>(), n, nq, lcnt.as<uint32_t>());
        GS_HIP_CHECK(hipGetLastError());
        GS_HIP_CHECK(hipMemcpyAsync(hc, ctr.p, 16, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipMemcpyAsync(hlq, labq.p, 4 * (size_t)nq, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipMemcpyAsync(hcn, lcnt.p, 4 * (size_t)nq, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        lap("labels + query labels down");
        if (hc[0] <= pair_cap || attempt || hc[0] > JPAIR_MAX) break;
        if (pairs.alloc((size_t)hc[0] * 8 + 4096) != GS_OK) { (void)hipGetLastError(); (void)pairs.alloc(JPAIR_CAP * 8); pair_cap = JPAIR_CAP; break; }      // no room: as before
        pair_cap = hc[0] + 512;
    }
    const uint64_t npairs = hc[0];
    // ---- clusters: labels with >= 2 queries and >= 1 node, largest blocks dropped while the tile budget is exceeded
    std::vector<uint32_t> cq(nq, 0), ce_(nq, 0), cid(nq, 0);
    uint32_t K = 0, ntiles = 0, nhq = 0, dedup_below = 1;
    uint64_t nhe = 0;
    const uint32_t minq = getenv("GS_JOIN_CLUSTER_MINQ") ? (uint32_t)std::max(2, atoi(getenv("GS_JOIN_CLUSTER_MINQ"))) : (ce && atoi(ce) == 2 ? 2u : 12u);
    // A component becomes a cluster when it has enough QUERIES (the run of equal keys its isolates leave in the table is what costs) or enough PAIRS: a family of
    // 2000 genomes with five isolates in the batch has 8000 related (query, node) pairs of thousands of matches each, of which a node's run-length accumulator
    // absorbs one query's - the rest are atomics. Uniform 100-member families never reach the pair bar (5 x 100 = 400); a skewed database (tools/skew_probe.py)
    // is full of such components. Where the bar sits: a component costs at least one 128 x 128 compare tile (~18 us), a related pair left to the atomics ~0.5 us
    // (thousands of matches): ~40 pairs per tile break even; 128 leaves the uniform regime alone (the `min_saved` guard below still decides whether the
    // cluster-aware pass runs at all). tools/skew_probe.py, 100 k rows: 32 -> 47.0 ms, 128 -> 50.2, 1024 -> 50.6 (first form), 4096 -> 57.9, none: 258.
    const uint64_t minpairs = getenv("GS_JOIN_CLUSTER_MINPAIRS") ? (uint64_t)atoll(getenv("GS_JOIN_CLUSTER_MINPAIRS")) : 128;
    auto is_cluster = [&](uint32_t l) { return cq[l] >= minq || (cq[l] >= 2 && (uint64_t)(cq[l] - 1) * hcn[l] >= minpairs); };
    // (query, node) pairs the would-be clusters take off the match-by-match path, from the per-label counts alone: below the bar of the cluster-aware pass (min_saved, further
    // down) the batch takes the plain join without the node labels ever leaving the device (1.2 MB + a host pass over 300 k nodes per batch: 3 ms per request when every
    // batch of unrelated isolates paid it for its handful of three-query components)
    const uint64_t min_scaled0 = (uint64_t)((double)nq * 32.0 * (double)n / 3.0e5);
    const uint64_t min_saved0 = getenv("GS_JOIN_CLUSTER_MIN") ? (uint64_t)atoll(getenv("GS_JOIN_CLUSTER_MIN")) : (ce && atoi(ce) == 2 ? 0 : std::max<uint64_t>((uint64_t)nq * 8, min_scaled0));
    bool any = false;
    if (npairs <= pair_cap) {
        for (uint32_t q = 0; q < nq; q++) if (hlq[q] < nq) ++cq[hlq[q]];
        uint64_t est = 0;
        for (uint32_t l = 0; l < nq; l++) if (cq[l] && hcn[l] && is_cluster(l)) { any = true; est += (uint64_t)(cq[l] - 1) * hcn[l]; }
        if (est < min_saved0) any = false;
    }
    if (any) {
        // only now the node labels (1.2 MB for 300 k nodes, and a pass over them): a batch of unrelated isolates never gets here
        GS_HIP_CHECK(hipMemcpyAsync(hle, labe.p, 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        for (uint64_t e = 0; e < n; e++) if (hle[e] < nq) ce_[hle[e]]++;
        // Which components become clusters. What a cluster saves grows with the number of its queries: in the plain join a node's hit walks the run
        // of equal keys its cluster's queries left in the table (83 isolates: 83 entries) while the rest of the wavefront waits, and all but one
        // of them cost an atomic per match. What it costs is the cluster-aware kernel's own-cluster test on every hit of its nodes. Measured on
        // 300 k genomes, 2500-query batches (profiles/r04_join_cluster_sweep.txt): 83 queries per family 4.1x faster, 8 per family even, 3 per
        // family 1.6x slower - components below GS_JOIN_CLUSTER_MINQ (12) queries stay with the run-length accumulator.
        std::vector<uint32_t> order;
        for (uint32_t l = 0; l < nq; l++) if (ce_[l] >= 1 && is_cluster(l)) order.push_back(l);
        auto tiles_of = [&](uint32_t l) { return ((cq[l] + HTILE - 1) / HTILE) * ((ce_[l] + HTILE - 1) / HTILE); };
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { const uint32_t ta = tiles_of(a), tb = tiles_of(b); return ta != tb ? ta < tb : a < b; });
        std::vector<uint32_t> chosen;
        for (uint32_t l : order) {
            if (chosen.size() >= JCL_MAX || ntiles + tiles_of(l) > JTILE_CAP) break;                          // cheapest blocks first
            chosen.push_back(l); ntiles += tiles_of(l); nhq += cq[l]; nhe += ce_[l];
        }
        // cluster ids: the large clusters (>= JDEDUP_MINQ queries) first - only they share table entries
        const uint32_t dq = getenv("GS_JOIN_DEDUP_MINQ") ? (uint32_t)std::max(2, atoi(getenv("GS_JOIN_DEDUP_MINQ"))) : JDEDUP_MINQ;
        std::stable_sort(chosen.begin(), chosen.end(), [&](uint32_t a, uint32_t b) { return (cq[a] >= dq) > (cq[b] >= dq); });
        for (uint32_t l : chosen) { cid[l] = ++K; if (cq[l] >= dq) dedup_below = K + 1; }
    }
    // (query, node) pairs the blocks take off the match-by-match path; a handful is not worth a second kernel variant and the tile launch
    uint64_t saved_pairs = 0;
    for (uint32_t l = 0; l < nq; l++) if (cid[l]) saved_pairs += (uint64_t)(cq[l] - 1) * ce_[l];
    // (the cluster-aware kernel costs ~7 ms more than the plain one per 2500 queries x 300 k nodes - in proportion to queries x nodes - whatever it saves:
    // 32 pairs per query at 300 k nodes is where the two meet since the survivor queue made the plain kernel faster. 2500-query batches of isolates of 300
    // species - a handful of 12-query components by chance - ran 137 ms clustered against 125 plain; 200 species and fewer cluster as before:
    // profiles/r04_join_cluster_sweep.txt, bottom)
    const uint64_t min_scaled = (uint64_t)((double)nq * 32.0 * (double)n / 3.0e5);
    const uint64_t min_saved = getenv("GS_JOIN_CLUSTER_MIN") ? (uint64_t)atoll(getenv("GS_JOIN_CLUSTER_MIN")) : (ce && atoi(ce) == 2 ? 0 : std::max<uint64_t>((uint64_t)nq * 8, min_scaled));
    if (verbose)
        fprintf(stderr, "[GS_JOIN] nq=%u n=%llu heavy pairs %llu (matches in the first %u slots %llu): %u clusters, %u queries x %llu nodes in %u tiles, %llu pairs off the atomics%s\n", nq,
                (unsigned long long)n, (unsigned long long)npairs, JS0, hc[1], K, nhq, (unsigned long long)nhe, ntiles, (unsigned long long)saved_pairs,
                K && saved_pairs < min_saved ? " - not worth the cluster-aware pass" : "");
    if (K && saved_pairs < min_saved) K = 0;
    lap("host: components");
    if (K == 0) {
        // nothing to cluster (or too much: pair list overflow): the plain join over the remaining slots - unless the matches of phase 0, scaled to
        // all slots as if every one cost an atomic, say the tile kernel may be cheaper: then the sampled estimate (which knows about runs) decides
        const double t_tile = (double)((nq + 127) / 128 * 128) * (double)n * (double)m / (KIND == GS_KIND_U64 ? 1.4e13 : 1.6e13);
        if (may_decline && (npairs > pair_cap || (double)hc[1] * ((double)m / JS0) / 1.6e10 > 0.5 * t_tile)) {
            int dec = 0;
            if ((rc = join_sample_declines<KIND, T>(c, g, m, k0.as<T>(), nq, cols, colcap, n, scratch, &dec))) return rc;
            if (dec) { *declined = 1; return GS_OK; }
        }
        return join_launch<KIND, T, false>(c, g, k0.as<T>(), nq, cols, colcap, n, JS0, m, out16, ld, stats, 0, false, nullptr, nullptr, nullptr, 0, nullptr, nullptr);
    }
    // member lists in cluster order, tiles, per-query / per-node cluster ids
    std::vector<uint32_t> cl_lo(K + 2, 0), ce_lo(K + 2, 0);
    for (uint32_t l = 0; l < nq; l++) if (cid[l]) { cl_lo[cid[l] + 1] = cq[l]; ce_lo[cid[l] + 1] = ce_[l]; }
    for (uint32_t k = 1; k <= K + 1; k++) { cl_lo[k] += cl_lo[k - 1]; ce_lo[k] += ce_lo[k - 1]; }
    std::vector<uint32_t> qlist(nhq), elist(nhe), fq(cl_lo.begin(), cl_lo.end()), fe(ce_lo.begin(), ce_lo.end());
    // node labels in the layout of the join's lanes: node e = chunk * 8192 + i * 1024 + lane sits at [chunk][i / 4][lane][i % 4] (padded to whole chunks)
    const uint64_t npad = (uint64_t)g.chunks * JT * JN;
    uint16_t *hqcl = (uint16_t *)(hle + n), *hnl = hqcl + ((nq + 3) & ~3u);          // (pinned, behind the labels; 8-byte aligned)
    for (uint32_t q = 0; q < nq; q++) { const uint32_t l = hlq[q]; hqcl[q] = 0; if (l < nq && cid[l]) { hqcl[q] = (uint16_t)cid[l]; qlist[fq[cid[l]]++] = q; } }
    memset(hnl, 0, 2 * npad);
    for (uint64_t e = 0; e < n; e++) {
        const uint32_t l = hle[e];
        if (l < nq && cid[l]) {
            const uint64_t ch = e / (JT * JN), w = e % (JT * JN), i = w / JT, ln = w % JT;
            hnl[((ch * (JN / JU) + i / JU) * JT + ln) * JU + i % JU] = (uint16_t)cid[l];
            elist[fe[cid[l]]++] = (uint32_t)e;
        }
    }
    std::vector<uint4> tiles;
    tiles.reserve(ntiles);
    // the thin tiles first (at most 16 query rows - most clusters of a skewed database: hamming_blocks has a kernel of its own for them), then the others
    uint32_t n_thin = 0;
    for (int pass = 0; pass < 2; pass++)
        for (uint32_t k = 1; k <= K; k++)
            for (uint32_t a = cl_lo[k]; a < cl_lo[k + 1]; a += HTILE) {
                const uint32_t qr = std::min<uint32_t>(HTILE, cl_lo[k + 1] - a);
                if ((qr <= 16) != (pass == 0)) continue;
                for (uint32_t b = ce_lo[k]; b < ce_lo[k + 1]; b += HTILE) {
                    tiles.push_back(make_uint4(a, qr, b, std::min<uint32_t>(HTILE, ce_lo[k + 1] - b)));
                    n_thin += pass == 0;
                }
            }
    lap("host: lists");
    DevBuf &dqcl = scratch[6], &dnl = scratch[7], &dql = scratch[8], &del = scratch[9], &dtl = scratch[10], &dqs = scratch[11], &dcl = scratch[13];
    if ((rc = dqcl.ensure(2 * (size_t)nq)) || (rc = dnl.ensure(2 * (size_t)npad)) || (rc = dql.ensure(4 * (size_t)nhq)) || (rc = del.ensure(4 * (size_t)nhe)) ||
        (rc = dtl.ensure(16 * tiles.size())) || (rc = dqs.ensure(sizeof(T) * (size_t)m * nhq)) || (rc = dcl.ensure(4 * (size_t)(K + 2)))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(dqcl.p, hqcl, 2 * (size_t)nq, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dnl.p, hnl, 2 * (size_t)npad, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dql.p, qlist.data(), 4 * (size_t)nhq, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(del.p, elist.data(), 4 * (size_t)nhe, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dtl.p, tiles.data(), 16 * tiles.size(), hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dcl.p, cl_lo.data(), 4 * (size_t)(K + 2), hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));                  // (the host vectors above are pageable: the copies must be done before they go)
    hipLaunchKernelGGL((k_query_cols_list<T>), dim3((nhq + 31) / 32, (m + 31) / 32), dim3(256), 0, c->stream, qrows, qstride, dql.as<uint32_t>(), nhq, m, dqs.as<T>());
    GS_HIP_CHECK(hipGetLastError());
    lap("uploads + sorted key copy");
    // the in-place own-cluster test of k_match_join pays where a good part of the NODES belongs to clusters (a skewed database: 72 %); with a percent of them (isolates of a few
    // species against a uniform database) nearly every wavefront still holds one such node and pays the test's fixed part for nothing: +6 ms per request_redundant step
    const bool inplace = getenv("GS_JOIN_INPLACE") ? atoi(getenv("GS_JOIN_INPLACE")) != 0 : nhe * 10 >= n;
    if ((rc = join_launch<KIND, T, true>(c, g, k0.as<T>(), nq, cols, colcap, n, JS0, m, out16, ld, stats, 0, false, dqcl.as<uint16_t>(), dnl.as<uint16_t>(), dqs.as<T>(), nhq,
                                         dcl.as<uint32_t>(), dql.as<uint32_t>(), dedup_below | (inplace ? 0x80000000u : 0u)))) return rc;
    {
        ProfScope ps(c, FAM_HAMMING);
        lap("cluster-aware join");
        if ((rc = hamming_blocks(c, KIND, m, qrows, qstride, rows, rstride, dtl.p, (uint32_t)tiles.size(), dql.as<uint32_t>(), del.as<uint32_t>(), out16, ld, n_thin))) return rc;
        lap("block compare");
    }
    return GS_OK;
}

uint64_t match_join_max_queries() { const char *e = getenv("GS_JOIN_MAXQ"); return e ? (uint64_t)std::max(1, std::min(4094, atoi(e))) : JQ_MAX; }

// `declined` (optional): set to 1 - and out16 is left at its initial value m - when the sampled match density says the compare tile kernel is the
// cheaper producer for this batch (the caller then runs it)
int match_join_counts(gs_ctx *c, int kind, uint32_t m, const void *qrows, uint64_t qstride, uint64_t nq, const void *cols, uint64_t colcap, uint64_t n,
                      uint16_t *out16, uint64_t ld, DevBuf *scratch, int *declined, unsigned long long *stats, bool init, uint64_t col0, const void *rows, uint64_t rstride)
{
    GS_REQUIRE(nq >= 1 && nq <= 4094 && m <= 65535 && (ld % 2) == 0 && ((uintptr_t)out16 % 4) == 0, GS_ERR_INVALID, "match_join_counts: bad shape");
    GS_REQUIRE((uint64_t)m * nq < ((uint64_t)1 << 31), GS_ERR_INVALID, "match_join_counts: batch too large");
    if (kind == GS_KIND_U64) return join_impl<GS_KIND_U64, uint64_t>(c, m, (const uint8_t *)qrows, qstride, (uint32_t)nq, cols, colcap, n, out16, ld, scratch, declined, stats, init, col0, rows, rstride);
    if (kind == GS_KIND_F32) return join_impl<GS_KIND_F32, uint32_t>(c, m, (const uint8_t *)qrows, qstride, (uint32_t)nq, cols, colcap, n, out16, ld, scratch, declined, stats, init, col0, rows, rstride);
    return join_impl<GS_KIND_U32, uint32_t>(c, m, (const uint8_t *)qrows, qstride, (uint32_t)nq, cols, colcap, n, out16, ld, scratch, declined, stats, init, col0, rows, rstride);
}

int rows_to_cols(gs_ctx *c, int kind, uint32_t m, const void *rows, uint64_t stride, uint64_t nrows, void *cols, uint64_t colcap, uint64_t first)
{
    if (nrows == 0) return GS_OK;
    dim3 g((uint32_t)((nrows + 31) / 32), (m + 31) / 32);
    if (kind == GS_KIND_U64) hipLaunchKernelGGL((k_rows_to_cols<uint64_t>), g, dim3(256), 0, c->stream, (const uint8_t *)rows, stride, nrows, m, (uint64_t *)cols, colcap, first, 0);
    else hipLaunchKernelGGL((k_rows_to_cols<uint32_t>), g, dim3(256), 0, c->stream, (const uint8_t *)rows, stride, nrows, m, (uint32_t *)cols, colcap, first, (int)(kind == GS_KIND_F32));
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
