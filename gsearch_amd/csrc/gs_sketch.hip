// gs_sketch.hip — k-mer extraction + MinHash-family sketching on gfx950.
//
// Replaces, at batch level, SeqSketcherT::sketch_compressedkmer{,_seqs} (call sites
// /root/reference/src/dna/dnasketch.rs:336,357, src/dna/dnarequest.rs:272,287, src/aa/aasketch.rs:313,329,
// src/aa/aarequest.rs:268,283) together with the kmer_hash_fn closures (dnasketch.rs:164-169,
// aasketch.rs:156-160) and the probminhash sketchers they feed. Arithmetic: SPEC.md 1-3.
//
// Layout: a genome is a flat range of 32-symbol "units" (DNA: one 8-byte packed word, AA: 32 bytes).
// One workgroup (or `parts` workgroups) per genome; every lane owns one unit per iteration, rebuilds the
// rolling forward / reverse-complement state from the k-1 symbols before it, and pushes every valid
// k-mer through the element hash into an m-slot min table held in LDS (ds_min_u32). HBM traffic is the
// algorithmic minimum: each packed word is read once (plus an L1/L2-served halo word) and m slots written.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "gs_internal.hpp"
#include "gs_spec.hpp"

namespace gs {

#define GS_EMPTY32 0xFFFFFFFFu
static constexpr int SK_THREADS = 512;

__constant__ uint8_t c_aa_code[32] = {
    // index = ASCII & 31 : @ A B C D E F G H I J K L M N O P Q R S T U V W X Y Z ...
    0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 0, 8, 9, 10, 11, 0, 12, 13, 14, 15, 16, 0, 17, 18, 0, 19, 0, 0, 0, 0, 0, 0};

// the same table as c_aa_code, packed 8 codes x 5 bits per word and kept in registers: a per-lane (divergent) index into
// __constant__ memory is a vector memory load per residue, this is four VALU operations
__device__ __forceinline__ uint32_t aa_code_reg(uint32_t ch)
{
    const uint32_t idx = ch & 31, sh = (idx & 7) * 5;
    const uint64_t lo = (idx & 8) ? 0x2d49400e6ull : 0x2906208000ull, hi = (idx & 8) ? 0x260ull : 0x944107b9acull;
    return (uint32_t)(((idx & 16) ? hi : lo) >> sh) & 31u;
}

// per-record unit counts -> exclusive prefix inside each genome (one thread per genome; record lists are short)
__global__ void k_unit_prefix(const uint64_t *rec_start, const uint64_t *rec_len, const uint64_t *genome_rec_off,
                              uint64_t n_genomes, uint32_t k, uint64_t *rec_upre, uint64_t *gen_units)
{
    // one wavefront per genome, 64 records per trip (a proteome has thousands of records: a serial loop of dependent loads per
    // genome cost more than sketching it)
    const uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (g >= n_genomes) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r0 = genome_rec_off[g], r1 = genome_rec_off[g + 1];
    uint64_t base = 0;
    for (uint64_t rb = r0; rb < r1; rb += 64) {
        const uint64_t r = rb + lane;
        uint64_t u = 0;
        if (r < r1) { const uint64_t len = rec_len[r]; if (len >= k) { const uint64_t s = rec_start[r]; u = ((s + len - 1) >> 5) - (s >> 5) + 1; } }
        uint64_t inc = u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint64_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
        if (r < r1) rec_upre[r] = base + inc - u;
        base += __shfl(inc, 63);
    }
    if (lane == 0) gen_units[g] = base;
}

// level-0 candidate of every slot-min sketcher (SPEC 3.1 / 3.2): element hash -> (key, slot) -> min.
//   optdens, revoptdens, super : key = 23-bit mantissa of U32f          (T = u32)
//   super2 / u32               : key = next32                            (T = u32)
//   super2 / u64               : key = next64                            (T = u64)
template <int ALGO, int VBITS, typename T, bool LDS_TABLE>
struct MinEmit {
    T *table; uint32_t m; uint64_t zone;
    uint16_t *bound;              // !LDS_TABLE: per-slot upper bound (top 16 bits) of this workgroup's minima, in LDS
    __device__ __forceinline__ void operator()(uint64_t v, uint64_t /*rec*/, uint64_t /*pos*/) const
    {
        uint64_t o1; uint32_t b;
        two_draw(elem_hash<ALGO, VBITS>(v), m, zone, o1, b);
        T key;
        if (ALGO == ALGO_SUPER2) key = sizeof(T) == 8 ? (T)o1 : (T)(o1 >> 32);
        else key = (T)(o1 >> 41);
        if (LDS_TABLE) atomicMin(&table[b], key);
        else {
            // The slot table does not fit in LDS (e.g. 24000 x u64 = 192 kB): it lives in global memory, but a 2-byte LDS filter
            // keeps all but the ~ln(n)/n fraction of k-mers that can still lower a minimum away from it. bound[b] only ever holds
            // the top bits of a key this workgroup has applied, so it is >= the top bits of the true minimum: skipping keys whose
            // top bits exceed it is exact; plain (racy) 16-bit stores can only leave it staler, i.e. more conservative.
            const uint16_t hi = (uint16_t)(key >> (8 * sizeof(T) - 16));
            if (hi <= bound[b]) { atomicMin(&table[b], key); bound[b] = hi; }
        }
    }
};

// Hooks of walk_genome that only the filtered emitter overrides: a k-mer emitted by a FULL wavefront (all 64 lanes inside a record),
// the end of such a word, and the end of the walk.
template <class E> __device__ __forceinline__ void emit_full_wave(const E &e, uint64_t v, uint64_t rec, uint64_t pos) { e(v, rec, pos); }
template <class E> __device__ __forceinline__ void emit_word_done(const E &) {}
template <class E> __device__ __forceinline__ void emit_finish(const E &) {}

// Filtered form of MinEmit for an LDS-resident slot table. The key of a k-mer only needs two of the three SplitMix64
// outputs (o1 = f(s0, s3)); its slot needs the third and the second xoshiro output. A key that is not below `thr` - an upper bound of
// EVERY slot's current minimum, refreshed from the table now and then - cannot lower any slot, whichever slot it falls in, so the rest
// of its arithmetic is skipped: exact. Lanes diverge on that test, so the survivors (a few % once the table has warmed up) are
// compacted through a per-wave LDS queue of element hashes and finished 64 at a time by the whole wave. Only full waves use the queue
// (its fill count stays wave-uniform in a register); record boundaries and partial waves take the direct form.
constexpr int SKQ = 128;          // queue entries per wave: < 64 pending + <= 64 new
template <int ALGO, int VBITS, typename T>
struct MinEmitF {
    T *table; uint32_t m; uint64_t zone;
    uint64_t *q;                                  // this wave's queue
    uint32_t lane;
    mutable uint32_t qn, words;                   // wave-uniform: pending survivors, full-wave words walked
    mutable T thr;                                // wave-uniform: every slot's minimum is <= thr
    // Round 5, the SPECULATIVE bound: `thr` starts at - and never exceeds - `cap`, a guess of where the largest slot minimum of THIS genome will end
    // ((m / N)(ln m + c): the coupon-collector tail of N k-mers over m slots). Every k-mer whose key is not below the cap is dropped from the first word
    // on - no warm-up during which everything passes, ~6 % survivors instead of the ~15 % a running maximum lets through over a 5 Mbp genome - and the
    // kernel CHECKS the guess afterwards: a slot whose minimum is below the cap has seen every k-mer that could matter to it (only keys >= cap were
    // dropped), a slot that is not sends the workgroup through the genome again under the running bound alone (same table: minima are idempotent).
    // The cap is a performance parameter only - the signature never depends on it. EMPTY = no cap.
    T cap;
    static __device__ __forceinline__ T key_of(uint64_t o1)
    {
        if (ALGO == ALGO_SUPER2) return sizeof(T) == 8 ? (T)o1 : (T)(o1 >> 32);
        return (T)(o1 >> 41);
    }
    static __device__ __forceinline__ T direct_above()        // survivor rate >= 1/4: the queue costs more than it saves (break-even ~0.34)
    {
        if (ALGO == ALGO_SUPER2) return (T)((T)1 << (8 * sizeof(T) - 2));
        return (T)((T)1 << 21);
    }
    __device__ __forceinline__ void apply(uint64_t h) const
    {
        uint64_t o1; uint32_t b;
        two_draw(h, m, zone, o1, b);
        atomicMin(&table[b], key_of(o1));
    }
    __device__ __forceinline__ void operator()(uint64_t v, uint64_t, uint64_t) const { apply(elem_hash<ALGO, VBITS>(v)); }
    __device__ __forceinline__ void full(uint64_t v) const
    {
        const uint64_t h = elem_hash<ALGO, VBITS>(v);
        if (thr >= direct_above()) { apply(h); return; }            // wave-uniform
        const uint64_t s0 = splitmix_mix(h + GS_GAMMA), s3 = splitmix_mix(h + 4 * GS_GAMMA);
        const T key = key_of(rotl64(s0 + s3, 23) + s0);
        const bool pass = key < thr;
        const uint64_t bal = __ballot(pass);
        if (bal) {
            if (pass) q[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = h;
            qn += (uint32_t)__popcll(bal);
            if (qn >= 64) {
                apply(q[lane]);
                const uint32_t rest = qn - 64;
                const uint64_t mv = q[64 + lane];
                if (lane < rest) q[lane] = mv;
                qn = rest;
            }
        }
    }
    __device__ __forceinline__ void word_done() const
    {
        words++;
        if (cap != (T)~(T)0) { if ((words & 63u) != 0) return; }                                             // under a cap the running maximum only matters late in the genome
        else if (words <= 8 ? (words & (words - 1)) != 0 : (words & (words <= 128 ? 7 : 15)) != 0) return;     // 1, 2, 4, 8, then every 8th word, every 16th after 128
        T mx = 0;
        for (uint32_t i = lane; i < m; i += 64) { const T x = table[i]; mx = x > mx ? x : mx; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const T y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
        thr = mx < cap ? mx : cap;
    }
    __device__ __forceinline__ void finish() const
    {
        if (lane < qn) apply(q[lane]);
        qn = 0;
    }
};
template <int ALGO, int VBITS, typename T> __device__ __forceinline__ void emit_full_wave(const MinEmitF<ALGO, VBITS, T> &e, uint64_t v, uint64_t, uint64_t) { e.full(v); }
template <int ALGO, int VBITS, typename T> __device__ __forceinline__ void emit_word_done(const MinEmitF<ALGO, VBITS, T> &e) { e.word_done(); }
template <int ALGO, int VBITS, typename T> __device__ __forceinline__ void emit_finish(const MinEmitF<ALGO, VBITS, T> &e) { e.finish(); }

// The streaming part shared by every sketcher. walk_unit: flat unit f of a genome (32 symbols: one packed word of DNA, 32 bytes of AA) ->
// emit(v) for each valid canonical k-mer value that starts... ends in it; walk_genome: the units of genome g assigned to this workgroup.
// RCM: how the strand rule reaches the loop - 2 = at run time through rc_or (every sketcher but the hot one), 0 / 1 = compiled in (canonical / forward only):
// k_sketch_min is VALU-issue bound and the run-time form turns its two v_or_b32 per k-mer into three-operand v_or3_b32, which issue 1.6x slower
// (profiles/r02_ubench_valu.txt: 1.75 against 1.08 ns) - 115.7 instead of 112.5 ms per 10 000 genomes (profiles/r05_bench_request_rc_runtime.log)
template <bool AA, class Emit, int RCM = 2>
__device__ __forceinline__ void walk_unit(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len,
                                          const uint64_t *__restrict__ rec_upre, uint64_t r0, uint64_t r1, uint64_t f, uint32_t k, uint64_t mask, uint32_t rcshift,
                                          uint64_t rc_or_rt, const Emit &emit)
{
    const uint64_t rc_or = RCM == 0 ? (uint64_t)0 : RCM == 1 ? ~(uint64_t)0 : rc_or_rt;
    {
        // record owning flat unit f: last r in [r0,r1) with rec_upre[r] <= f
        uint64_t lo = r0, hi = r1;
        while (hi - lo > 1) { uint64_t mid = (lo + hi) >> 1; if (rec_upre[mid] <= f) lo = mid; else hi = mid; }
        const uint64_t rb = rec_start[lo], re = rb + rec_len[lo];
        const uint64_t u = (rb >> 5) + (f - rec_upre[lo]);
        const uint64_t a0 = u << 5;
        const uint64_t first_valid = rb + k - 1;
        if (!AA) {
            const uint64_t *w64 = (const uint64_t *)seq;
            uint64_t w = __builtin_bswap64(w64[u]);
            uint64_t fwd = 0, rc = 0;
            if (a0 > rb && k > 1) {
                // the state after the k-1 bases in front of this word, in closed form (round 5: the loop over them - k-1 = 20 trips of ~9 instructions per 32 k-mers -
                // was 5.6 of the ~82 VALU instructions per k-mer): the forward window is the low 2(k-1) bits of the previous word; the reverse-complement
                // register holds base i of those k-1 at bit 2i, complemented - the 2-bit groups in reverse order, one group up
                const uint64_t pw = __builtin_bswap64(w64[u - 1]);
                const uint64_t lowm = ((uint64_t)1 << (2 * (k - 1))) - 1;            // k - 1 <= 31
                fwd = pw & lowm;
                uint64_t br = __builtin_bitreverse64(fwd);
                br = ((br >> 1) & 0x5555555555555555ull) | ((br & 0x5555555555555555ull) << 1);      // bit order inside each group back
                rc = (((~(br >> (2 * (33 - k)))) & lowm) << 2) | rc_or;
            }
            // (rc never exceeds 2k bits and fwd is masked every step: the minimum needs no further mask.) When every lane of the wave
            // holds an interior word - all 32 windows inside its record, the case for all but the first and last word of a record - the
            // per-window bounds test (two 64-bit compares, an exec save / restore and a branch per k-mer) is dropped for the whole word.
            const bool interior = a0 >= first_valid && a0 + 32 <= re;
            const uint64_t bint = __ballot(interior);
            if (bint == ~(uint64_t)0) {                      // all 64 lanes: emitters that compact across the wave may do so
#pragma unroll 2
                for (uint32_t j = 0; j < 32; j++) {
                    uint64_t c = w >> 62; w <<= 2;
                    fwd = ((fwd << 2) | c) & mask;
                    rc = (rc >> 2) | ((3 - c) << rcshift) | rc_or;
                    emit_full_wave(emit, fwd < rc ? fwd : rc, lo, a0 + j);
                }
                emit_word_done(emit);
            } else if (bint == __ballot(true)) {
#pragma unroll 2
                for (uint32_t j = 0; j < 32; j++) {
                    uint64_t c = w >> 62; w <<= 2;
                    fwd = ((fwd << 2) | c) & mask;
                    rc = (rc >> 2) | ((3 - c) << rcshift) | rc_or;
                    emit(fwd < rc ? fwd : rc, lo, a0 + j);
                }
            } else {
#pragma unroll 2
                for (uint32_t j = 0; j < 32; j++) {
                    uint64_t c = w >> 62; w <<= 2;
                    fwd = ((fwd << 2) | c) & mask;
                    rc = (rc >> 2) | ((3 - c) << rcshift) | rc_or;
                    uint64_t a = a0 + j;
                    if (a >= first_valid && a < re) emit(fwd < rc ? fwd : rc, lo, a);
                }
            }
        } else {
            const uint64_t *w64 = (const uint64_t *)seq;
            uint64_t val = 0;
            if (a0 > rb && k > 1) {
                // previous k-1 residues: bytes a0-(k-1) .. a0-1 (k-1 <= 11 -> inside the previous two 8-byte words)
                for (uint32_t j = 0; j + 1 < k; j++) {
                    uint64_t a = a0 - (k - 1) + j;
                    uint8_t ch = seq[a];
                    val = ((val << 5) | aa_code_reg(ch)) & mask;
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                uint64_t x = w64[u * 4 + q];
#pragma unroll 2
                for (uint32_t j = 0; j < 8; j++) {
                    uint32_t ch = (uint32_t)(x & 0xFF); x >>= 8;
                    val = ((val << 5) | aa_code_reg(ch)) & mask;
                    uint64_t a = a0 + q * 8 + j;
                    if (a >= first_valid && a < re) emit(val, lo, a);
                }
            }
        }
    }
}
// Kernels that walk sequences take `kq` = k | KQ_FWD: bit 8 set means GS_DATA_DNA_FWD - the k-mer is the forward window itself, no
// reverse-complement minimum (the k <= 14 closure of /root/reference/src/bin/bindash.rs:346-354: `kmer.get_compressed_value() & mask`).
// The walkers keep ONE code path: rc_or = ~0 pins the reverse-complement register at all ones, so `min(fwd, rc)` is fwd (the OR folds into the
// v_or3 that already merges the shifted halves - no instruction more on the canonical path).
enum { KQ_FWD = 0x100 };
__device__ __forceinline__ uint32_t kq_k(uint32_t kq) { return kq & 0xFFu; }
__device__ __forceinline__ uint64_t kq_rc_or(uint32_t kq) { return (kq & KQ_FWD) ? ~(uint64_t)0 : (uint64_t)0; }
static inline uint32_t kq_of(const gs_sketch_params *p) { return p->k | (p->data_t == GS_DATA_DNA_FWD ? (uint32_t)KQ_FWD : 0u); }
__device__ __forceinline__ uint64_t kmer_mask(bool aa, uint32_t k) { return aa ? (((uint64_t)1 << (5 * k)) - 1) : (k == 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1)); }
template <bool AA, class Emit, int RCM = 2>
__device__ __forceinline__ void walk_genome(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start,
                                            const uint64_t *__restrict__ rec_len, const uint64_t *__restrict__ rec_upre,
                                            uint64_t r0, uint64_t r1, uint64_t units, uint32_t kq, uint32_t part,
                                            uint32_t parts, const Emit &emit)
{
    const uint32_t k = kq_k(kq);
    const uint64_t mask = kmer_mask(AA, k), rc_or = kq_rc_or(kq);
    const uint32_t rcshift = 2 * (k - 1);
    for (uint64_t f = (uint64_t)part * blockDim.x + threadIdx.x; f < units; f += (uint64_t)parts * blockDim.x)
        walk_unit<AA, Emit, RCM>(seq, rec_start, rec_len, rec_upre, r0, r1, f, k, mask, rcshift, rc_or, emit);
    emit_finish(emit);
}

// ---- main kernel of optdens / revoptdens / super / super2: per-slot minimum over all k-mers (SPEC 3.1, 3.2 level 0)
template <bool AA, bool LDS_TABLE, int ALGO, int VBITS, typename T, bool FILT, int RCM = 0>
__global__ __launch_bounds__(SK_THREADS) void k_sketch_min(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start,
                                                            const uint64_t *__restrict__ rec_len, const uint64_t *__restrict__ rec_upre,
                                                            const uint64_t *__restrict__ genome_rec_off, const uint64_t *__restrict__ gen_units,
                                                            uint32_t k, uint32_t m, uint64_t zone, T *__restrict__ table_out, float cap_c)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_raw_table[];
    T *s_table = (T *)s_raw_table;
    const T EMPTY = (T)~(T)0;
    const uint64_t g = blockIdx.y;
    const uint32_t part = blockIdx.x, parts = gridDim.x;
    T *gtab = table_out + g * (uint64_t)m;
    T *table = LDS_TABLE ? s_table : gtab;
    uint16_t *s_bound = (uint16_t *)s_raw_table;
    if (LDS_TABLE) {
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) s_table[i] = EMPTY;
        __syncthreads();
    } else {
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) s_bound[i] = 0xFFFFu;
        __syncthreads();
    }
    if (FILT) {
        uint64_t *qbase = (uint64_t *)(s_raw_table + (((size_t)m * sizeof(T) + 15) & ~(size_t)15));
        // speculative bound (MinEmitF::cap): whole genomes only (a part's table is not the genome's), and only where it is a real cut (< 1/4 of the key range)
        T cap = EMPTY;
        if (LDS_TABLE && parts == 1 && cap_c != 0.0f) {
            const float nk = (float)gen_units[g] * 32.0f;                        // k-mers of the genome, from above
            const float frac = (float)m / nk * (__logf((float)m) + cap_c);
            if (frac > 0.0f && frac < 0.25f) {
                const float scale = ALGO == ALGO_SUPER2 ? (sizeof(T) == 8 ? 0x1.0p64f : 0x1.0p32f) : 0x1.0p23f;
                cap = (T)(frac * scale);
            }
        }
        for (int pass = 0; pass < 2; pass++) {
            MinEmitF<ALGO, VBITS, T> emit{table, m, zone, qbase + (threadIdx.x >> 6) * SKQ, threadIdx.x & 63, 0u, 0u, cap, cap};
            walk_genome<AA, MinEmitF<ALGO, VBITS, T>, RCM>(seq, rec_start, rec_len, rec_upre, genome_rec_off[g], genome_rec_off[g + 1], gen_units[g], k, part, parts, emit);
            if (cap == EMPTY) break;
            __syncthreads();
            int bad = 0;
            for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) bad |= (int)(s_table[i] >= cap);
            if (!__syncthreads_or(bad)) break;                                   // every slot's minimum lies below the cap: nothing that was dropped could have mattered
            cap = EMPTY;                                                         // (~1 genome in 1000 at c = 7: once more, under the running bound alone)
        }
    } else {
        MinEmit<ALGO, VBITS, T, LDS_TABLE> emit{table, m, zone, s_bound};
        walk_genome<AA, MinEmit<ALGO, VBITS, T, LDS_TABLE>, RCM>(seq, rec_start, rec_len, rec_upre, genome_rec_off[g], genome_rec_off[g + 1], gen_units[g], k, part, parts, emit);
    }
    if (LDS_TABLE) {
        __syncthreads();
        if (parts == 1) { for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) gtab[i] = s_table[i]; }
        else { for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) { T v = s_table[i]; if (v != EMPTY) atomicMin(&gtab[i], v); } }
    }
}

// ---- finish: u32 min table -> f32 signature, densification of empty bins (cold path) ----------------
template <int ALGO>
__global__ __launch_bounds__(256) void k_oph_finish(const uint32_t *__restrict__ table, uint32_t m, uint64_t zone,
                                                     uint32_t *__restrict__ win_scratch, float *__restrict__ sig)
{
    __shared__ uint32_t s_filled, s_nempty;
    const uint64_t g = blockIdx.x;
    const uint32_t *slot = table + g * (uint64_t)m;
    float *out = sig + g * (uint64_t)m;
    if (threadIdx.x == 0) s_filled = 0;
    __syncthreads();
    uint32_t loc = 0;
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) loc += (slot[i] != GS_EMPTY32);
    if (loc) atomicAdd(&s_filled, loc);
    __syncthreads();
    const uint32_t nf = s_filled;
    if (nf == m) { for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) out[i] = (float)slot[i] * 0x1.0p-23f; return; }
    if (nf == 0) { for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) out[i] = 1.0f; return; }
    if (ALGO == ALGO_OPTDENS) {
        for (uint32_t b = threadIdx.x; b < m; b += blockDim.x) {
            uint32_t v = slot[b];
            if (v == GS_EMPTY32) {
                Rng rg; rg.seed((uint64_t)b);
                for (;;) { uint32_t j = (uint32_t)rng_uint(rg, (uint64_t)m, zone); v = slot[j]; if (v != GS_EMPTY32) break; }
            }
            out[b] = (float)v * 0x1.0p-23f;
        }
    } else {
        uint32_t *dens = (uint32_t *)out;            // u32 image until the final conversion
        uint32_t *win = win_scratch + g * (uint64_t)m;
        for (uint32_t b = threadIdx.x; b < m; b += blockDim.x) dens[b] = slot[b];
        if (threadIdx.x == 0) s_nempty = m - nf;
        __syncthreads();
        for (uint64_t t = 0; s_nempty > 0; t++) {
            for (uint32_t b = threadIdx.x; b < m; b += blockDim.x) win[b] = GS_EMPTY32;
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < m; j += blockDim.x) {
                if (slot[j] == GS_EMPTY32) continue;
                Rng rg; rg.seed(((uint64_t)j << 20) + t);
                uint32_t i = (uint32_t)rng_uint(rg, (uint64_t)m, zone);
                if (dens[i] == GS_EMPTY32) atomicMin(&win[i], j);
            }
            __syncthreads();
            uint32_t got = 0;
            for (uint32_t b = threadIdx.x; b < m; b += blockDim.x) if (win[b] != GS_EMPTY32) { dens[b] = slot[win[b]]; got++; }
            if (got) atomicSub(&s_nempty, got);
            __syncthreads();
        }
        for (uint32_t b = threadIdx.x; b < m; b += blockDim.x) out[b] = (float)dens[b] * 0x1.0p-23f;
    }
}

// ---- super / super2 finish: every slot filled at level 0 (always true at BASELINE sizes) -> done; otherwise the genome
// is flagged for the exact sequential walk below.
template <int ALGO, typename T>
__global__ __launch_bounds__(256) void k_smh_finish(const T *__restrict__ table, uint32_t m, void *__restrict__ sig, uint8_t *__restrict__ cold)
{
    __shared__ uint32_t s_filled;
    const uint64_t g = blockIdx.x;
    const T *slot = table + g * (uint64_t)m;
    const T EMPTY = (T)~(T)0;
    if (threadIdx.x == 0) s_filled = 0;
    __syncthreads();
    uint32_t loc = 0;
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) loc += (slot[i] != EMPTY);
    if (loc) atomicAdd(&s_filled, loc);
    __syncthreads();
    if (threadIdx.x == 0) cold[g] = s_filled != m;
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        if (ALGO == ALGO_SUPER) ((float *)sig)[g * (uint64_t)m + i] = slot[i] == EMPTY ? INFINITY : 0.0f + (float)slot[i] * 0x1.0p-23f;
        else ((T *)sig)[g * (uint64_t)m + i] = slot[i];
    }
}

// ---- cold path of super / super2: Ertl's SuperMinHash walk (SPEC 3.2) executed exactly, one lane per genome.
// Only genomes with fewer than ~m ln m distinct k-mers get here (plasmids, test inputs).
struct SmhScratch { uint32_t *lvl, *perm, *hist, *q; uint64_t *rr; uint8_t *filled; };
template <int ALGO, int VBITS, typename T>
struct SmhSeqEmit {
    SmhScratch s; uint32_t m; uint32_t *a; uint32_t *item;
    __device__ void operator()(uint64_t v) const
    {
        Rng g; g.seed(elem_hash<ALGO, VBITS>(v));
        const uint32_t it = (*item)++;
        for (uint32_t j = 0; j <= *a; j++) {
            uint64_t r;
            if (ALGO == ALGO_SUPER) r = g.r23();
            else r = sizeof(T) == 8 ? g.next64() : (uint64_t)g.next32();
            const uint64_t range = (uint64_t)(m - j);
            const uint32_t t = j + (uint32_t)rng_uint(g, range, uint_zone(range));
            if (s.q[j] != it) { s.q[j] = it; s.perm[j] = j; }
            if (s.q[t] != it) { s.q[t] = it; s.perm[t] = t; }
            const uint32_t tmp = s.perm[j]; s.perm[j] = s.perm[t]; s.perm[t] = tmp;
            const uint32_t sl = s.perm[j];
            const bool better = !s.filled[sl] || j < s.lvl[sl] || (j == s.lvl[sl] && r < s.rr[sl]);
            if (better) {
                const uint32_t jp = s.lvl[sl];
                s.filled[sl] = 1; s.lvl[sl] = j; s.rr[sl] = r;
                if (j < jp) { s.hist[jp]--; s.hist[j]++; while (s.hist[*a] == 0) (*a)--; }
            }
        }
    }
};
template <bool AA, class Emit>
__device__ void walk_genome_seq(const uint8_t *seq, const uint64_t *rec_start, const uint64_t *rec_len, uint64_t r0, uint64_t r1, uint32_t kq, const Emit &emit)
{
    const uint32_t k = kq_k(kq);
    const uint64_t rc_or = kq_rc_or(kq);
    for (uint64_t r = r0; r < r1; r++) {
        const uint64_t rb = rec_start[r], len = rec_len[r];
        if (len < k) continue;
        if (!AA) {
            const uint64_t mask = k == 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
            uint64_t fwd = 0, rc = 0;
            for (uint64_t i = 0; i < len; i++) {
                const uint64_t a = rb + i;
                const uint64_t c = (seq[a >> 2] >> (6 - 2 * (a & 3))) & 3;
                fwd = ((fwd << 2) | c) & mask;
                rc = (rc >> 2) | ((3 - c) << (2 * (k - 1))) | rc_or;
                if (i + 1 >= k) emit((fwd < rc ? fwd : rc) & mask);
            }
        } else {
            const uint64_t mask = ((uint64_t)1 << (5 * k)) - 1;
            uint64_t val = 0;
            for (uint64_t i = 0; i < len; i++) {
                val = ((val << 5) | c_aa_code[seq[rb + i] & 31]) & mask;
                if (i + 1 >= k) emit(val);
            }
        }
    }
}
// ---- cold path, parallel form: one workgroup per genome, every lane walks its own elements. The signature is an order-free
// lexicographic minimum over (level j, r_j) keys (SPEC 3.2), so the slots take atomicMin of packed keys in LDS and the histogram
// bound `a` of the paper (largest level still held by a slot; only a pruning device) is shared: hist[] is incremented BEFORE a key is
// published and decremented after it is displaced, so it never under-counts a level that holds a slot and `a` can only be lowered
// when that is safe. Lane-private Fisher-Yates state (perm / stamp arrays, m words each) lives in global scratch, element i of lane
// l at [i * CW_T + l]. u64 signatures need 64 + 16 bits per key: pass A orders by (level, top 48 bits of r), pass B re-walks the
// elements and takes the minimum of the full r among the visits that tie with the winner.
constexpr int CW_T = 256;
template <int ALGO, int VBITS, typename T, bool PASS_B>
struct SmhWgEmit {
    uint64_t *K, *K2; uint32_t *hist, *sa, *q, *perm; uint32_t m; uint32_t *stamp;
    static __device__ __forceinline__ uint64_t pack(uint32_t j, uint64_t r) { return sizeof(T) == 8 ? (((uint64_t)j << 48) | (r >> 16)) : (((uint64_t)j << 32) | r); }
    static __device__ __forceinline__ uint32_t level(uint64_t key) { return (uint32_t)(key >> (sizeof(T) == 8 ? 48 : 32)); }
    __device__ void operator()(uint64_t v, uint64_t, uint64_t) const
    {
        Rng g; g.seed(elem_hash<ALGO, VBITS>(v));
        const uint32_t st = (*stamp)++, l = threadIdx.x;
        for (uint32_t j = 0; j <= *(volatile uint32_t *)&sa[0]; j++) {
            uint64_t r;
            if (ALGO == ALGO_SUPER) r = g.r23();
            else r = sizeof(T) == 8 ? g.next64() : (uint64_t)g.next32();
            const uint64_t range = (uint64_t)(m - j);
            const uint32_t t = j + (uint32_t)rng_uint(g, range, uint_zone(range));
            const uint64_t ij = (uint64_t)j * CW_T + l, it = (uint64_t)t * CW_T + l;
            if (q[ij] != st) { q[ij] = st; perm[ij] = j; }
            if (q[it] != st) { q[it] = st; perm[it] = t; }
            const uint32_t tmp = perm[ij]; perm[ij] = perm[it]; perm[it] = tmp;
            const uint32_t sl = perm[ij];
            const uint64_t key = pack(j, r);
            if (PASS_B) { if (key == K[sl]) atomicMin((unsigned long long *)&K2[sl], (unsigned long long)r); continue; }
            if (key < K[sl]) {                                    // a stale (larger) read only costs a redundant attempt
                atomicAdd(&hist[j], 1u);
                const uint64_t old = atomicMin((unsigned long long *)&K[sl], (unsigned long long)key);
                if (key < old) {
                    const uint32_t jp = old == ~(uint64_t)0 ? m - 1 : level(old);
                    atomicSub(&hist[jp], 1u);                     // jp == j: same level, smaller r -> undoes the increment
                    uint32_t a0 = *(volatile uint32_t *)&sa[0];
                    const uint32_t a1 = a0;
                    while (a0 > 0 && *(volatile uint32_t *)&hist[a0] == 0) a0--;
                    if (a0 < a1) atomicMin(&sa[0], a0);
                } else atomicSub(&hist[j], 1u);                   // another lane got there first
            }
        }
    }
};
template <bool AA, int ALGO, int VBITS, typename T>
__global__ __launch_bounds__(CW_T) void k_smh_cold_wg(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len,
                                                      const uint64_t *__restrict__ rec_upre, const uint64_t *__restrict__ genome_rec_off,
                                                      const uint64_t *__restrict__ gen_units, const uint32_t *__restrict__ cold_list, uint32_t ncold, uint32_t k,
                                                      uint32_t m, uint32_t *__restrict__ lane_q, uint32_t *__restrict__ lane_perm,
                                                      unsigned long long *__restrict__ counter, void *__restrict__ sig)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_cw[];
    uint64_t *K = (uint64_t *)s_cw;
    uint64_t *K2 = sizeof(T) == 8 ? K + m : K;
    uint32_t *hist = (uint32_t *)(K + (sizeof(T) == 8 ? 2 * (size_t)m : (size_t)m));
    uint32_t *sa = hist + m;                                      // sa[0] = a, sa[1] = work item
    uint32_t *q = lane_q + (uint64_t)blockIdx.x * m * CW_T, *perm = lane_perm + (uint64_t)blockIdx.x * m * CW_T;
    uint32_t stamp = 0;                                           // unique per (lane, element) for the whole launch; the host fills q with 0xFF
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) sa[1] = (uint32_t)atomicAdd(counter, 1ull);
        __syncthreads();
        const uint32_t c = sa[1];
        if (c >= ncold) break;
        const uint64_t g = cold_list[c];
        for (uint32_t i = threadIdx.x; i < m; i += CW_T) { K[i] = ~(uint64_t)0; if (sizeof(T) == 8) K2[i] = ~(uint64_t)0; hist[i] = 0; }
        __syncthreads();
        if (threadIdx.x == 0) { hist[m - 1] = m; sa[0] = m - 1; }
        __syncthreads();
        const uint64_t r0 = genome_rec_off[g], r1 = genome_rec_off[g + 1], units = gen_units[g];
        {
            SmhWgEmit<ALGO, VBITS, T, false> emit{K, K2, hist, sa, q, perm, m, &stamp};
            walk_genome<AA>(seq, rec_start, rec_len, rec_upre, r0, r1, units, k, 0, 1, emit);
        }
        __syncthreads();
        if (sizeof(T) == 8) {
            SmhWgEmit<ALGO, VBITS, T, true> emit{K, K2, hist, sa, q, perm, m, &stamp};
            walk_genome<AA>(seq, rec_start, rec_len, rec_upre, r0, r1, units, k, 0, 1, emit);
            __syncthreads();
        }
        for (uint32_t i = threadIdx.x; i < m; i += CW_T) {
            const uint64_t key = K[i];
            if (ALGO == ALGO_SUPER) ((float *)sig)[g * (uint64_t)m + i] = key == ~(uint64_t)0 ? INFINITY : (float)(uint32_t)(key >> 32) + (float)(uint32_t)key * 0x1.0p-23f;
            else if (sizeof(T) == 8) ((T *)sig)[g * (uint64_t)m + i] = (T)K2[i];
            else ((T *)sig)[g * (uint64_t)m + i] = key == ~(uint64_t)0 ? (T)~(T)0 : (T)(uint32_t)key;
        }
    }
}

template <bool AA, int ALGO, int VBITS, typename T>
__global__ void k_smh_cold(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len,
                           const uint64_t *__restrict__ genome_rec_off, const uint32_t *__restrict__ cold_list, uint32_t ncold, uint32_t k, uint32_t m,
                           uint8_t *__restrict__ scratch, void *__restrict__ sig)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncold) return;
    const uint64_t g = cold_list[c];
    uint8_t *base = scratch + (uint64_t)c * 32 * m;
    SmhScratch s;
    s.rr = (uint64_t *)base; s.lvl = (uint32_t *)(base + 8 * (uint64_t)m); s.perm = s.lvl + m; s.hist = s.perm + m; s.q = s.hist + m;
    s.filled = (uint8_t *)(s.q + m);
    for (uint32_t i = 0; i < m; i++) { s.lvl[i] = m - 1; s.rr[i] = ~(uint64_t)0; s.q[i] = 0xFFFFFFFFu; s.hist[i] = 0; s.filled[i] = 0; }
    s.hist[m - 1] = m;
    uint32_t a = m - 1, item = 0;
    SmhSeqEmit<ALGO, VBITS, T> emit{s, m, &a, &item};
    walk_genome_seq<AA>(seq, rec_start, rec_len, genome_rec_off[g], genome_rec_off[g + 1], k, emit);
    for (uint32_t i = 0; i < m; i++) {
        if (ALGO == ALGO_SUPER) ((float *)sig)[g * (uint64_t)m + i] = s.filled[i] ? (float)s.lvl[i] + (float)(uint32_t)s.rr[i] * 0x1.0p-23f : INFINITY;
        else ((T *)sig)[g * (uint64_t)m + i] = s.filled[i] ? (T)s.rr[i] : (T)~(T)0;
    }
}

// launch geometry shared by the slot-min sketchers
struct MinGeom { uint32_t parts; bool use_lds; size_t lds; };
static MinGeom min_geom(gs_ctx *c, uint32_t m, size_t esz, uint64_t n_genomes, uint64_t avg_units)
{
    MinGeom g;
    g.lds = (size_t)m * esz;
    g.use_lds = g.lds <= 160 * 1024 - 256;
    g.parts = 1;
    if (n_genomes < (uint64_t)2 * c->n_cu) {
        g.parts = (uint32_t)((2 * (uint64_t)c->n_cu + n_genomes - 1) / n_genomes);
        uint64_t maxp = avg_units / SK_THREADS + 1;       // at least one full sweep per part
        if (g.parts > maxp) g.parts = (uint32_t)maxp;
        if (g.parts < 1) g.parts = 1;
    }
    return g;
}

template <int ALGO, int VBITS, typename T>
static int launch_min(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, const uint64_t *rec_start, const uint64_t *rec_len,
                      const uint64_t *rec_upre, const uint64_t *genome_rec_off, const uint64_t *gen_units, uint64_t n_genomes,
                      uint64_t avg_units, T *table)
{
    const uint32_t m = p->sketch_size;
    const uint64_t zone = uint_zone(m);
    const MinGeom ge = min_geom(c, m, sizeof(T), n_genomes, avg_units);
    if (!ge.use_lds || ge.parts > 1) GS_HIP_CHECK(hipMemsetAsync(table, 0xFF, (size_t)n_genomes * m * sizeof(T), c->stream));
    const bool aa = p->data_t == GS_DATA_AA;
    const size_t lds = ge.lds;
    // survivor queues of the filtered emitter (DNA, table in LDS): only where they do not cost a resident workgroup
    const size_t lds_f = ((lds + 15) & ~(size_t)15) + (size_t)(SK_THREADS / 64) * SKQ * 8;
    const size_t half_cu = (160 * 1024) / 2 / 1280 * 1280;
    // (and only for genomes long enough to warm the table up: >= 64 k-mers per slot; below that the survivor rate stays above 1/4)
    const bool filt = !aa && ge.use_lds && !(getenv("GS_SKETCH_FILTER") && !atoi(getenv("GS_SKETCH_FILTER"))) && avg_units * 32 >= (uint64_t)64 * m &&
                      (lds_f <= half_cu || (lds > half_cu && lds_f <= 160 * 1024 - 256));
    c->last_sketch[0] = filt; c->last_sketch[1] = ge.use_lds; c->last_sketch[2] = ge.parts; c->last_sketch[3] = 0;
    // GS_SKETCH_CAP: the c of the speculative bound (m / N)(ln m + c) of the filtered emitter (default 7: ~1 genome in 1000 walks twice); 0 = off
    const float cap_c = getenv("GS_SKETCH_CAP") ? (float)atof(getenv("GS_SKETCH_CAP")) : 7.0f;
    for (uint64_t g0 = 0; g0 < n_genomes; g0 += 65535) {       // grid.y limit
        uint64_t ng = n_genomes - g0 < 65535 ? n_genomes - g0 : 65535;
        dim3 grid(ge.parts, (uint32_t)ng), block(SK_THREADS);
        c->last_sketch[3]++;
        const uint64_t *gro = genome_rec_off + g0; const uint64_t *gu = gen_units + g0;
        T *tab = table + g0 * m;
        ProfScope ps(c, FAM_SKETCH);
#define GS_LAUNCH_MIN(AAV, LDSV, FV) do { if (!AAV && p->data_t == GS_DATA_DNA_FWD) GS_LAUNCH_MIN_RC(AAV, LDSV, FV, AAV ? 0 : 1); else GS_LAUNCH_MIN_RC(AAV, LDSV, FV, 0); } while (0)
#define GS_LAUNCH_MIN_RC(AAV, LDSV, FV, RC)                                                                             \
    do {                                                                                                                \
        auto kern = k_sketch_min<AAV, LDSV, ALGO, VBITS, T, FV, RC>;                                                    \
        const size_t l0_ = LDSV ? (FV ? lds_f : lds) : ((size_t)2 * m + 15) & ~(size_t)15;    /* slot table (+ survivor queues), or its 2-byte filter */  \
        const size_t l = std::min<size_t>(std::max<size_t>(l0_, c->sketch_min_lds), 160 * 1024 - 256);                                       \
        if (l > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l)); \
        hipLaunchKernelGGL(kern, grid, block, l, c->stream, seq, rec_start, rec_len, rec_upre, gro, gu, kq_of(p), m, zone, tab, cap_c); \
    } while (0)
        if (aa) { if (ge.use_lds) GS_LAUNCH_MIN(true, true, false); else GS_LAUNCH_MIN(true, false, false); }
        else if (!ge.use_lds) GS_LAUNCH_MIN(false, false, false);
        else if (filt) GS_LAUNCH_MIN(false, true, true);
        else GS_LAUNCH_MIN(false, true, false);
#undef GS_LAUNCH_MIN
#undef GS_LAUNCH_MIN_RC
        GS_HIP_CHECK(hipGetLastError());
    }
    return GS_OK;
}

// super / super2 driver: level-0 pass, finish, then the exact cold walk for the (rare) genomes with empty slots
template <int ALGO, int VBITS, typename T>
static int run_smh(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, const uint64_t *rec_start, const uint64_t *rec_len,
                   const uint64_t *rec_upre, const uint64_t *genome_rec_off, const uint64_t *gen_units, uint64_t n_genomes, uint64_t avg_units,
                   void *sig_out)
{
    const uint32_t m = p->sketch_size;
    PoolBuf table(c, 24), cold(c, 25);
    int rc;
    if ((rc = table.alloc((size_t)n_genomes * m * sizeof(T)))) return rc;
    if ((rc = cold.alloc(n_genomes))) return rc;
    if ((rc = launch_min<ALGO, VBITS, T>(c, p, seq, rec_start, rec_len, rec_upre, genome_rec_off, gen_units, n_genomes, avg_units, table.as<T>()))) return rc;
    hipLaunchKernelGGL((k_smh_finish<ALGO, T>), dim3((uint32_t)n_genomes), dim3(256), 0, c->stream, table.as<T>(), m, sig_out, cold.as<uint8_t>());
    GS_HIP_CHECK(hipGetLastError());
    std::vector<uint8_t> h(n_genomes);
    GS_HIP_CHECK(hipMemcpyAsync(h.data(), cold.p, n_genomes, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> list;
    for (uint64_t g = 0; g < n_genomes; g++) if (h[g]) list.push_back((uint32_t)g);
    const bool aa = p->data_t == GS_DATA_AA;
    // parallel form (one workgroup per genome) when the slot keys + level histogram fit in LDS
    const size_t lds_wg = (sizeof(T) == 8 ? 20 : 12) * (size_t)m + 16;
    if (!list.empty() && lds_wg <= 150 * 1024 && m <= 65535 && !getenv("GS_SMH_COLD_SERIAL")) {
        const uint32_t nc = (uint32_t)list.size();
        const uint32_t wgs = std::min<uint32_t>(nc, (uint32_t)c->n_cu * 2);
        PoolBuf dl(c, 26), lq(c, 27), lp(c, 37), cnt(c, 38);
        if ((rc = dl.alloc(4 * (size_t)nc))) return rc;
        if ((rc = lq.alloc((size_t)4 * wgs * m * CW_T))) return rc;
        if ((rc = lp.alloc((size_t)4 * wgs * m * CW_T))) return rc;
        if ((rc = cnt.alloc(64))) return rc;
        GS_HIP_CHECK(hipMemcpyAsync(dl.p, list.data(), 4 * (size_t)nc, hipMemcpyHostToDevice, c->stream));
        GS_HIP_CHECK(hipMemsetAsync(lq.p, 0xFF, (size_t)4 * wgs * m * CW_T, c->stream));
        GS_HIP_CHECK(hipMemsetAsync(cnt.p, 0, 8, c->stream));
#define GS_LAUNCH_COLDWG(AAV)                                                                                                  \
    do {                                                                                                                       \
        auto kern = k_smh_cold_wg<AAV, ALGO, VBITS, T>;                                                                        \
        if (lds_wg > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wg)); \
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(CW_T), lds_wg, c->stream, seq, rec_start, rec_len, rec_upre, genome_rec_off, gen_units, dl.as<uint32_t>(), nc, kq_of(p), m, \
                           lq.as<uint32_t>(), lp.as<uint32_t>(), cnt.as<unsigned long long>(), sig_out);                       \
    } while (0)
        if (aa) GS_LAUNCH_COLDWG(true); else GS_LAUNCH_COLDWG(false);
#undef GS_LAUNCH_COLDWG
        GS_HIP_CHECK(hipGetLastError());
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        return GS_OK;
    }
    // one lane per cold genome: take as many genomes per launch as 16 GB of per-lane scratch (32 m bytes each) allows - a launch lasts
    // as long as its slowest lane, so 4096-genome chunks serialised 25 launches for 10^5 small genomes
    const size_t per_launch = std::max<size_t>(4096, std::min<size_t>(((size_t)16 << 30) / ((size_t)32 * m), (size_t)1 << 20));
    for (size_t l0 = 0; l0 < list.size(); l0 += per_launch) {
        const uint32_t nc = (uint32_t)std::min<size_t>(per_launch, list.size() - l0);
        PoolBuf dl(c, 26), scratch(c, 27);
        if ((rc = dl.alloc(4 * (size_t)nc))) return rc;
        if ((rc = scratch.alloc((size_t)nc * 32 * m))) return rc;
        GS_HIP_CHECK(hipMemcpyAsync(dl.p, list.data() + l0, 4 * (size_t)nc, hipMemcpyHostToDevice, c->stream));
        if (aa) hipLaunchKernelGGL((k_smh_cold<true, ALGO, VBITS, T>), dim3((nc + 63) / 64), dim3(64), 0, c->stream, seq, rec_start, rec_len, genome_rec_off, dl.as<uint32_t>(), nc, kq_of(p), m, scratch.as<uint8_t>(), sig_out);
        else hipLaunchKernelGGL((k_smh_cold<false, ALGO, VBITS, T>), dim3((nc + 63) / 64), dim3(64), 0, c->stream, seq, rec_start, rec_len, genome_rec_off, dl.as<uint32_t>(), nc, kq_of(p), m, scratch.as<uint8_t>(), sig_out);
        GS_HIP_CHECK(hipGetLastError());
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    return GS_OK;
}

static int launch_oph(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, const uint64_t *rec_start, const uint64_t *rec_len,
                      const uint64_t *rec_upre, const uint64_t *genome_rec_off, const uint64_t *gen_units, uint64_t n_genomes,
                      uint64_t avg_units, uint32_t *table, uint32_t *win, float *sig)
{
    const uint32_t m = p->sketch_size;
    const uint64_t zone = uint_zone(m);
    int rc = launch_min<ALGO_OPTDENS, 64, uint32_t>(c, p, seq, rec_start, rec_len, rec_upre, genome_rec_off, gen_units, n_genomes, avg_units, table);
    if (rc) return rc;
    if (p->algo == GS_ALGO_OPTDENS)
        hipLaunchKernelGGL(k_oph_finish<ALGO_OPTDENS>, dim3((uint32_t)n_genomes), dim3(256), 0, c->stream, table, m, zone, win, sig);
    else
        hipLaunchKernelGGL(k_oph_finish<ALGO_REVOPTDENS>, dim3((uint32_t)n_genomes), dim3(256), 0, c->stream, table, m, zone, win, sig);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// =====================================================================================================
// prob (ProbMinHash3a, SPEC 3.3): multiset of canonical k-mers -> weighted per-slot argmin of (h, v).
//   1. every k-mer value is written to a buffer (composite key = genome-in-chunk << vbits | value), one global radix sort,
//      run-length encode -> distinct elements with their multiplicity w;
//   2. pass i = 1,2,..: every element still alive (w^-1 (i-1) <= max_b q[b]) replays its generator up to its i-th point
//      h = w^-1 (i-1) + w^-1 TE, slot b, and atomically lowers q[b]; the winners (h == q[b]) then race for the smallest v.
//      At genome sizes of interest all m slots are filled in pass 1 and only k-mers repeated >~ 30 times see pass 2.
// The result is the exact per-slot argmin of SPEC 3.3 (pruning is sound in any order).
// =====================================================================================================
struct ProbConst { double lambda, c1, c2, c3; };
__device__ __forceinline__ double em1_spec(double z)
{
    double t = 1.0 + z / 6.0;
    t = 1.0 + (z / 5.0) * t;
    t = 1.0 + (z / 4.0) * t;
    t = 1.0 + (z / 3.0) * t;
    t = 1.0 + (z / 2.0) * t;
    return z * t;
}
__device__ __forceinline__ double texp_sample(const ProbConst &t, Rng &g)
{
    double x = t.c1 * g.u64f();
    if (x < 1.0) return x;
    for (;;) {
        x = g.u64f();
        if (x < t.c2) return x;
        double y = 0.5 * g.u64f();
        if (y > 1.0 - x) { x = 1.0 - x; y = 1.0 - y; }
        if (x <= t.c3 * (1.0 - y)) return x;
        if (y * t.c1 <= 1.0 - x) return x;
        if ((y * t.c1) * t.lambda <= em1_spec(t.lambda * (1.0 - x))) return x;
    }
}
#define GS_INF_BITS 0x7FF0000000000000ULL

// k-mers per record -> exclusive prefix inside each genome
__global__ void k_kmer_prefix(const uint64_t *rec_len, const uint64_t *genome_rec_off, uint64_t n_genomes, uint32_t k, uint64_t *rec_kpre, uint64_t *gen_kmers)
{
    const uint64_t g = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);       // one wavefront per genome
    if (g >= n_genomes) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t r0 = genome_rec_off[g], r1 = genome_rec_off[g + 1];
    uint64_t base = 0;
    for (uint64_t rb = r0; rb < r1; rb += 64) {
        const uint64_t r = rb + lane;
        uint64_t u = 0;
        if (r < r1) { const uint64_t len = rec_len[r]; if (len >= k) u = len - k + 1; }
        uint64_t inc = u;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint64_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
        if (r < r1) rec_kpre[r] = base + inc - u;
        base += __shfl(inc, 63);
    }
    if (lane == 0) gen_kmers[g] = base;
}
struct ValueEmit {
    uint64_t *out; const uint64_t *rec_start; const uint64_t *rec_kpre; uint64_t base; uint64_t tag; uint32_t k;
    __device__ __forceinline__ void operator()(uint64_t v, uint64_t rec, uint64_t pos) const
    {
        out[base + rec_kpre[rec] + (pos - rec_start[rec] - (k - 1))] = tag | v;
    }
};
template <bool AA>
__global__ __launch_bounds__(SK_THREADS) void k_emit_values(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len,
                                                             const uint64_t *__restrict__ rec_upre, const uint64_t *__restrict__ rec_kpre,
                                                             const uint64_t *__restrict__ genome_rec_off, const uint64_t *__restrict__ gen_units,
                                                             const uint64_t *__restrict__ gen_base, uint64_t g0, uint32_t k, uint32_t vbits, uint64_t *__restrict__ out)
{
    const uint64_t gl = blockIdx.y, g = g0 + gl;
    ValueEmit emit{out, rec_start, rec_kpre, gen_base[gl], vbits >= 64 ? 0 : (gl << vbits), kq_k(k)};
    walk_genome<AA>(seq, rec_start, rec_len, rec_upre, genome_rec_off[g], genome_rec_off[g + 1], gen_units[g], k, blockIdx.x, gridDim.x, emit);
}
__global__ void k_prob_init(uint64_t *q, uint64_t *qprev, uint64_t *sig, uint64_t *sigpass, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        q[i] = GS_INF_BITS; qprev[i] = GS_INF_BITS; sig[i] = ~(uint64_t)0; sigpass[i] = ~(uint64_t)0;
    }
}
// pass `it`, phase A: i-th point of every live element -> q[b] = min
__global__ void k_prob_point(const uint64_t *__restrict__ ukey, const uint32_t *__restrict__ ucnt, uint64_t ne, uint32_t vbits, uint32_t m, uint64_t zone,
                             ProbConst pc, uint32_t it, const double *__restrict__ qmax, uint64_t *__restrict__ q, uint64_t *__restrict__ cand_h,
                             uint32_t *__restrict__ cand_b, uint32_t *__restrict__ wmax)
{
    const uint64_t vmask = vbits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << vbits) - 1);
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = ukey[e];
        const uint64_t gl = vbits >= 64 ? 0 : (key >> vbits), v = key & vmask;
        // largest multiplicity per genome (pass 1 only): read first - hundreds of millions of elements, a few hundred words; a stale
        // smaller value only costs a redundant atomic, unconditional atomics serialise on the same addresses
        if (wmax) { const uint32_t cnt = ucnt[e]; if (cnt > *(volatile uint32_t *)&wmax[gl]) atomicMax(&wmax[gl], cnt); }
        const double winv = 1.0 / (double)ucnt[e];
        const double base = winv * (double)(it - 1);
        uint32_t b = 0xFFFFFFFFu; uint64_t hb = 0;
        if (!(base > qmax[gl])) {
            Rng rg; rg.seed(v);                                 // prob: identity element hash (SPEC 2)
            double x = 0;
            for (uint32_t t = 0; t < it; t++) { x = texp_sample(pc, rg); b = (uint32_t)rng_uint(rg, (uint64_t)m, zone); }
            const double h = base + winv * x;
            hb = (uint64_t)__double_as_longlong(h);             // h >= 0: the bit pattern orders like the value
            uint64_t *slot = q + gl * (uint64_t)m + b;
            if (hb < *slot) atomicMin((unsigned long long *)slot, (unsigned long long)hb);
        }
        cand_b[e] = b; cand_h[e] = hb;
    }
}
// after pass 1: elements that can still reach a slot (w^-1 <= max q) are compacted into a list with their generator state,
// so that later passes touch only them and never replay
__global__ void k_prob_compact(const uint64_t *__restrict__ ukey, const uint32_t *__restrict__ ucnt, uint64_t ne, uint32_t vbits, uint32_t m, uint64_t zone,
                               ProbConst pc, const double *__restrict__ qmax, uint32_t cap, uint32_t *__restrict__ n_act, uint64_t *__restrict__ akey,
                               uint32_t *__restrict__ acnt, uint64_t *__restrict__ astate)
{
    const uint64_t vmask = vbits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << vbits) - 1);
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t key = ukey[e];
        const uint64_t gl = vbits >= 64 ? 0 : (key >> vbits);
        const double winv = 1.0 / (double)ucnt[e];
        if (winv * 1.0 > qmax[gl]) continue;
        const uint32_t pos = atomicAdd(n_act, 1u);
        if (pos >= cap) continue;                                  // overflow: the host falls back to replay mode
        Rng rg; rg.seed(key & vmask);
        (void)texp_sample(pc, rg); (void)rng_uint(rg, (uint64_t)m, zone);
        akey[pos] = key; acnt[pos] = ucnt[e];
        astate[pos] = rg.s0; astate[(uint64_t)cap + pos] = rg.s1; astate[2 * (uint64_t)cap + pos] = rg.s2; astate[3 * (uint64_t)cap + pos] = rg.s3;
    }
}
__global__ void k_prob_point_list(const uint64_t *__restrict__ akey, const uint32_t *__restrict__ acnt, uint32_t na, uint32_t cap, uint32_t vbits, uint32_t m,
                                  uint64_t zone, ProbConst pc, uint32_t it, const double *__restrict__ qmax, uint64_t *__restrict__ q, uint64_t *__restrict__ astate,
                                  uint64_t *__restrict__ cand_h, uint32_t *__restrict__ cand_b)
{
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < na; e += gridDim.x * blockDim.x) {
        const uint64_t key = akey[e];
        const uint64_t gl = vbits >= 64 ? 0 : (key >> vbits);
        const double winv = 1.0 / (double)acnt[e];
        const double base = winv * (double)(it - 1);
        uint32_t b = 0xFFFFFFFFu; uint64_t hb = 0;
        if (!(base > qmax[gl])) {
            Rng rg; rg.s0 = astate[e]; rg.s1 = astate[(uint64_t)cap + e]; rg.s2 = astate[2 * (uint64_t)cap + e]; rg.s3 = astate[3 * (uint64_t)cap + e];
            const double x = texp_sample(pc, rg);
            b = (uint32_t)rng_uint(rg, (uint64_t)m, zone);
            astate[e] = rg.s0; astate[(uint64_t)cap + e] = rg.s1; astate[2 * (uint64_t)cap + e] = rg.s2; astate[3 * (uint64_t)cap + e] = rg.s3;
            const double h = base + winv * x;
            hb = (uint64_t)__double_as_longlong(h);
            uint64_t *slot = q + gl * (uint64_t)m + b;
            if (hb < *slot) atomicMin((unsigned long long *)slot, (unsigned long long)hb);
        }
        cand_b[e] = b; cand_h[e] = hb;
    }
}
// phase B: among the points that reached the slot minimum the smallest value wins
__global__ void k_prob_claim(const uint64_t *__restrict__ ukey, uint64_t ne, uint32_t vbits, uint32_t m, const uint64_t *__restrict__ q,
                             const uint64_t *__restrict__ cand_h, const uint32_t *__restrict__ cand_b, uint64_t *__restrict__ sigpass)
{
    const uint64_t vmask = vbits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << vbits) - 1);
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = cand_b[e];
        if (b == 0xFFFFFFFFu) continue;
        const uint64_t key = ukey[e], gl = vbits >= 64 ? 0 : (key >> vbits);
        if (cand_h[e] == q[gl * (uint64_t)m + b]) atomicMin((unsigned long long *)&sigpass[gl * (uint64_t)m + b], (unsigned long long)(key & vmask));
    }
}
// phase C (one workgroup per genome): fold the pass winners into sig, recompute max_b q[b], decide whether the genome is done
__global__ __launch_bounds__(256) void k_prob_fold(uint32_t m, uint32_t it, uint64_t *__restrict__ q, uint64_t *__restrict__ qprev, uint64_t *__restrict__ sig,
                                                    uint64_t *__restrict__ sigpass, const uint32_t *__restrict__ wmax, double *__restrict__ qmax,
                                                    uint32_t *__restrict__ n_active)
{
    __shared__ unsigned long long s_max;
    const uint64_t g = blockIdx.x;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    unsigned long long loc = 0;
    for (uint32_t b = threadIdx.x; b < m; b += blockDim.x) {
        const uint64_t i = g * (uint64_t)m + b;
        const uint64_t sp = sigpass[i], qq = q[i];
        if (sp != ~(uint64_t)0) { if (qq != qprev[i]) sig[i] = sp; else if (sp < sig[i]) sig[i] = sp; sigpass[i] = ~(uint64_t)0; }
        qprev[i] = qq;
        if (qq > loc) loc = qq;
    }
    atomicMax(&s_max, loc);
    __syncthreads();
    if (threadIdx.x == 0) {
        const double qm = __longlong_as_double((long long)s_max);
        qmax[g] = qm;
        const uint32_t w = wmax[g];
        if (w > 0 && !((1.0 / (double)w) * (double)it > qm)) atomicAdd(n_active, 1u);      // some element may still reach a slot in pass it+1
    }
}
template <typename T>
__global__ void k_prob_write(const uint64_t *__restrict__ q, const uint64_t *__restrict__ sig, uint64_t n, T *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = q[i] == GS_INF_BITS ? (T)0 : (T)sig[i];
}

// =====================================================================================================
// prob, bucketed form (round 3; DESIGN.md 3.1 "ProbMinHash3a"). The global 64-bit radix sort above moves every k-mer across HBM ~14
// times only to learn multiplicities. Here a genome's k-mers are PARTITIONED once by the top bits of a multiplicative hash into
// buckets of ~PB_AVG values (count pass -> exact offsets -> scatter pass: two cheap walks, 8 B written per k-mer), and each bucket is
// then turned into (value, multiplicity) pairs by an LDS hash table (one 64-bit LDS CAS per k-mer) inside the kernel that also
// evaluates the first ProbMinHash point of every distinct element - so the k-mers cross HBM three times (read text, write bucket,
// read bucket).
//   k_prob_count    per (genome, part): bucket histogram in LDS; the first tile of every part also applies its k-mers to q[] with
//                   w = 1: x / 1 >= x / w, so those are UPPER bounds of the final slot minima (the true points arrive later and can
//                   only be lower) - they give the bucket kernel a finite rejection threshold from its first bucket on
//   k_prob_scan     per genome: bucket offsets, per-part scatter bases, thr = max_b q[b]
//   k_prob_scatter  per (genome, part): the same walk, values appended to their buckets through LDS cursors (no global atomics)
//   k_prob_buckets  persistent workgroups over all buckets of the chunk: LDS hash -> (v, w); an element whose first point
//                   h = w^-1 TE exceeds thr (any snapshot of max_b q[b] bounds the final one) cannot win a slot and stops after two
//                   SplitMix64 mixes (exact: strict >, ties must reach the claim); the rest lower q[b] and, when they are the slot's
//                   minimum at that moment, go on a short candidate list; elements with w^-1 <= thr may see pass 2 and go on the
//                   active list with their generator state
//   k_prob_claim_list + k_prob_fold as before; passes >= 2 run over the active list only.
// Genomes the scheme does not suit fall back to the sorted form above: fewer than 64 k-mers per slot (no warm-up: every element stays
// alive for many passes), more than PB_NBMAX * PB_AVG k-mers, or a bucket with more than PB_TAB distinct values (flagged on the device).
// =====================================================================================================
constexpr int PBK_T = 1024;        // lanes of the count / scatter kernels
constexpr int PBK_WPL = 4;         // units per lane per tile
constexpr int PB2_T = 512;         // lanes of the bucket kernel
constexpr int PB_AVG = 1536;       // k-mers per bucket aimed at
constexpr int PB_TAB = 4096;       // LDS hash entries per bucket
constexpr int PB_NBMAX = 16384;    // buckets per genome (LDS cursors of the scatter kernel: 64 kB)
constexpr int PB_CST = 256;        // candidate winners staged per bucket
constexpr int PB_SVQ = 3072;       // elements (table slots) queued for the full generator per bucket
// The bucket of a value comes from a BIJECTION of the vbits-bit values (multiplication by an odd constant modulo 2^vbits): bucket = its top
// lg bits, and the low sh = vbits - lg bits identify the value inside its bucket - 30 bits for k = 21 with 4096 buckets, so the bucket
// kernel's hash table holds 4-byte ids instead of 8-byte values (half the LDS, full-rate 32-bit LDS atomics) and gets the value back by
// multiplying with the inverse constant.
#define GS_PB_MUL 0x9E3779B97F4A7C15ULL
constexpr uint64_t pb_inverse(uint64_t a) { uint64_t x = a; for (int i = 0; i < 6; i++) x *= 2 - a * x; return x; }      // Newton: a odd, inverse modulo 2^64
constexpr uint64_t GS_PB_INV = pb_inverse(GS_PB_MUL);
static_assert(GS_PB_MUL * GS_PB_INV == 1ULL, "modular inverse");
__device__ __forceinline__ uint64_t pb_hash(uint64_t v, uint64_t vmask) { return (v * GS_PB_MUL) & vmask; }
__device__ __forceinline__ uint64_t pb_unhash(uint64_t hv, uint64_t vmask) { return (hv * GS_PB_INV) & vmask; }
__device__ __forceinline__ uint32_t pb_bucket(uint64_t v, uint64_t vmask, uint32_t sh) { return sh >= 64 ? 0u : (uint32_t)(pb_hash(v, vmask) >> sh); }
struct PbCountEmit {
    uint32_t *hist; uint32_t sh; uint64_t vmask;
    __device__ __forceinline__ void operator()(uint64_t v, uint64_t, uint64_t) const { atomicAdd(&hist[pb_bucket(v, vmask, sh)], 1u); }
};
struct PbWarmEmit {
    uint32_t *hist; uint32_t sh; uint64_t vmask; uint64_t *q; uint32_t m; uint64_t zone; ProbConst pc;
    __device__ __forceinline__ void operator()(uint64_t v, uint64_t, uint64_t) const
    {
        atomicAdd(&hist[pb_bucket(v, vmask, sh)], 1u);
        Rng rg; rg.seed(v);
        const double x = texp_sample(pc, rg);
        const uint32_t b = (uint32_t)rng_uint(rg, (uint64_t)m, zone);
        const uint64_t hb = (uint64_t)__double_as_longlong(x);
        if (hb < q[b]) atomicMin((unsigned long long *)&q[b], (unsigned long long)hb);
    }
};
struct PbScatterEmit {
    uint32_t *cur; uint32_t sh; uint64_t vmask; uint64_t *out;
    __device__ __forceinline__ void operator()(uint64_t v, uint64_t, uint64_t) const { out[atomicAdd(&cur[pb_bucket(v, vmask, sh)], 1u)] = v; }
};
// MODE 0: count (+ warm-up on the first tile), MODE 1: scatter
template <bool AA, int MODE>
__global__ __launch_bounds__(PBK_T) void k_prob_partition(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len,
                                                           const uint64_t *__restrict__ rec_upre, const uint64_t *__restrict__ genome_rec_off, const uint64_t *__restrict__ gen_units,
                                                           uint64_t g0, uint32_t kq, uint32_t vbits, const uint32_t *__restrict__ g_sh, const uint32_t *__restrict__ g_boff, uint32_t parts,
                                                           uint32_t *__restrict__ hist, uint64_t *__restrict__ q, uint32_t m, uint64_t zone, ProbConst pc,
                                                           uint64_t *__restrict__ vals, const uint64_t *__restrict__ g_vbase)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pb[];
    const uint32_t gl = blockIdx.y, part = blockIdx.x;
    const uint64_t g = g0 + gl;
    const uint32_t sh = g_sh[gl], NB = 1u << (vbits - sh);        // sh = vbits - log2(NB): bits of a hashed value below its bucket number
    uint32_t *hg = hist + (uint64_t)g_boff[gl] * parts + (uint64_t)part * NB;
    for (uint32_t b = threadIdx.x; b < NB; b += PBK_T) s_pb[b] = MODE == 0 ? 0u : hg[b];
    __syncthreads();
    const uint64_t r0 = genome_rec_off[g], r1 = genome_rec_off[g + 1], units = gen_units[g];
    const uint32_t k = kq_k(kq);
    const uint64_t mask = kmer_mask(AA, k), rc_or = kq_rc_or(kq);
    const uint32_t rcshift = 2 * (k - 1);
    const uint64_t TILE = (uint64_t)PBK_T * PBK_WPL;
    for (uint64_t t0 = (uint64_t)part * TILE; t0 < units; t0 += (uint64_t)parts * TILE) {
#pragma unroll 1
        for (int j = 0; j < PBK_WPL; j++) {
            const uint64_t f = t0 + (uint64_t)j * PBK_T + threadIdx.x;
            if (f >= units) continue;
            if (MODE == 0) {
                if (t0 == (uint64_t)part * TILE) { PbWarmEmit e{s_pb, sh, mask, q + (uint64_t)gl * m, m, zone, pc}; walk_unit<AA>(seq, rec_start, rec_len, rec_upre, r0, r1, f, k, mask, rcshift, rc_or, e); }
                else { PbCountEmit e{s_pb, sh, mask}; walk_unit<AA>(seq, rec_start, rec_len, rec_upre, r0, r1, f, k, mask, rcshift, rc_or, e); }
            } else { PbScatterEmit e{s_pb, sh, mask, vals + g_vbase[gl]}; walk_unit<AA>(seq, rec_start, rec_len, rec_upre, r0, r1, f, k, mask, rcshift, rc_or, e); }
        }
    }
    if (MODE == 0) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < NB; b += PBK_T) hg[b] = s_pb[b];
    }
}
// per genome: bucket sizes / starts, per-part scatter bases (hist is rewritten in place), thr = max_b q[b]
// Two-level form (g_shc != nullptr): the scatter goes through COARSE buckets first (the top lgc = lg / 2 bits of the bucket number, k_prob_partition<., 1> with
// these shifts and bases) and k_prob_refine spreads every (coarse bucket, part) slice over its fine buckets. Here: ccur[part][c] = where part `part` writes its
// values of coarse bucket c (coarse bucket c starts where its first fine bucket does; inside it the parts follow each other), ccnt = how many.
constexpr uint32_t PB_CCMAX = 8192;                            // parts x coarse buckets of a genome that the scan kernel can total in LDS
__global__ __launch_bounds__(1024) void k_prob_scan(uint32_t vbits, const uint32_t *__restrict__ g_sh, const uint32_t *__restrict__ g_boff, uint32_t parts, uint32_t *__restrict__ hist,
                                                     uint32_t *__restrict__ bstart, uint32_t *__restrict__ bsize, uint32_t *__restrict__ bgen, const uint64_t *__restrict__ q, uint32_t m,
                                                     uint64_t *__restrict__ thr, const uint32_t *__restrict__ g_shc, const uint32_t *__restrict__ g_coff, uint32_t *__restrict__ ccur,
                                                     uint32_t *__restrict__ ccnt)
{
    __shared__ uint32_t s_w[16]; __shared__ unsigned long long s_mx;
    __shared__ uint32_t s_cc[PB_CCMAX], s_cs[256];
    const uint32_t gl = blockIdx.x, sh = g_sh[gl], NB = 1u << (vbits - sh);
    const uint32_t b0 = g_boff[gl];
    uint32_t *hg = hist + (uint64_t)b0 * parts;
    const uint32_t fsh = g_shc ? g_shc[gl] - sh : 0, NC = NB >> fsh;      // fine buckets per coarse one = 1 << fsh
    if (g_shc) { for (uint32_t i = threadIdx.x; i < parts * NC; i += 1024) s_cc[i] = 0; }
    const uint32_t CH = (NB + 1023) / 1024;                    // consecutive buckets per lane
    uint32_t loc = 0;
    for (uint32_t c = 0; c < CH; c++) { const uint32_t b = threadIdx.x * CH + c; if (b < NB) for (uint32_t p = 0; p < parts; p++) loc += hg[(uint64_t)p * NB + b]; }
    uint32_t inc = loc;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
    if (lane == 63) s_w[wv] = inc;
    if (threadIdx.x == 0) s_mx = 0;
    __syncthreads();
    uint32_t run = inc - loc;
    for (uint32_t w = 0; w < wv; w++) run += s_w[w];
    for (uint32_t c = 0; c < CH; c++) {
        const uint32_t b = threadIdx.x * CH + c;
        if (b >= NB) break;
        bstart[b0 + b] = run; bgen[b0 + b] = gl;
        if (g_shc && (b & ((1u << fsh) - 1u)) == 0) s_cs[b >> fsh] = run;
        uint32_t tot = 0;
        for (uint32_t p = 0; p < parts; p++) {
            const uint32_t x = hg[(uint64_t)p * NB + b]; hg[(uint64_t)p * NB + b] = run + tot; tot += x;
            if (g_shc && x) atomicAdd(&s_cc[p * NC + (b >> fsh)], x);
        }
        bsize[b0 + b] = tot; run += tot;
    }
    if (g_shc) {
        __syncthreads();
        const uint32_t c0 = g_coff[gl];
        for (uint32_t i = threadIdx.x; i < parts * NC; i += 1024) {
            const uint32_t p = i / NC, cb = i % NC;
            uint32_t base = s_cs[cb];
            for (uint32_t p2 = 0; p2 < p; p2++) base += s_cc[p2 * NC + cb];
            ccur[(uint64_t)c0 * parts + i] = base; ccnt[(uint64_t)c0 * parts + i] = s_cc[i];
        }
    }
    unsigned long long mx = 0;
    for (uint32_t i = threadIdx.x; i < m; i += 1024) { const unsigned long long x = q[(uint64_t)gl * m + i]; mx = x > mx ? x : mx; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
    if (lane == 0) atomicMax(&s_mx, mx);
    __syncthreads();
    if (threadIdx.x == 0) thr[gl] = s_mx;
}
// second level of the partition: one workgroup per (coarse bucket, part) slice of a genome. The slice is read front to back (coalesced) and each value is
// appended to its fine bucket through an LDS cursor that starts at that part's place in the bucket (the same per-(part, bucket) bases the one-level
// scatter used). A workgroup writes to at most 128 open streams of a few hundred consecutive values each: the L2 completes their lines before they are
// evicted, where the one-level scatter's 4096 streams per workgroup left it as 32-byte sector writes (WRITE_SIZE 3.9x the values, profiles/r03_prob_pmc.txt).
constexpr int PBR_T = 512, PBR_V = 8, PBR_TILE = PBR_T * PBR_V;
__global__ __launch_bounds__(PBR_T) void k_prob_refine(const uint64_t *__restrict__ tmp, uint64_t *__restrict__ vals, const uint64_t *__restrict__ g_vbase, uint32_t vbits,
                                                       const uint32_t *__restrict__ g_sh, const uint32_t *__restrict__ g_boff, const uint32_t *__restrict__ g_shc,
                                                       const uint32_t *__restrict__ g_coff, uint32_t parts, const uint32_t *__restrict__ hist, const uint32_t *__restrict__ ccur,
                                                       const uint32_t *__restrict__ ccnt, uint32_t store32)
{
    // store32: every genome's in-bucket id fits 4 bytes - the id (low sh bits of the hashed value) is what k_prob_buckets keeps in its table, so that is
    // what is written, element i of the chunk at ((uint32_t *)vals)[i]
    // a tile of 4096 values is counting-sorted by fine bucket in LDS and written out run by run: consecutive lanes store consecutive values of one
    // bucket (8-byte stores scattered over 64 streams were bound by the number of write requests, not by bytes: 8.7 ms per 1.3e9 values)
    __shared__ uint64_t s_val[PBR_TILE];
    __shared__ uint8_t s_bk[PBR_TILE];
    __shared__ uint32_t s_cur[256], s_cnt[256], s_start[256];
    const uint32_t gl = blockIdx.y, sh = g_sh[gl], shc = g_shc[gl], fsh = shc - sh, NB = 1u << (vbits - sh), NC = NB >> fsh, NF = 1u << fsh;
    const uint32_t cb = blockIdx.x / parts, part = blockIdx.x % parts;
    if (cb >= NC) return;
    const uint32_t *hg = hist + (uint64_t)g_boff[gl] * parts + (uint64_t)part * NB + (uint64_t)cb * NF;
    for (uint32_t f = threadIdx.x; f < NF; f += PBR_T) s_cur[f] = hg[f];
    const uint64_t ci = (uint64_t)g_coff[gl] * parts + (uint64_t)part * NC + cb;
    const uint32_t n = ccnt[ci];
    const uint64_t *src = tmp + g_vbase[gl] + ccur[ci];
    uint64_t *dst = vals + g_vbase[gl];
    uint32_t *dst32 = (uint32_t *)vals + g_vbase[gl];
    const uint64_t vmask = vbits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << vbits) - 1);
    const uint64_t idmask = sh >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << sh) - 1);
    for (uint32_t i0 = 0; i0 < n; i0 += PBR_TILE) {
        const uint32_t tn = n - i0 < (uint32_t)PBR_TILE ? n - i0 : (uint32_t)PBR_TILE;
        uint64_t v[PBR_V]; uint32_t f[PBR_V], rk[PBR_V];
#pragma unroll
        for (int u = 0; u < PBR_V; u++) { const uint32_t i = u * PBR_T + threadIdx.x; v[u] = i < tn ? src[i0 + i] : 0; }
        if (threadIdx.x < 256) s_cnt[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PBR_V; u++) {
            const uint32_t i = u * PBR_T + threadIdx.x;
            f[u] = pb_bucket(v[u], vmask, sh) & (NF - 1u);
            rk[u] = i < tn ? atomicAdd(&s_cnt[f[u]], 1u) : 0u;
        }
        __syncthreads();
        if (threadIdx.x < 64) {                                   // exclusive prefix of the <= 256 bucket counts: four per lane of one wavefront
            uint32_t c4[4], loc = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { c4[j] = s_cnt[threadIdx.x * 4 + j]; loc += c4[j]; }
            uint32_t inc = loc;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)threadIdx.x >= o) inc += y; }
            uint32_t run = inc - loc;
#pragma unroll
            for (int j = 0; j < 4; j++) { s_start[threadIdx.x * 4 + j] = run; run += c4[j]; }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PBR_V; u++) {
            const uint32_t i = u * PBR_T + threadIdx.x;
            if (i < tn) { const uint32_t pos = s_start[f[u]] + rk[u]; s_val[pos] = v[u]; s_bk[pos] = (uint8_t)f[u]; }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PBR_V; u++) {
            const uint32_t i = u * PBR_T + threadIdx.x;
            if (i < tn) {
                const uint32_t fb = s_bk[i]; const uint32_t at = s_cur[fb] + (i - s_start[fb]);
                if (store32) dst32[at] = (uint32_t)(pb_hash(s_val[i], vmask) & idmask); else dst[at] = s_val[i];
            }
        }
        __syncthreads();
        if (threadIdx.x < 256) s_cur[threadIdx.x] += s_cnt[threadIdx.x];
    }
}
struct PbLists {
    uint64_t *cand_v, *cand_h, *cand_gb; uint32_t cand_cap, ovf_cap; uint32_t *n_cand, *seg_n;      // [0, cand_cap): per-workgroup segments; [cand_cap, + ovf_cap): shared overflow
    uint64_t *akey; uint32_t *agl, *acnt; uint64_t *astate; uint32_t act_cap; uint32_t *n_act;
    unsigned long long *prof;                                     // GS_PROB_PROFILE: cycle stamps of workgroup 0 per phase (nullptr otherwise)
};
// KT = uint32_t: the table holds the sh-bit id of a value inside its bucket (sh <= 31 for every genome of the launch; ~0 = empty);
// KT = uint64_t: it holds the value itself (k = 32, AA k = 12, or genomes with so few buckets that an id needs 32 bits).
template <typename KT>
__global__ __launch_bounds__(PB2_T, 6) void k_prob_buckets(const uint64_t *__restrict__ vals, const uint64_t *__restrict__ g_vbase, const uint32_t *__restrict__ bstart,
                                                           const uint32_t *__restrict__ bsize, uint32_t vbits, const uint32_t *__restrict__ g_sh, const uint32_t *__restrict__ g_boff,
                                                           uint32_t ng, uint32_t lg_max, uint32_t m, uint64_t zone,
                                                           ProbConst pc, uint64_t *__restrict__ q, uint64_t *__restrict__ thr, uint32_t *__restrict__ wmax, PbLists L,
                                                           uint32_t *__restrict__ ovf, uint32_t stored32)
{
    // stored32 (KT = uint32_t only): the partition left the 4-byte in-bucket ids in `vals` (element i of the chunk at ((uint32_t *)vals)[i]) instead of
    // the 8-byte values - half the bytes written by the refinement and read here; the value comes back through the inverse multiplication
    // LDS per workgroup (4-byte ids): 16 kB table + 8 kB duplicate counts + 6 kB queue + 5 kB candidates = 35 kB. What the kernel spends
    // its time on (GS_PROB_PROFILE, cycles per bucket of 1220 keys on 512 lanes, before / after this form): LDS atomics of the insert
    // 6360 / see DESIGN (one 32-bit CAS per k-mer instead of a 64-bit CAS plus an add: the LDS pipeline is shared by the whole CU, so
    // residency does not help there), the cheap test 2930 (its two SplitMix64 mixes now run during the insert, under the LDS wait), the
    // threshold read 1390 (now fetched one bucket ahead), the few full points 4600 (global read + atomicMin latency).
    __shared__ KT tab[PB_TAB];
    __shared__ uint32_t dup[PB_TAB / 2];                          // 16-bit counts of the REPEATED occurrences, two per word (65535 saturates: flagged)
    __shared__ unsigned long long s_mx;
    // elements that pass the cheap threshold test are compacted into an LDS queue (of table slots) so that the full generator (truncated
    // exponential + uniform slot) runs on dense wavefronts; possible winners are staged too and go to this workgroup's PRIVATE segment
    // of the candidate list (no global counter: one bumped per candidate by thousands of lanes serialised in the L2)
    __shared__ uint16_t sv_s[PB_SVQ];
    __shared__ uint64_t sc_v[PB_CST], sc_h[PB_CST]; __shared__ uint32_t sc_b[PB_CST];
    __shared__ uint32_t s_nc, s_ns;
    const KT EMPTY = (KT)~(KT)0;
    const uint64_t EMPTY64 = ~(uint64_t)0;
    const uint64_t vmask = vbits >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << vbits) - 1);
    constexpr int KPL = 6;                                        // keys per lane held in registers (buckets of up to KPL * PB2_T keys)
    const uint32_t seg = L.cand_cap / gridDim.x;                  // this workgroup's share of the candidate list
    uint32_t my_nc = 0;
    // Work items in BUCKET-major order over the chunk: (position j of 2^lg_max, genome) - a genome with fewer buckets takes part at every
    // (2^lg_max / NB)-th position. All genomes advance together, so a genome's buckets are spread over the whole launch and its
    // threshold has time to tighten (the first buckets see max_b q[b] of the warm-up, the last ones nearly the final one); genome-major
    // order had ~1000 workgroups finish a genome's 8192 buckets within microseconds of each other, all under the loosest bound.
    // The description of the item after this one (and its genome's threshold: a slightly older bound is still a bound) is fetched while
    // this one is worked on.
    const uint64_t n_items = (uint64_t)ng << lg_max;
    auto describe = [&](uint64_t idx, uint32_t &gl_, uint32_t &n_, uint32_t &st_, uint64_t &vb_, uint32_t &pos_, uint32_t &sh_, uint32_t &bk_, uint64_t &thr_) {
        n_ = 0; gl_ = 0; st_ = 0; vb_ = 0; pos_ = 1; sh_ = 0; bk_ = 0; thr_ = 0;
        if (idx >= n_items) return;
        const uint32_t j = (uint32_t)(idx / ng), g = (uint32_t)(idx % ng);
        const uint32_t sh = g_sh[g], lg = vbits - sh, rs = lg_max - lg;          // this genome has 2^lg buckets
        if (j & ((1u << rs) - 1u)) return;
        const uint32_t fb = g_boff[g] + (j >> rs);
        gl_ = g; n_ = bsize[fb]; st_ = bstart[fb]; vb_ = g_vbase[g]; pos_ = j; sh_ = sh; bk_ = j >> rs;
        thr_ = __hip_atomic_load(&thr[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    uint64_t item = blockIdx.x;
    uint32_t n_gl, n_n, n_st, n_pos, n_sh, n_bk; uint64_t n_vb, n_thr;
    describe(item, n_gl, n_n, n_st, n_vb, n_pos, n_sh, n_bk, n_thr);
    for (; item < n_items; item += gridDim.x) {
        const uint32_t gl = n_gl, n = n_n, jpos = n_pos, sh = n_sh, bk = n_bk;
        const uint64_t *keys = vals + n_vb + n_st;
        const uint32_t *keys32 = (const uint32_t *)vals + n_vb + n_st;
        const bool st32 = sizeof(KT) == 4 && stored32;
        uint64_t *qg = q + (uint64_t)gl * m;
        uint64_t thr_b = n_thr;
        uint64_t kreg[KPL];
#pragma unroll
        for (int u = 0; u < KPL; u++) { const uint32_t i = u * PB2_T + threadIdx.x; kreg[u] = i < n ? (st32 ? (uint64_t)keys32[i] : keys[i]) : EMPTY64; }      // st32: the id, for now
        describe(item + gridDim.x, n_gl, n_n, n_st, n_vb, n_pos, n_sh, n_bk, n_thr);
        if (n == 0) continue;                                      // (workgroup-uniform) nothing of this genome at this position
        const bool pf = L.prof && blockIdx.x == 0 && threadIdx.x == 0;
        long long t0 = pf ? clock64() : 0, t1;
#define GS_PSTAMP(i) do { if (pf) { t1 = clock64(); atomicAdd(&L.prof[i], (unsigned long long)(t1 - t0)); t0 = t1; } } while (0)
        __syncthreads();                                           // the previous bucket's LDS is dead
        for (uint32_t s = threadIdx.x; s < PB_TAB; s += PB2_T) tab[s] = EMPTY;
        for (uint32_t s = threadIdx.x; s < PB_TAB / 2; s += PB2_T) dup[s] = 0;
        if (threadIdx.x == 0) { s_nc = 0; s_ns = 0; s_mx = 0; }
        // rejection threshold: any snapshot of max_b q[b] bounds the final maximum (q only ever decreases). 32 times per genome a bucket
        // rescans the genome's q[] (m loads, eight in flight per lane) and publishes the new bound; everybody else reads the published one.
        if ((jpos & ((1u << (lg_max > 5 ? lg_max - 5 : 0)) - 1u)) == 0) {
            unsigned long long mx = 0;
            for (uint32_t i0 = 0; i0 < m; i0 += 8 * PB2_T) {
                unsigned long long x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const uint32_t i = i0 + u * PB2_T + threadIdx.x; x[u] = i < m ? __hip_atomic_load(&qg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull; }
#pragma unroll
                for (int u = 0; u < 8; u++) mx = x[u] > mx ? x[u] : mx;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
            __syncthreads();
            if ((threadIdx.x & 63) == 0) atomicMax(&s_mx, mx);
            __syncthreads();
            const unsigned long long t = s_mx;
            if (threadIdx.x == 0) atomicMin((unsigned long long *)&thr[gl], t);
            if (t < thr_b) thr_b = t;
        }
        const double thr_d = __longlong_as_double((long long)thr_b);
        __syncthreads();
        GS_PSTAMP(0);
        // ---- (value, multiplicity) pairs: LDS hash table, ONE atomic per k-mer. The lane whose CAS created an entry OWNS that element
        //      (count 1); a later occurrence of the same value finds it there and bumps the entry's duplicate count instead (a second
        //      atomic, for repeats only). After the barrier the owners - dense over the lanes, unlike the 70 %-empty table - go on with
        //      multiplicity 1 + duplicates. The first draw of every key (two SplitMix64 mixes) is computed here, under the LDS wait.
        bool over = false;
        uint32_t own[KPL]; double x0r[KPL];
        const uint64_t idmask = sizeof(KT) == 4 ? (((uint64_t)1 << sh) - 1) : ~(uint64_t)0;
        auto insert_at = [&](KT id, uint32_t s) -> uint32_t {
            for (uint32_t probe = 0; probe < PB_TAB; probe++) {
                const KT old = atomicCAS(&tab[s], EMPTY, id);
                if (old == EMPTY) return s;
                if (old == id) {
                    const uint32_t before = atomicAdd(&dup[s >> 1], 1u << ((s & 1) * 16));
                    if (((before >> ((s & 1) * 16)) & 0xFFFFu) == 0xFFFFu) over = true;      // a k-mer 65537 times in one genome: 16 bits wrapped
                    return 0xFFFFFFFFu;
                }
                s = (s + 1) & (PB_TAB - 1);
            }
            over = true;
            return 0xFFFFFFFFu;
        };
        auto insert = [&](uint64_t v) -> uint32_t {
            const KT id = sizeof(KT) == 4 ? (KT)(pb_hash(v, vmask) & idmask) : (KT)v;
            return insert_at(id, (uint32_t)((v * 0xD6E8FEB86659FD93ULL) >> 40) & (PB_TAB - 1));
        };
        // stored ids: the table slot comes from the id's own top bits (inside a bucket the ids are a bijection of the values and the upper bits of a
        // multiplicative hash's window are its best mixed) - no second multiplication, and the value is only needed for the draw
        auto insert_id = [&](uint32_t id) -> uint32_t { return insert_at((KT)id, (sh > 12 ? id >> (sh - 12) : id) & (uint32_t)(PB_TAB - 1)); };
        auto first_draw = [&](uint64_t v) -> double {             // x = c1 * U64f from two of the four state words; when x < 1 it IS the truncated exponential (SPEC 3.3)
            const uint64_t s0 = splitmix_mix(v + GS_GAMMA), s3 = splitmix_mix(v + 4 * GS_GAMMA);
            return pc.c1 * ((double)((rotl64(s0 + s3, 23) + s0) >> 12) * 0x1.0p-52);
        };
#pragma unroll
        for (int u = 0; u < KPL; u++) {
            own[u] = 0xFFFFFFFFu; x0r[u] = 0.0;
            if ((uint32_t)(u * PB2_T) + threadIdx.x < n) {
                if (st32) { const uint32_t id = (uint32_t)kreg[u]; own[u] = insert_id(id); x0r[u] = first_draw(pb_unhash(((uint64_t)bk << sh) | (uint64_t)id, vmask)); }
                else { own[u] = insert(kreg[u]); x0r[u] = first_draw(kreg[u]); }
            }
        }
        // (buckets beyond KPL * PB2_T keys - heavy repeats - : the tail's owners are found by the table sweep below)
        const bool tail = n > (uint32_t)(KPL * PB2_T);
        for (uint32_t i = KPL * PB2_T + threadIdx.x; i < n; i += PB2_T) { if (st32) (void)insert_id(keys32[i]); else (void)insert(keys[i]); }
        if (over) ovf[gl] = 1;                                  // table full or a count wrapped: the host redoes this genome the sorted way
        __syncthreads();
        GS_PSTAMP(1);
        // ---- cheap test of every distinct element; the ones that may matter go to the queue
        uint32_t wloc = 0;
        auto count_of = [&](uint32_t s) -> uint32_t { return 1u + ((dup[s >> 1] >> ((s & 1) * 16)) & 0xFFFFu); };
        auto value_of = [&](uint32_t s) -> uint64_t {
            if (sizeof(KT) == 4) return pb_unhash(((uint64_t)bk << sh) | (uint64_t)tab[s], vmask);
            return (uint64_t)tab[s];
        };
        auto full_point = [&](uint32_t s) {
            const uint64_t v = value_of(s);
            const uint32_t w = count_of(s);
            const double winv = w == 1 ? 1.0 : 1.0 / (double)w;
            const bool alive2 = !(winv > thr_d);
            Rng rg; rg.seed(v);
            const double x = texp_sample(pc, rg);
            const uint32_t b = (uint32_t)rng_uint(rg, (uint64_t)m, zone);
            const double h = 0.0 + winv * x;
            if (!(h > thr_d)) {
                const uint64_t hb = (uint64_t)__double_as_longlong(h);
                uint64_t *slot = qg + b;
                if (hb <= __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    // not above the slot's minimum a moment ago: a possible winner. The atomicMin's answer is not waited for (a second memory round trip per
                    // bucket): whoever passes the read is listed - a superset of those the atomic would confirm, and the claim only takes candidates whose
                    // point equals the slot's final minimum
                    (void)__hip_atomic_fetch_min((unsigned long long *)slot, (unsigned long long)hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t sp = atomicAdd(&s_nc, 1u);
                    if (sp < (uint32_t)PB_CST) { sc_v[sp] = v; sc_h[sp] = hb; sc_b[sp] = b; }
                    else {                                       // staging full (first buckets of a genome): the shared overflow region behind the segments
                        const uint32_t pos = atomicAdd(L.n_cand, 1u);
                        if (pos < L.ovf_cap) { const uint32_t o = L.cand_cap + pos; L.cand_v[o] = v; L.cand_h[o] = hb; L.cand_gb[o] = (uint64_t)gl * m + b; }
                    }
                }
            }
            if (alive2) {                                        // may still reach a slot in pass 2 (superset: thr >= the final max q)
                const uint32_t pos = atomicAdd(L.n_act, 1u);
                if (pos < L.act_cap) {
                    L.akey[pos] = v; L.agl[pos] = gl; L.acnt[pos] = w;
                    L.astate[pos] = rg.s0; L.astate[(uint64_t)L.act_cap + pos] = rg.s1; L.astate[2 * (uint64_t)L.act_cap + pos] = rg.s2; L.astate[3 * (uint64_t)L.act_cap + pos] = rg.s3;
                }
            }
        };
        auto cheap_test = [&](double x0, uint32_t s) {
            const uint32_t w = count_of(s);
            wloc = w > wloc ? w : wloc;
            const double winv = w == 1 ? 1.0 : 1.0 / (double)w;      // (1.0 / 1.0 is exact: the common case skips the f64 division)
            if (x0 < 1.0 && winv * x0 > thr_d && winv > thr_d) return;      // cannot win a slot (strict: ties must reach the claim), dead in pass 2
            const uint32_t sp = atomicAdd(&s_ns, 1u);
            if (sp < (uint32_t)PB_SVQ) sv_s[sp] = (uint16_t)s;
            else full_point(s);                                  // queue full (a first bucket under a loose threshold): straight away
        };
        if (!tail) {
#pragma unroll
            for (int u = 0; u < KPL; u++) if (own[u] != 0xFFFFFFFFu) cheap_test(x0r[u], own[u]);
        } else {
            for (uint32_t s = threadIdx.x; s < PB_TAB; s += PB2_T) if (tab[s] != EMPTY) cheap_test(first_draw(value_of(s)), s);
        }
        __syncthreads();
        GS_PSTAMP(2);
        const uint32_t ns = s_ns < (uint32_t)PB_SVQ ? s_ns : (uint32_t)PB_SVQ;
        if (pf) { atomicAdd(&L.prof[6], (unsigned long long)ns); atomicAdd(&L.prof[7], 1ull); atomicAdd(&L.prof[8], (unsigned long long)n); }
        for (uint32_t i = threadIdx.x; i < ns; i += PB2_T) full_point(sv_s[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)wloc, o); wloc = y > wloc ? y : wloc; }
        // (wmax starts at 1 for every genome of a bucketed chunk: the common all-unique bucket sends nothing; the read goes to the L2
        // like the atomics do - a plain load may come from a stale L1 line and would let every wave of every bucket send an atomic to
        // the same address)
        if ((threadIdx.x & 63) == 0 && wloc > 1 && wloc > __hip_atomic_load(&wmax[gl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&wmax[gl], wloc);
        __syncthreads();
        GS_PSTAMP(3);
        const uint32_t nst = s_nc < (uint32_t)PB_CST ? s_nc : (uint32_t)PB_CST;
        if (pf) atomicAdd(&L.prof[9], (unsigned long long)s_nc);
        if (nst) {
            if (my_nc + nst > seg) { if (threadIdx.x == 0) atomicMax(L.n_cand, 0xFFFFFFFFu); }       // segment full: the host redoes the chunk the sorted way
            else {
                const uint32_t cb = blockIdx.x * seg + my_nc;
                for (uint32_t i = threadIdx.x; i < nst; i += PB2_T) { L.cand_v[cb + i] = sc_v[i]; L.cand_h[cb + i] = sc_h[i]; L.cand_gb[cb + i] = (uint64_t)gl * m + sc_b[i]; }
                my_nc += nst;
            }
        }
        GS_PSTAMP(4);
#undef GS_PSTAMP
    }
    if (threadIdx.x == 0) L.seg_n[blockIdx.x] = my_nc;
}
__global__ void k_prob_claim_list(const uint64_t *__restrict__ cand_v, const uint64_t *__restrict__ cand_h, const uint64_t *__restrict__ cand_gb, const uint32_t *__restrict__ seg_n,
                                  uint32_t seg, uint32_t nseg, uint32_t cand_cap, const uint32_t *__restrict__ n_ovf, uint32_t ovf_cap, const uint64_t *__restrict__ q,
                                  uint64_t *__restrict__ sigpass)
{
    // one workgroup per segment (the bucket kernel's workgroups each filled their own), then the shared overflow region, dealt over all workgroups (it holds
    // the first buckets' candidates - every point is one while a slot is empty: millions, 13 ms when one workgroup walked them)
    for (uint32_t sgm = blockIdx.x; sgm < nseg; sgm += gridDim.x) {
        const uint32_t base = sgm * seg, n = seg_n[sgm];
        for (uint32_t e = threadIdx.x; e < n; e += blockDim.x) {
            const uint64_t gb = cand_gb[base + e];
            if (cand_h[base + e] == q[gb]) atomicMin((unsigned long long *)&sigpass[gb], (unsigned long long)cand_v[base + e]);
        }
    }
    uint32_t n = *n_ovf;
    if (n > ovf_cap) n = ovf_cap;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const uint64_t gb = cand_gb[cand_cap + e];
        if (cand_h[cand_cap + e] == q[gb]) atomicMin((unsigned long long *)&sigpass[gb], (unsigned long long)cand_v[cand_cap + e]);
    }
}
// passes >= 2 over the active list (generator state carried from point to point)
__global__ void k_prob_point_act(const uint64_t *__restrict__ akey, const uint32_t *__restrict__ agl, const uint32_t *__restrict__ acnt, uint32_t na, uint32_t cap, uint32_t m,
                                 uint64_t zone, ProbConst pc, uint32_t it, const double *__restrict__ qmax, uint64_t *__restrict__ q, uint64_t *__restrict__ astate,
                                 uint64_t *__restrict__ cand_h, uint32_t *__restrict__ cand_b)
{
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < na; e += gridDim.x * blockDim.x) {
        const uint32_t gl = agl[e];
        const double winv = 1.0 / (double)acnt[e];
        const double base = winv * (double)(it - 1);
        uint32_t b = 0xFFFFFFFFu; uint64_t hb = 0;
        if (!(base > qmax[gl])) {
            Rng rg; rg.s0 = astate[e]; rg.s1 = astate[(uint64_t)cap + e]; rg.s2 = astate[2 * (uint64_t)cap + e]; rg.s3 = astate[3 * (uint64_t)cap + e];
            const double x = texp_sample(pc, rg);
            b = (uint32_t)rng_uint(rg, (uint64_t)m, zone);
            astate[e] = rg.s0; astate[(uint64_t)cap + e] = rg.s1; astate[2 * (uint64_t)cap + e] = rg.s2; astate[3 * (uint64_t)cap + e] = rg.s3;
            const double h = base + winv * x;
            hb = (uint64_t)__double_as_longlong(h);
            uint64_t *slot = q + (uint64_t)gl * m + b;
            if (hb < *slot) atomicMin((unsigned long long *)slot, (unsigned long long)hb);
        }
        cand_b[e] = b; cand_h[e] = hb;
    }
}
__global__ void k_prob_claim_act(const uint64_t *__restrict__ akey, const uint32_t *__restrict__ agl, uint32_t na, uint32_t m, const uint64_t *__restrict__ q,
                                 const uint64_t *__restrict__ cand_h, const uint32_t *__restrict__ cand_b, uint64_t *__restrict__ sigpass)
{
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < na; e += gridDim.x * blockDim.x) {
        const uint32_t b = cand_b[e];
        if (b == 0xFFFFFFFFu) continue;
        const uint64_t gb = (uint64_t)agl[e] * m + b;
        if (cand_h[e] == q[gb]) atomicMin((unsigned long long *)&sigpass[gb], (unsigned long long)akey[e]);
    }
}

static int run_prob_sorted(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, uint64_t seq_bytes, const uint64_t *rec_start, const uint64_t *rec_len,
                           uint64_t n_rec, const uint64_t *genome_rec_off, uint64_t n_genomes, void *sig_out);

// Genomes that will be redone anyway must not keep the pass loop alive (round 6, found by tools/prob_fuzz.py): a flagged genome can be left with an EMPTY slot - its largest slot
// minimum is then +inf, no element of it ever falls out of "w^-1 (pass - 1) <= max q", and the loop over passes >= 2 never ended. Their multiplicity bound is zeroed on
// the device (k_prob_fold skips a genome with wmax == 0) and the count of active genomes recomputed from the survivors.
static int prob_retire_flagged(gs_ctx *c, const std::vector<uint8_t> &redo, uint32_t ng, uint32_t *wmax_dev, const double *qmax_dev, uint32_t &na)
{
    bool any = false;
    for (uint32_t i = 0; i < ng; i++) any |= redo[i] != 0;
    if (!any) return GS_OK;
    std::vector<uint32_t> hw(ng); std::vector<double> hq(ng);
    GS_HIP_CHECK(hipMemcpyAsync(hw.data(), wmax_dev, 4 * (size_t)ng, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(hq.data(), qmax_dev, 8 * (size_t)ng, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    na = 0;
    for (uint32_t i = 0; i < ng; i++) {
        if (redo[i]) hw[i] = 0;
        else if (hw[i] > 0 && !((1.0 / (double)hw[i]) * 1.0 > hq[i])) na++;       // (k_prob_fold's rule after pass 1)
    }
    GS_HIP_CHECK(hipMemcpyAsync(wmax_dev, hw.data(), 4 * (size_t)ng, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

// one chunk of genomes [g0, g0 + ng) through the bucketed form; hk = k-mers per genome (host). *redo (ng flags, host) marks genomes that
// must be redone by the sorted form; returns GS_OK with every flag set when a list overflowed.
static int run_prob_buckets(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, const uint64_t *rec_start, const uint64_t *rec_len, const uint64_t *upre,
                            const uint64_t *genome_rec_off, const uint64_t *gunits, uint64_t g0, uint32_t ng, const uint64_t *hk, const ProbConst &pc, void *sig_rows,
                            std::vector<uint8_t> &redo)
{
    const uint32_t m = p->sketch_size, k = p->k;
    const bool aa = p->data_t == GS_DATA_AA;
    const int sigbits = gs_value_bits(p);
    const uint64_t zone = uint_zone(m);
    int rc;
    // host plan: buckets per genome, flat bucket offsets, value offsets
    const uint32_t vbits = aa ? 5 * k : 2 * k;
    std::vector<uint32_t> sh(ng), boff(ng + 1); std::vector<uint64_t> vbase(ng);
    uint64_t T = 0, maxk = 0; uint32_t nbmax = 1, nbt = 0, shmax = 0;
    for (uint32_t i = 0; i < ng; i++) {
        uint32_t lg = 0; while (((uint64_t)PB_AVG << lg) < hk[i] && lg < vbits) lg++;
        sh[i] = vbits - lg; shmax = std::max(shmax, sh[i]); boff[i] = nbt; nbt += 1u << lg; nbmax = std::max(nbmax, 1u << lg);
        vbase[i] = T; T += hk[i]; maxk = std::max(maxk, hk[i]);
    }
    const bool id32 = shmax <= 31 && !getenv("GS_PROB_ID64");       // every genome's in-bucket id fits 4 bytes (with ~0 left over for "empty")
    boff[ng] = nbt;
    // two-level partition: coarse buckets = the top half of the bucket bits
    std::vector<uint32_t> cinfo(2 * (size_t)ng + 1);                // [0, ng): coarse shifts, [ng, 2 ng]: flat coarse-bucket offsets
    uint32_t nct = 0, ncmax = 1, nfmax = 1;
    for (uint32_t i = 0; i < ng; i++) {
        const uint32_t lg = vbits - sh[i], lgc = lg / 2;
        cinfo[i] = sh[i] + (lg - lgc); cinfo[ng + i] = nct; nct += 1u << lgc; ncmax = std::max(ncmax, 1u << lgc); nfmax = std::max(nfmax, 1u << (lg - lgc));
    }
    cinfo[2 * (size_t)ng] = nct;
    const uint64_t avg_units = maxk / 32 + 1;
    const uint32_t parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(avg_units / ((uint64_t)PBK_T * PBK_WPL) + 1, std::max<uint64_t>(1, (2 * (uint64_t)c->n_cu + ng - 1) / ng)));
    PoolBuf dsh(c, 0), dboff(c, 1), dvb(c, 2), hist(c, 3), bst(c, 4), bsz(c, 5), bgn(c, 6), vals(c, 8), q(c, 9), qprev(c, 10), sig(c, 11), sigpass(c, 12), thr(c, 13), wmax(c, 14),
        qmax(c, 15), ctr(c, 7);
    PoolBuf cv(c, 16), chh(c, 17), cgb(c, 18), akey(c, 19), agl(c, 24), acnt(c, 25), astate(c, 26), ph(c, 27), pb(c, 37), ovf(c, 38);
    const uint32_t cand_cap = (uint32_t)std::min<uint64_t>((uint64_t)ng * m * 16 + 65536, (uint64_t)1 << 30), ovf_cap = cand_cap / 4, act_cap = 1u << 24;
    PoolBuf segn(c, 28), tmpv(c, 61), dcin(c, 62);
    bool two_level = !getenv("GS_PROB_ONELEVEL") && (uint64_t)parts * ncmax <= PB_CCMAX && ncmax <= 256 && nfmax <= 256;
    // the second copy of the values is the price of the two levels: a device that has no room for it (an index with its pair cache beside the
    // sketcher, say) partitions in one level as in round 3
    if (two_level && tmpv.alloc(8 * (size_t)T + 64) != GS_OK) { (void)hipGetLastError(); two_level = false; }
    uint32_t *d_shc = nullptr, *d_coff = nullptr, *d_ccur = nullptr, *d_ccnt = nullptr;
    if (two_level) {
        if ((rc = dcin.alloc(4 * (cinfo.size() + 2 * (size_t)nct * parts) + 64))) return rc;
        d_shc = dcin.as<uint32_t>(); d_coff = d_shc + ng; d_ccur = d_coff + ng + 1; d_ccnt = d_ccur + (size_t)nct * parts;
        GS_HIP_CHECK(hipMemcpyAsync(dcin.p, cinfo.data(), 4 * cinfo.size(), hipMemcpyHostToDevice, c->stream));
    }
    if ((rc = dsh.alloc(4 * (size_t)ng)) || (rc = dboff.alloc(4 * (size_t)(ng + 1))) || (rc = dvb.alloc(8 * (size_t)ng)) || (rc = hist.alloc((size_t)4 * nbt * parts)) ||
        (rc = bst.alloc((size_t)4 * nbt)) || (rc = bsz.alloc((size_t)4 * nbt)) || (rc = bgn.alloc((size_t)4 * nbt)) || (rc = vals.alloc(8 * (size_t)T + 64)) ||
        (rc = q.alloc((size_t)8 * ng * m)) || (rc = qprev.alloc((size_t)8 * ng * m)) || (rc = sig.alloc((size_t)8 * ng * m)) || (rc = sigpass.alloc((size_t)8 * ng * m)) ||
        (rc = thr.alloc(8 * (size_t)ng)) || (rc = wmax.alloc(4 * (size_t)ng)) || (rc = qmax.alloc(8 * (size_t)ng)) || (rc = ctr.alloc(64)) ||
        (rc = cv.alloc((size_t)8 * (cand_cap + ovf_cap))) || (rc = chh.alloc((size_t)8 * (cand_cap + ovf_cap))) || (rc = cgb.alloc((size_t)8 * (cand_cap + ovf_cap))) ||
        (rc = ovf.alloc(4 * (size_t)ng)) || (rc = segn.alloc((size_t)4 * c->n_cu * 8)))
        return rc;
    GS_HIP_CHECK(hipMemcpyAsync(dsh.p, sh.data(), 4 * (size_t)ng, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dboff.p, boff.data(), 4 * (size_t)(ng + 1), hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dvb.p, vbase.data(), 8 * (size_t)ng, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_prob_init, dim3(c->n_cu * 4), dim3(256), 0, c->stream, q.as<uint64_t>(), qprev.as<uint64_t>(), sig.as<uint64_t>(), sigpass.as<uint64_t>(), ng * (uint64_t)m);
    {
        std::vector<uint32_t> ones(ng, 1u);                        // every genome of this chunk has k-mers (>= 64 per slot)
        GS_HIP_CHECK(hipMemcpyAsync(wmax.p, ones.data(), 4 * (size_t)ng, hipMemcpyHostToDevice, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    }
    GS_HIP_CHECK(hipMemsetAsync(ovf.p, 0, 4 * (size_t)ng, c->stream));
    GS_HIP_CHECK(hipMemsetAsync(ctr.p, 0, 64, c->stream));          // [0..1] work counter, [2] n_cand, [3] n_act, [4] n_active genomes
    uint32_t *ctr32 = ctr.as<uint32_t>();
    size_t lds = (size_t)4 * nbmax;
    if (getenv("GS_PROB_LDS_PAD")) lds = std::max<size_t>(lds, (size_t)atoi(getenv("GS_PROB_LDS_PAD")) * 1024);     // experiment: fewer resident scatter workgroups
    {
        ProfScope ps(c, FAM_SKETCH);
        dim3 grid(parts, ng), block(PBK_T);
#define GS_LAUNCH_PBP(AAV, MODE)                                                                                              \
    do {                                                                                                                      \
        auto kern = k_prob_partition<AAV, MODE>;                                                                              \
        if (lds > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, block, lds, c->stream, seq, rec_start, rec_len, upre, genome_rec_off, gunits, g0, kq_of(p), vbits, dsh.as<uint32_t>(), dboff.as<uint32_t>(), parts, \
                           hist.as<uint32_t>(), q.as<uint64_t>(), m, zone, pc, vals.as<uint64_t>(), dvb.as<uint64_t>());      \
    } while (0)
        if (aa) GS_LAUNCH_PBP(true, 0); else GS_LAUNCH_PBP(false, 0);
        hipLaunchKernelGGL(k_prob_scan, dim3(ng), dim3(1024), 0, c->stream, vbits, dsh.as<uint32_t>(), dboff.as<uint32_t>(), parts, hist.as<uint32_t>(), bst.as<uint32_t>(), bsz.as<uint32_t>(),
                           bgn.as<uint32_t>(), q.as<uint64_t>(), m, thr.as<uint64_t>(), d_shc, d_coff, d_ccur, d_ccnt);
        if (!two_level) { if (aa) GS_LAUNCH_PBP(true, 1); else GS_LAUNCH_PBP(false, 1); }
        else {
            // coarse scatter: the same kernel with the coarse shifts, offsets and per-part bases, into the intermediate copy; then the refinement
            const size_t lds_c = (size_t)4 * ncmax;
#define GS_LAUNCH_PBC(AAV)                                                                                                   \
    hipLaunchKernelGGL((k_prob_partition<AAV, 1>), grid, block, lds_c, c->stream, seq, rec_start, rec_len, upre, genome_rec_off, gunits, g0, kq_of(p), vbits, d_shc, d_coff, parts, \
                       d_ccur, q.as<uint64_t>(), m, zone, pc, tmpv.as<uint64_t>(), dvb.as<uint64_t>())
            if (aa) GS_LAUNCH_PBC(true); else GS_LAUNCH_PBC(false);
#undef GS_LAUNCH_PBC
            hipLaunchKernelGGL(k_prob_refine, dim3(parts * ncmax, ng), dim3(PBR_T), 0, c->stream, tmpv.as<uint64_t>(), vals.as<uint64_t>(), dvb.as<uint64_t>(), vbits, dsh.as<uint32_t>(),
                               dboff.as<uint32_t>(), d_shc, d_coff, parts, hist.as<uint32_t>(), d_ccur, d_ccnt, (uint32_t)id32);
        }
#undef GS_LAUNCH_PBP
        GS_HIP_CHECK(hipGetLastError());
        if ((rc = akey.alloc((size_t)8 * act_cap)) || (rc = agl.alloc((size_t)4 * act_cap)) || (rc = acnt.alloc((size_t)4 * act_cap)) || (rc = astate.alloc((size_t)32 * act_cap))) return rc;
        PbLists L{cv.as<uint64_t>(), chh.as<uint64_t>(), cgb.as<uint64_t>(), cand_cap, ovf_cap, ctr32 + 2, segn.as<uint32_t>(), akey.as<uint64_t>(), agl.as<uint32_t>(), acnt.as<uint32_t>(),
                  astate.as<uint64_t>(), act_cap, ctr32 + 3, nullptr};
        DevBuf profbuf;
        if (getenv("GS_PROB_PROFILE")) { if ((rc = profbuf.alloc(128))) return rc; GS_HIP_CHECK(hipMemsetAsync(profbuf.p, 0, 128, c->stream)); L.prof = profbuf.as<unsigned long long>(); }
        int per_cu = 3;
        if (id32) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_prob_buckets<uint32_t>, PB2_T, 0);
        else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_prob_buckets<uint64_t>, PB2_T, 0);
        uint32_t lg_max = 0; while ((1u << lg_max) < nbmax) lg_max++;
        const uint32_t wgs = (uint32_t)std::min<uint64_t>((uint64_t)ng << lg_max, (uint64_t)c->n_cu * std::min(std::max(per_cu, 1), 8));     // resident workgroups only: the items are dealt statically
#define GS_LAUNCH_PBB(KT)                                                                                                     \
        hipLaunchKernelGGL(k_prob_buckets<KT>, dim3(wgs), dim3(PB2_T), 0, c->stream, vals.as<uint64_t>(), dvb.as<uint64_t>(), bst.as<uint32_t>(), bsz.as<uint32_t>(), vbits, \
                           dsh.as<uint32_t>(), dboff.as<uint32_t>(), ng, lg_max, m, zone, pc, q.as<uint64_t>(), thr.as<uint64_t>(), wmax.as<uint32_t>(), L, ovf.as<uint32_t>(), \
                           (uint32_t)(two_level && id32))
        if (id32) GS_LAUNCH_PBB(uint32_t); else GS_LAUNCH_PBB(uint64_t);
#undef GS_LAUNCH_PBB
        if (L.prof) {
            unsigned long long h[16];
            GS_HIP_CHECK(hipMemcpyAsync(h, L.prof, 128, hipMemcpyDeviceToHost, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            const double it = (double)std::max<unsigned long long>(h[7], 1);
            fprintf(stderr, "[GS_PROB_PROFILE] workgroup 0 of %u (%d per CU): %llu buckets, keys/bucket %.0f, queued %.0f, candidates %.1f | cycles per bucket: zero+thr %.0f, insert %.0f, cheap test %.0f, full points %.0f, flush %.0f\n",
                    wgs, per_cu, h[7], h[8] / it, h[6] / it, h[9] / it, h[0] / it, h[1] / it, h[2] / it, h[3] / it, h[4] / it);
        }
        hipLaunchKernelGGL(k_prob_claim_list, dim3(wgs + 1), dim3(256), 0, c->stream, cv.as<uint64_t>(), chh.as<uint64_t>(), cgb.as<uint64_t>(), segn.as<uint32_t>(), cand_cap / wgs, wgs,
                           cand_cap, ctr32 + 2, ovf_cap, q.as<uint64_t>(), sigpass.as<uint64_t>());
        hipLaunchKernelGGL(k_prob_fold, dim3(ng), dim3(256), 0, c->stream, m, 1u, q.as<uint64_t>(), qprev.as<uint64_t>(), sig.as<uint64_t>(), sigpass.as<uint64_t>(), wmax.as<uint32_t>(),
                           qmax.as<double>(), ctr32 + 4);
        GS_HIP_CHECK(hipGetLastError());
    }
    uint32_t hc[8]; std::vector<uint32_t> hovf(ng);
    GS_HIP_CHECK(hipMemcpyAsync(hc, ctr.p, 32, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(hovf.data(), ovf.p, 4 * (size_t)ng, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));               // the one host round trip of a chunk whose genomes finish in pass 1
    redo.assign(ng, 0);
    if (hc[2] > ovf_cap || hc[3] > act_cap) { redo.assign(ng, 1); return GS_OK; }      // a list overflowed: the whole chunk goes the sorted way
    for (uint32_t i = 0; i < ng; i++) redo[i] = hovf[i] ? 1 : 0;
    uint32_t na = hc[4];
    if ((rc = prob_retire_flagged(c, redo, ng, wmax.as<uint32_t>(), qmax.as<double>(), na))) return rc;
    const uint32_t n_list = hc[3];
    if (na && n_list) {
        if ((rc = ph.alloc((size_t)8 * n_list)) || (rc = pb.alloc((size_t)4 * n_list))) return rc;
        const uint32_t lg = std::max<uint32_t>(1, std::min<uint32_t>((n_list + 255) / 256, (uint32_t)c->n_cu * 16));
        for (uint32_t it = 2; na; it++) {
            GS_HIP_CHECK(hipMemsetAsync(ctr32 + 4, 0, 4, c->stream));
            hipLaunchKernelGGL(k_prob_point_act, dim3(lg), dim3(256), 0, c->stream, akey.as<uint64_t>(), agl.as<uint32_t>(), acnt.as<uint32_t>(), n_list, act_cap, m, zone, pc, it,
                               qmax.as<double>(), q.as<uint64_t>(), astate.as<uint64_t>(), ph.as<uint64_t>(), pb.as<uint32_t>());
            hipLaunchKernelGGL(k_prob_claim_act, dim3(lg), dim3(256), 0, c->stream, akey.as<uint64_t>(), agl.as<uint32_t>(), n_list, m, q.as<uint64_t>(), ph.as<uint64_t>(), pb.as<uint32_t>(),
                               sigpass.as<uint64_t>());
            hipLaunchKernelGGL(k_prob_fold, dim3(ng), dim3(256), 0, c->stream, m, it, q.as<uint64_t>(), qprev.as<uint64_t>(), sig.as<uint64_t>(), sigpass.as<uint64_t>(), wmax.as<uint32_t>(),
                               qmax.as<double>(), ctr32 + 4);
            GS_HIP_CHECK(hipGetLastError());
            GS_HIP_CHECK(hipMemcpyAsync(&na, ctr32 + 4, 4, hipMemcpyDeviceToHost, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (getenv("GS_PROB_VERBOSE") && (it < 8 || (it & (it - 1)) == 0)) fprintf(stderr, "[GS_PROB] pass %u done: %u genomes still active, %u elements on the active list\n", it, na, n_list);
        }
    }
    if (sigbits == 32) hipLaunchKernelGGL(k_prob_write<uint32_t>, dim3(c->n_cu * 4), dim3(256), 0, c->stream, q.as<uint64_t>(), sig.as<uint64_t>(), ng * (uint64_t)m, (uint32_t *)sig_rows);
    else hipLaunchKernelGGL(k_prob_write<uint64_t>, dim3(c->n_cu * 4), dim3(256), 0, c->stream, q.as<uint64_t>(), sig.as<uint64_t>(), ng * (uint64_t)m, (uint64_t *)sig_rows);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// =====================================================================================================
// prob, TIERED form (round 6; DESIGN.md 3.1 "ProbMinHash3a, tiers"). Two observations carry it:
//  (1) every copy of a k-mer v draws from the same RNG(H(v)), so its first truncated-exponential draw x1 is a property of the VALUE, and its smallest point
//      is h1 = x1 / w. With thr >= max_b q[b] (final), v can only matter when x1 <= w thr: a k-mer whose x1 lies in [T thr, (T + 1) thr) needs a multiplicity
//      above T to matter at all - at thr ~ 0.04 (5 Mbp, s = 18000) 96 % of the k-mers need w >= 2, 85 % need w >= 5.
//  (2) an UPPER bound c >= w is enough to drop such a k-mer (1 / c and the product are monotone in c, in IEEE arithmetic too), and a count-min cell - the sum
//      of the multiplicities of everything that shares the cell - is one: one non-returning 16-bit LDS add per k-mer instead of a CAS insert with probing.
// So a bucket is counted twice: pass A adds every k-mer to its count-min cell; pass B reads the cell, draws x1 and drops the k-mer when even c copies could
// not bring its first point under thr (nor keep it alive for pass 2: 1 / c > thr). What is not dropped - the ~thr fraction that matters as singletons, the
// real repeats, and the false alarms of shared cells - enters the exact LDS hash table (CAS + duplicate count, as in k_prob_buckets: ALL copies of a value
// see the same cell, so they enter or stay out together and the multiplicity is exact), and the table is swept by the cheap test / full generator of the
// bucketed form. thr starts at a SPECULATIVE cap (m / N)(ln m + c) per genome (total k-mers N: the point process has rate sum w = N whatever the repeats) and
// is verified afterwards: max_b q[b] <= cap proves that nothing dropped could have been a slot minimum (a dropped point lies above the cap, hence above a
// point that stayed in its slot); a genome that fails the check, overflows a slice or a table is redone by the bucketed form (exact fallback).
// The partition is ONE pass without a count pass: a workgroup walks its tiles of the genome twice (count, place), counting-sorts each tile of <= 32 768
// k-mers by bucket in LDS and appends run by run to its PRIVATE slice of every bucket (fixed capacity: mean + 5 sigma; an overflow flags the genome) - 4 bytes
// per k-mer cross HBM twice (the in-bucket id of a bijection of the values, pt_bucket / pt_value) where the two-level partition moved 8 + 8 + 4 + 4.
// =====================================================================================================
constexpr int PT_T = 1024;            // lanes of the partition kernel = units (32 symbols) per tile
constexpr int PT_LGMAX = 11;          // buckets per genome <= 2048 (three LDS arrays of NB words beside the 128 kB tile)
constexpr int PT_AVG = 4096;          // k-mers per bucket aimed at (2048 .. 4096)
constexpr int PT_MINB = 256;          // fewer k-mers per bucket than this: the bucketed / sorted forms
constexpr int PT2_T = 512;            // lanes of the bucket kernel
constexpr int PT_CM = 8192;           // count-min cells per bucket (16 bits each)
constexpr int PT_Q = 1536;            // ids queued for the exact table per bucket
// The tiered form's bijection of the vbits-bit values is cheaper than pb_hash (one 32-bit multiplication instead of a 64-bit one - the bucket kernel undoes it
// once per k-mer and is bound by exactly these quarter-rate multiplications): id = the low sh bits of the value as they are, bucket = its top lg bits XOR a
// lg-bit hash of the id. Given (bucket, id) the top bits come back by the same XOR. Buckets are as even as the hash of the low 31 bits; whatever indexes a
// table by the id hashes it first (the raw low bits of a k-mer are its last bases).
__device__ __forceinline__ uint32_t pt_mix(uint32_t id, uint32_t lg) { return lg ? (id * 0x9E3779B1u) >> (32 - lg) : 0u; }
__device__ __forceinline__ uint32_t pt_bucket(uint64_t v, uint32_t sh, uint32_t lg, uint32_t idmask) { return (uint32_t)(v >> sh) ^ pt_mix((uint32_t)v & idmask, lg); }
__device__ __forceinline__ uint64_t pt_value(uint32_t bk, uint32_t id, uint32_t sh, uint32_t lg) { return ((uint64_t)(bk ^ pt_mix(id, lg)) << sh) | (uint64_t)id; }
struct PtCountEmit {
    uint32_t *cnt; uint32_t sh, lg, idmask;
    __device__ __forceinline__ void operator()(uint64_t v, uint64_t, uint64_t) const { atomicAdd(&cnt[pt_bucket(v, sh, lg, idmask)], 1u); }
};
struct PtPlaceEmit {
    uint32_t *pos, *ids; uint32_t sh, lg, idmask;
    __device__ __forceinline__ void operator()(uint64_t v, uint64_t, uint64_t) const { ids[atomicAdd(&pos[pt_bucket(v, sh, lg, idmask)], 1u)] = (uint32_t)v & idmask; }
};
// grid (parts, genomes of the chunk). vals32[g_vbase[gl] + (b * parts + part) * g_cap[gl] + i] = i-th id this part found for bucket b; cnt[(g_boff[gl] + b) * parts + part] = how many.
template <bool AA>
__global__ __launch_bounds__(PT_T) void k_prob_part1(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len,
                                                     const uint64_t *__restrict__ rec_upre, const uint64_t *__restrict__ genome_rec_off, const uint64_t *__restrict__ gen_units,
                                                     uint64_t g0, uint32_t kq, uint32_t vbits, const uint32_t *__restrict__ g_sh, const uint32_t *__restrict__ g_boff,
                                                     const uint64_t *__restrict__ g_vbase, const uint32_t *__restrict__ g_cap, uint32_t parts, uint32_t *__restrict__ vals32,
                                                     uint32_t *__restrict__ cnt, uint32_t *__restrict__ ovf)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pt[];
    __shared__ uint32_t s_w[16];
    const uint32_t gl = blockIdx.y, part = blockIdx.x;
    const uint64_t g = g0 + gl;
    const uint32_t sh = g_sh[gl], NB = 1u << (vbits - sh), cap = g_cap[gl];
    uint32_t *s_cnt = s_pt, *s_start = s_pt + NB, *s_cur = s_pt + 2 * NB, *s_ids = s_pt + 3 * NB;       // s_cnt doubles as the placement cursor of walk B
    const uint64_t r0 = genome_rec_off[g], r1 = genome_rec_off[g + 1], units = gen_units[g];
    const uint32_t k = kq_k(kq);
    const uint64_t mask = kmer_mask(AA, k), rc_or = kq_rc_or(kq);
    const uint32_t idmask = (uint32_t)(((uint64_t)1 << sh) - 1), lg = vbits - sh;      // sh <= 31
    const uint32_t rcshift = 2 * (k - 1);
    const uint64_t tiles = (units + PT_T - 1) / PT_T, t_lo = tiles * part / parts, t_hi = tiles * (part + 1) / parts;
    uint32_t *out = vals32 + g_vbase[gl];
    for (uint32_t b = threadIdx.x; b < NB; b += PT_T) s_cur[b] = 0;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t CH = NB > (uint32_t)PT_T ? NB / PT_T : 1u;     // consecutive buckets per lane in the prefix
    bool over = false;
    for (uint64_t t = t_lo; t < t_hi; t++) {
        for (uint32_t b = threadIdx.x; b < NB; b += PT_T) s_cnt[b] = 0;
        __syncthreads();
        const uint64_t f = t * PT_T + threadIdx.x;
        if (f < units) { PtCountEmit e{s_cnt, sh, lg, idmask}; walk_unit<AA>(seq, rec_start, rec_len, rec_upre, r0, r1, f, k, mask, rcshift, rc_or, e); }
        __syncthreads();
        // exclusive prefix of the bucket counts -> s_start, and the cursors of walk B
        uint32_t loc = 0;
        for (uint32_t c = 0; c < CH; c++) { const uint32_t b = threadIdx.x * CH + c; if (b < NB) loc += s_cnt[b]; }
        uint32_t inc = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        uint32_t run = inc - loc;
        for (uint32_t w = 0; w < wv; w++) run += s_w[w];
        for (uint32_t c = 0; c < CH; c++) { const uint32_t b = threadIdx.x * CH + c; if (b < NB) { const uint32_t x = s_cnt[b]; s_start[b] = run; s_cnt[b] = run; run += x; } }
        __syncthreads();
        if (f < units) { PtPlaceEmit e{s_cnt, s_ids, sh, lg, idmask}; walk_unit<AA>(seq, rec_start, rec_len, rec_upre, r0, r1, f, k, mask, rcshift, rc_or, e); }
        __syncthreads();
        // run by run to this part's slices: half a wavefront per bucket (a run is ~16-32 ids)
        const uint32_t hw = threadIdx.x >> 5, hl = threadIdx.x & 31;
        for (uint32_t b = hw; b < NB; b += PT_T / 32) {
            const uint32_t a = s_start[b], e = s_cnt[b], cur = s_cur[b], nb = e - a;
            if (nb == 0) continue;
            if (cur + nb > cap) { over = true; continue; }
            uint32_t *dst = out + ((uint64_t)b * parts + part) * cap + cur;
            for (uint32_t i = hl; i < nb; i += 32) dst[i] = s_ids[a + i];
            if (hl == 0) s_cur[b] = cur + nb;
        }
        __syncthreads();
    }
    if (over) ovf[gl] = 1;
    uint32_t *cg = cnt + (uint64_t)g_boff[gl] * parts;
    for (uint32_t b = threadIdx.x; b < NB; b += PT_T) cg[(uint64_t)b * parts + part] = s_cur[b];
}

// DNA form of k_prob_part1 with ONE walk per tile (the two-walk form above stays for amino acids): a lane keeps the 32 k-mers of its unit in registers - the id
// and (bucket, rank), the rank being what the counting atomic returns - so that after the prefix over the bucket counts each id goes straight to
// s_ids[start[bucket] + rank]: no second walk (the walk is ~half of the kernel's instructions), no second atomic.
template <bool CHECK>
__device__ __forceinline__ void pt_walk_dna(uint64_t w, uint64_t fwd, uint64_t rc, uint64_t mask, uint32_t rcshift, uint64_t rc_or, uint32_t jlo, uint32_t jhi, uint32_t sh, uint32_t lg,
                                            uint32_t idmask, uint32_t *s_cnt, uint32_t (&idr)[32], uint32_t (&pkr)[32])
{
#pragma unroll
    for (uint32_t j = 0; j < 32; j++) {
        const uint64_t c = w >> 62; w <<= 2;
        fwd = ((fwd << 2) | c) & mask;
        rc = (rc >> 2) | ((3 - c) << rcshift) | rc_or;
        const uint64_t v = fwd < rc ? fwd : rc;
        const uint32_t id = (uint32_t)v & idmask, b = (uint32_t)(v >> sh) ^ pt_mix(id, lg);
        idr[j] = id;
        if (!CHECK || (j >= jlo && j < jhi)) pkr[j] = (b << 16) | atomicAdd(&s_cnt[b], 1u);      // rank < 32 768: 15 bits
        else pkr[j] = 0xFFFFFFFFu;
    }
}
// inclusive maximum over the lanes 0 .. l of a wavefront (DPP: shifts inside the rows of 16, then the row totals broadcast to the rows behind)
__device__ __forceinline__ uint32_t wave_incl_max_scan(uint32_t v)
{
#define GS_DPP_MAX(ctrl, rows) do { const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, false); v = y > v ? y : v; } while (0)
    GS_DPP_MAX(0x111, 0xF); GS_DPP_MAX(0x112, 0xF); GS_DPP_MAX(0x114, 0xF); GS_DPP_MAX(0x118, 0xF);      // row_shr:1 / 2 / 4 / 8
    GS_DPP_MAX(0x142, 0xA);                                                                            // row_bcast:15 -> rows 1 and 3
    GS_DPP_MAX(0x143, 0xC);                                                                            // row_bcast:31 -> rows 2 and 3
#undef GS_DPP_MAX
    return v;
}
__global__ __launch_bounds__(PT_T) void k_prob_part1_dna(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start, const uint64_t *__restrict__ rec_len,
                                                         const uint64_t *__restrict__ rec_upre, const uint64_t *__restrict__ genome_rec_off, const uint64_t *__restrict__ gen_units,
                                                         uint64_t g0, uint32_t kq, uint32_t vbits, const uint32_t *__restrict__ g_sh, const uint32_t *__restrict__ g_boff,
                                                         const uint64_t *__restrict__ g_vbase, const uint32_t *__restrict__ g_cap, uint32_t parts, uint32_t *__restrict__ vals32,
                                                         uint32_t *__restrict__ cnt, uint32_t *__restrict__ ovf)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_pt[];
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_mark[PT_T];                             // a 64-word mark row per wavefront (the copy-out below)
    const uint32_t gl = blockIdx.y, part = blockIdx.x;
    const uint64_t g = g0 + gl;
    const uint32_t sh = g_sh[gl], NB = 1u << (vbits - sh), cap = g_cap[gl];
    uint32_t *s_cnt = s_pt, *s_start = s_pt + NB, *s_cur = s_pt + 2 * NB + 1, *s_ids = s_pt + 3 * NB + 4;      // s_start has NB + 1 entries (the total closes the last bucket)
    const uint64_t r0 = genome_rec_off[g], r1 = genome_rec_off[g + 1], units = gen_units[g];
    const uint32_t k = kq_k(kq);
    const uint64_t mask = kmer_mask(false, k), rc_or = kq_rc_or(kq);
    const uint32_t idmask = (uint32_t)(((uint64_t)1 << sh) - 1), lg = vbits - sh;
    const uint32_t rcshift = 2 * (k - 1);
    const uint64_t tiles = (units + PT_T - 1) / PT_T, t_lo = tiles * part / parts, t_hi = tiles * (part + 1) / parts;
    uint32_t *out = vals32 + g_vbase[gl];
    const uint64_t *w64 = (const uint64_t *)seq;
    for (uint32_t b = threadIdx.x; b < NB; b += PT_T) s_cur[b] = 0;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t CH = NB > (uint32_t)PT_T ? NB / PT_T : 1u;
    bool over = false;
    // a lane's unit of a tile: the packed word, the word in front of it (the k - 1 bases before the unit) and which of its 32 windows are k-mers of its record.
    // The unit of tile t + 1 is fetched while tile t is worked on: with one workgroup per CU nothing else hides the two dependent round trips to HBM.
    struct Unit { uint64_t w, pw; uint32_t jlo, jhi; bool init; };
    auto fetch = [&](uint64_t t) -> Unit {
        Unit un{0, 0, 0, 0, false};
        const uint64_t f = t * PT_T + threadIdx.x;
        if (t < t_hi && f < units) {
            uint64_t lo = r0, hi = r1;
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (rec_upre[mid] <= f) lo = mid; else hi = mid; }
            const uint64_t rb = rec_start[lo], re = rb + rec_len[lo];
            const uint64_t u = (rb >> 5) + (f - rec_upre[lo]), a0 = u << 5, first_valid = rb + k - 1;
            un.w = w64[u];
            un.init = a0 > rb && k > 1;
            if (un.init) un.pw = w64[u - 1];
            un.jlo = first_valid > a0 ? (uint32_t)std::min<uint64_t>(first_valid - a0, 32) : 0u;
            un.jhi = re > a0 ? (uint32_t)std::min<uint64_t>(re - a0, 32) : 0u;
            if (un.jhi <= un.jlo) { un.jlo = 0; un.jhi = 0; }
        }
        return un;
    };
    Unit cur = fetch(t_lo);
    for (uint64_t t = t_lo; t < t_hi; t++) {
        for (uint32_t b = threadIdx.x; b < NB; b += PT_T) s_cnt[b] = 0;
        const Unit nxt = fetch(t + 1);
        __syncthreads();
        uint32_t idr[32], pkr[32];
        {
            const uint64_t w = __builtin_bswap64(cur.w);
            uint64_t fwd = 0, rc = 0;
            if (cur.init) {
                // the state after the k - 1 bases in front of the word, in closed form (walk_unit)
                const uint64_t pw = __builtin_bswap64(cur.pw);
                const uint64_t lowm = ((uint64_t)1 << (2 * (k - 1))) - 1;
                fwd = pw & lowm;
                uint64_t br = __builtin_bitreverse64(fwd);
                br = ((br >> 1) & 0x5555555555555555ull) | ((br & 0x5555555555555555ull) << 1);
                rc = (((~(br >> (2 * (33 - k)))) & lowm) << 2) | rc_or;
            }
            const bool have = cur.jhi > cur.jlo, full = cur.jlo == 0 && cur.jhi == 32;
            if (__ballot(full) == __ballot(true)) pt_walk_dna<false>(w, fwd, rc, mask, rcshift, rc_or, 0, 32, sh, lg, idmask, s_cnt, idr, pkr);      // (wavefront-uniform)
            else if (__ballot(have)) pt_walk_dna<true>(w, fwd, rc, mask, rcshift, rc_or, cur.jlo, cur.jhi, sh, lg, idmask, s_cnt, idr, pkr);
            else {
#pragma unroll
                for (int j = 0; j < 32; j++) pkr[j] = 0xFFFFFFFFu;
            }
        }
        __syncthreads();
        uint32_t loc = 0;
        for (uint32_t c = 0; c < CH; c++) { const uint32_t b = threadIdx.x * CH + c; if (b < NB) loc += s_cnt[b]; }
        uint32_t inc = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t y = __shfl_up(inc, o); if ((int)lane >= o) inc += y; }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        uint32_t run = inc - loc;
        for (uint32_t w = 0; w < wv; w++) run += s_w[w];
        for (uint32_t c = 0; c < CH; c++) { const uint32_t b = threadIdx.x * CH + c; if (b < NB) { s_start[b] = run; run += s_cnt[b]; } }
        if (threadIdx.x == PT_T - 1) s_start[NB] = run;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 32; j++) if (pkr[j] != 0xFFFFFFFFu) s_ids[s_start[pkr[j] >> 16] + (pkr[j] & 0xFFFFu)] = idr[j];
        __syncthreads();
        // to this part's slices: a wavefront takes 64 buckets at a time - their ids are one contiguous stretch of s_ids, walked in windows of 64 positions. The
        // buckets that START inside a window leave their number at their start position (a 64-word mark row per wavefront); an inclusive max-scan over the lanes
        // (the marks increase along the window) tells every position its bucket, the bucket's destination comes from its lane by one permute. ~25 wave
        // instructions per 64 ids (a 6-step search per id over the 64 starts was ~45, a half wavefront per run a chain of LDS round trips per ~16 ids).
        {
            uint32_t *mk = s_mark + wv * 64;
            for (uint32_t b0 = wv * 64; b0 < NB; b0 += (PT_T / 64) * 64) {
                const uint32_t nbk = NB - b0 < 64u ? NB - b0 : 64u;
                const uint32_t st = lane < nbk ? s_start[b0 + lane] : 0u, en = lane < nbk ? s_start[b0 + lane + 1] : 0u, sc = lane < nbk ? s_cur[b0 + lane] : 0u;
                const uint32_t nb = en - st;
                const bool fits = sc + nb <= cap;
                if (nb && !fits) over = true;
                const uint64_t okb = __ballot(fits);
                const uint32_t dbase = (uint32_t)(((uint64_t)(b0 + lane) * parts + part) * cap) + sc - st;      // destination of position pos = dbase + pos (mod 2^32)
                const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)st), p1 = s_start[b0 + nbk];
                const uint64_t nz = __ballot(nb != 0);
                uint32_t carry = nz ? (uint32_t)__builtin_ctzll(nz) : 0u;                                      // the bucket (lane) that holds position p0
                for (uint32_t pw = p0; pw < p1; pw += 64) {
                    mk[lane] = 0;
                    __builtin_amdgcn_wave_barrier();
                    if (nb && st >= pw && st - pw < 64u) mk[st - pw] = lane + 1;
                    __builtin_amdgcn_wave_barrier();
                    uint32_t v = mk[lane];
                    __builtin_amdgcn_wave_barrier();
                    v = wave_incl_max_scan(v);
                    v = v ? v - 1 : carry;
                    carry = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
                    const uint32_t pos = pw + lane;
                    const uint32_t d = (uint32_t)__shfl((int)dbase, (int)v);
                    if (pos < p1 && ((okb >> v) & 1ull)) out[d + pos] = s_ids[pos];
                }
                if (lane < nbk) s_cur[b0 + lane] = fits ? sc + nb : cap;
            }
        }
        __syncthreads();
        cur = nxt;
    }
    if (over) ovf[gl] = 1;
    uint32_t *cg = cnt + (uint64_t)g_boff[gl] * parts;
    for (uint32_t b = threadIdx.x; b < NB; b += PT_T) cg[(uint64_t)b * parts + part] = s_cur[b];
}

// ---- the bucket work of the tiered form, in two kernels (one kernel with both halves ran at three workgroups per CU - 52 kB of LDS, 80 VGPRs - and 0.66 of its
//      VALU issue; the filter half is 9/10 of the instructions and needs neither the table nor the registers of the generator):
// k_prob_tier_filter  workgroup per bucket: count-min of every id (pass A), first draw of every id against the bound its cell allows (pass B); what may matter
//                     (~1 id in 12) is compacted to a list in global memory, (offset, count) per bucket in `desc`. 22 kB of LDS, <= 64 VGPRs: four workgroups per CU.
// k_prob_tier_points  one WAVEFRONT per bucket (no barriers; T = 64, table of 1024) - or a workgroup for the few buckets with more than 512 kept ids (T = 512,
//                     table of 4096, the ones the wavefront form lists in `big`): the kept ids enter the exact table, the owners of the entries run the
//                     generator with the exact multiplicity and lower q[], exactly as k_prob_buckets does from its table.
__global__ __launch_bounds__(PT2_T, 8) void k_prob_tier_filter(const uint32_t *__restrict__ vals32, const uint64_t *__restrict__ g_vbase, const uint32_t *__restrict__ g_cap,
                                                               const uint32_t *__restrict__ cnt, uint32_t parts, uint32_t vbits, const uint32_t *__restrict__ g_sh,
                                                               const uint32_t *__restrict__ g_boff, uint32_t ng, uint32_t lg_max, ProbConst pc, const uint64_t *__restrict__ thr,
                                                               uint32_t *__restrict__ kept, uint32_t kept_cap, uint32_t *__restrict__ kept_n, uint2 *__restrict__ desc,
                                                               uint32_t *__restrict__ ovf, unsigned long long *__restrict__ prof)
{
    __shared__ __attribute__((aligned(16))) uint32_t cm[PT_CM / 2];
    __shared__ uint32_t s_q[PT_Q];
    __shared__ uint32_t s_ns;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr uint32_t NW = PT2_T / 64;
    constexpr int KPL = 8;                                        // ids per lane held in registers between the two passes (more: re-read from the slices)
    const uint64_t n_items = (uint64_t)ng << lg_max;              // bucket-major over the chunk, as k_prob_buckets
    const uint32_t region = kept_cap / gridDim.x; uint32_t my_kept = 0;
    // (jpos, gl) of item = jpos ng + gl, stepped without a division per bucket (a 64-bit division is ~150 instructions: a fifth of a bucket's work)
    const uint32_t dj = gridDim.x / ng, dg = gridDim.x % ng;
    uint32_t jpos = blockIdx.x / ng, gl = blockIdx.x % ng;
    const uint64_t k_one = (uint64_t)(0x1.0p52 / pc.c1 * (1.0 - 0x1.0p-40));
    for (uint64_t item = blockIdx.x; item < n_items; item += gridDim.x, jpos += dj, gl += dg) {
        if (gl >= ng) { gl -= ng; jpos++; }
        const uint32_t sh = g_sh[gl], lg = vbits - sh, rs = lg_max - lg;
        if (jpos & ((1u << rs) - 1u)) continue;                    // (workgroup-uniform) this genome has fewer buckets: it takes part at every 2^rs-th position
        const uint32_t bk = jpos >> rs;
        const uint64_t fb = (uint64_t)g_boff[gl] + bk;
        const bool pf = prof && blockIdx.x == 0 && threadIdx.x == 0;
        long long t0 = pf ? clock64() : 0, t1;
#define GS_PSTAMP(i) do { if (pf) { t1 = clock64(); atomicAdd(&prof[i], (unsigned long long)(t1 - t0)); t0 = t1; } } while (0)
        const uint32_t capg = g_cap[gl];
        const uint32_t *base = vals32 + g_vbase[gl] + (uint64_t)bk * parts * capg;
        // ---- this wavefront's share of the bucket's slices: whole slices (parts >= 8) or an equal piece of one; the ids go to registers at once (every load
        //      in flight together: the slices are cold in HBM and a load per loop trip was a round trip per trip)
        const uint32_t mycnt = lane < parts ? cnt[fb * parts + lane] : 0u;      // (every wavefront reads the counts itself: no barrier in front of the loads)
        uint32_t n = mycnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n += (uint32_t)__shfl_xor((int)n, o);
        const uint32_t IT = capg >> 6;                             // 64-id trips per slice (the capacity is a multiple of 64)
        uint32_t lo = 0, hi = 0, p_small = 0, NU;
        if (parts >= NW) NU = (parts / NW) * IT;
        else {
            const uint32_t lgp = 31u - (uint32_t)__builtin_clz(parts), lgs = 3u - lgp, piece = wv >> lgp; p_small = wv & (parts - 1u);      // parts is a power of two below NW = 8
            const uint32_t np = (uint32_t)__shfl((int)mycnt, (int)p_small);
            lo = (np * piece) >> lgs; hi = (np * (piece + 1)) >> lgs;
            NU = (hi - lo + 63) >> 6;
        }
        auto key_at = [&](uint32_t u, uint32_t &id) -> bool {      // u-th trip of this wavefront: the lane's id, or false
            if (parts >= NW) {
                const uint32_t p = wv + NW * (u / IT), i = (u % IT) * 64 + lane;
                if (p >= parts || i >= (uint32_t)__shfl((int)mycnt, (int)p)) return false;
                id = base[(uint64_t)p * capg + i];
                return true;
            }
            const uint32_t i = lo + u * 64 + lane;
            if (i >= hi) return false;
            id = base[(uint64_t)p_small * capg + i];
            return true;
        };
        uint32_t kreg[KPL]; uint32_t kval = 0;                     // kval: bit u = kreg[u] holds an id
#pragma unroll
        for (int u = 0; u < KPL; u++) { kreg[u] = 0; if ((uint32_t)u < NU && key_at(u, kreg[u])) kval |= 1u << u; }
        // the genome's cap (q[] moves only in the second kernel), as thresholds on the 52 uniform bits K of the first draw x0 = c1 K 2^-52: x0 < 1 for K < k_one, and
        // x0 > c thr for K > c t_one - both with a margin of 2^-40 relative on the safe side (a k-mer that is kept needlessly costs time, never the result)
        const uint64_t t_one = thr[ng + gl];                       // (the host's: run_prob_tiers)
        __syncthreads();                                           // the previous bucket's LDS is dead
        for (uint32_t s = threadIdx.x; s < PT_CM / 8; s += PT2_T) ((uint4 *)cm)[s] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) s_ns = 0;
        __syncthreads();
        if (n == 0) { if (threadIdx.x == 0) desc[fb] = make_uint2(0u, 0u); continue; }      // (workgroup-uniform)
        GS_PSTAMP(0);
        auto cell_of = [&](uint32_t id) -> uint32_t { return (id * 0x9E3779B1u) >> 19; };      // 13 bits: PT_CM cells (the product pt_mix takes its lg bits from: one multiplication)
        const bool wide = n > 65535u;                              // a 16-bit cell could wrap: everything counts as "many copies" (the queue overflows: redone)
        // ---- pass A: every k-mer into its count-min cell (one non-returning LDS add)
        if (!wide) {
#pragma unroll
            for (int u = 0; u < KPL; u++) if (kval & (1u << u)) { const uint32_t c = cell_of(kreg[u]); atomicAdd(&cm[c >> 1], 1u << ((c & 1) * 16)); }
            for (uint32_t u = KPL; u < NU; u++) { uint32_t id; if (key_at(u, id)) { const uint32_t c = cell_of(id); atomicAdd(&cm[c >> 1], 1u << ((c & 1) * 16)); } }
        }
        __syncthreads();
        GS_PSTAMP(1);
        // ---- pass B: first draw of every k-mer against the bound its cell allows; what may matter is queued, compacted across the wavefront
        auto keep_b = [&](uint32_t id) -> bool {
            const uint32_t ce = cell_of(id);
            const uint32_t c = wide ? 0xFFFFu : ((cm[ce >> 1] >> ((ce & 1) * 16)) & 0xFFFFu);      // >= the multiplicity of this value
            const uint64_t v = pt_value(bk, id, sh, lg);
            const uint64_t s0 = splitmix_mix(v + GS_GAMMA), s3 = splitmix_mix(v + 4 * GS_GAMMA);
            const uint64_t K = (rotl64(s0 + s3, 23) + s0) >> 12;     // x0 = c1 K 2^-52; when < 1 it IS the first truncated exponential (SPEC 3.3)
            // even c copies leave its first point above every slot minimum (x1 / w > thr), and it is dead in pass 2 (1 / w > thr), for every w <= c: tested without
            // the division and in integers - x0 > c thr (1 + 2^-40) implies fl(fl(1 / w) x0) > thr (the margin covers the roundings), and with x0 < 1 also 1 / w > thr
            return !(c < 2048u && K < k_one && K > (uint64_t)c * t_one);
        };
        auto queue_b = [&](bool valid, uint32_t id) {
            const bool keep = valid && keep_b(id);
            const uint64_t bal = __ballot(keep);
            if (bal) {                                               // (wavefront-uniform)
                uint32_t qb = 0;
                if (lane == 0) qb = atomicAdd(&s_ns, (uint32_t)__popcll(bal));
                qb = (uint32_t)__builtin_amdgcn_readfirstlane((int)qb);
                const uint32_t at = qb + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                if (keep && at < (uint32_t)PT_Q) s_q[at] = id;
            }
        };
#pragma unroll
        for (int u = 0; u < KPL; u++) if ((uint32_t)u < NU) queue_b((kval >> u) & 1u, kreg[u]);
        for (uint32_t u = KPL; u < NU; u++) { uint32_t id = 0; const bool ok = key_at(u, id); queue_b(ok, id); }
        __syncthreads();
        GS_PSTAMP(2);
        // ---- the kept ids to the global list (one returning atomic per bucket reserves their place)
        // ---- the kept ids to this workgroup's own stretch of the global list (one cursor for all workgroups was a same-address returning atomic per bucket:
        //      13 ns each, serialised - the whole kernel's time)
        const uint32_t nk = s_ns;
        uint32_t off = 0xFFFFFFFFu;
        if (nk <= (uint32_t)PT_Q && my_kept + nk <= region) { off = blockIdx.x * region + my_kept; my_kept += nk; }
        if (threadIdx.x == 0) {
            if (off == 0xFFFFFFFFu) ovf[gl] = 1;                    // more than the queue holds (a loose cap over a large bucket) or the stretch is full: the genome is redone
            desc[fb] = make_uint2(off, off == 0xFFFFFFFFu ? 0u : nk);
        }
        if (off != 0xFFFFFFFFu) for (uint32_t i = threadIdx.x; i < nk; i += PT2_T) kept[off + i] = s_q[i];
        if (pf) { atomicAdd(&prof[6], (unsigned long long)nk); atomicAdd(&prof[7], 1ull); atomicAdd(&prof[8], (unsigned long long)n); }
        GS_PSTAMP(3);
#undef GS_PSTAMP
    }
    if (threadIdx.x == 0 && my_kept) atomicAdd(kept_n, my_kept);      // (statistics)
}
// T lanes per bucket (64: a wavefront, no other wavefront shares its LDS; 512: a workgroup), TAB table entries. BIG: the items come from the list `big`.
template <int T, int TAB, bool BIG>
__global__ __launch_bounds__(T) void k_prob_tier_points(const uint32_t *__restrict__ kept, const uint2 *__restrict__ desc, uint32_t vbits, const uint32_t *__restrict__ g_sh,
                                                        const uint32_t *__restrict__ g_boff, uint32_t ng, uint32_t lg_max, uint32_t m, uint64_t zone, ProbConst pc,
                                                        uint64_t *__restrict__ q, uint64_t *__restrict__ thr, uint32_t *__restrict__ wmax, PbLists L, uint32_t *__restrict__ ovf,
                                                        uint32_t *__restrict__ big, uint32_t *__restrict__ n_big, uint32_t big_cap)
{
    constexpr int NT = T == 64 ? 8 : (PT_Q + T - 1) / T;        // trips over the kept ids: <= 512 for a wavefront, <= PT_Q for a workgroup
    __shared__ __attribute__((aligned(16))) uint32_t tab[TAB];
    __shared__ __attribute__((aligned(16))) uint32_t dup[TAB / 2];
    __shared__ uint32_t s_nc; __shared__ unsigned long long s_mx;
    const uint32_t EMPTY = 0xFFFFFFFFu;
    const uint32_t seg = BIG ? 0u : L.cand_cap / gridDim.x;      // possible winners go straight to this block's segment of the candidate list (BIG: the shared region behind
    uint32_t my_nc = 0;                                           // the segments - they belong to the wavefront form's blocks)
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t n_items = BIG ? (uint64_t)min(*n_big, big_cap) : ((uint64_t)ng << lg_max);
    // A bucket's description and its kept ids are fetched ONE ITERATION AHEAD (a wavefront works alone on its bucket: nothing else hides the two dependent
    // round trips - description, then ids - in front of the table insert).
    struct Item { uint32_t jpos, gl, bk, sh, nk; uint32_t id[NT]; };
    auto fetch = [&](uint64_t it0, Item &x) {
        x.nk = 0; x.jpos = 0; x.gl = 0; x.bk = 0; x.sh = 1;
        if (it0 >= n_items) return;
        const uint32_t item = BIG ? big[it0] : (uint32_t)it0;      // (n_items < 2^32: <= 65 535 genomes x 2048 buckets)
        x.jpos = item / ng; x.gl = item - x.jpos * ng;
        x.sh = g_sh[x.gl];
        const uint32_t rs = lg_max - (vbits - x.sh);
        if (x.jpos & ((1u << rs) - 1u)) return;                     // (uniform) this genome has fewer buckets
        x.bk = x.jpos >> rs;
        const uint2 d = desc[(uint64_t)g_boff[x.gl] + x.bk];
        x.nk = d.y;
        if (!BIG && x.nk > (uint32_t)(TAB / 2)) return;            // (listed for the workgroup form below: no ids needed)
#pragma unroll
        for (int t = 0; t < NT; t++) { const uint32_t i = t * T + threadIdx.x; x.id[t] = i < x.nk ? kept[d.x + i] : 0u; }
    };
    Item nx;
    fetch(blockIdx.x, nx);
    for (uint64_t it0 = blockIdx.x; it0 < n_items; it0 += gridDim.x) {
        const Item cu = nx;
        fetch(it0 + gridDim.x, nx);
        const uint32_t jpos = cu.jpos, gl = cu.gl, bk = cu.bk, sh = cu.sh, lg = vbits - sh, nk = cu.nk;
        if (nk == 0) continue;
        if (!BIG && nk > (uint32_t)(TAB / 2)) {                    // too many for a wavefront's table: listed for the workgroup form
            if (threadIdx.x == 0) { const uint32_t at = atomicAdd(n_big, 1u); if (at < big_cap) big[at] = (uint32_t)it0; else ovf[gl] = 1; }
            continue;
        }
        uint64_t *qg = q + (uint64_t)gl * m;
        __syncthreads();                                           // the previous bucket's LDS is dead
        for (uint32_t s = threadIdx.x; s < TAB / 4; s += T) ((uint4 *)tab)[s] = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
        for (uint32_t s = threadIdx.x; s < TAB / 8; s += T) ((uint4 *)dup)[s] = make_uint4(0, 0, 0, 0);
        if (threadIdx.x == 0) { s_nc = 0; s_mx = 0; }
        uint64_t thr_b = __hip_atomic_load(&thr[gl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((jpos & ((1u << (lg_max > 4 ? lg_max - 4 : 0)) - 1u)) == 0 && jpos != 0) {      // 15 times per genome: rescan q[], publish the tighter bound
            unsigned long long mx = 0;
            for (uint32_t i0 = 0; i0 < m; i0 += 8 * T) {
                unsigned long long x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const uint32_t i = i0 + u * T + threadIdx.x; x[u] = i < m ? __hip_atomic_load(&qg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull; }
#pragma unroll
                for (int u = 0; u < 8; u++) mx = x[u] > mx ? x[u] : mx;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long y = __shfl_xor(mx, o); mx = y > mx ? y : mx; }
            __syncthreads();
            if (lane == 0) atomicMax(&s_mx, mx);
            __syncthreads();
            const unsigned long long t = s_mx;
            if (threadIdx.x == 0) atomicMin((unsigned long long *)&thr[gl], t);
            if (t < thr_b) thr_b = t;
        }
        const double thr_d = __longlong_as_double((long long)thr_b);
        __syncthreads();
        // ---- the kept ids enter the exact table: CAS + duplicate count as in k_prob_buckets; the lane whose CAS created an entry owns it
        bool over = false;
        auto insert_id = [&](uint32_t id) -> uint32_t {
            uint32_t s = ((id * 0x85EBCA6Bu) >> 16) & (uint32_t)(TAB - 1);      // (the raw id is the k-mer's last bases)
            for (uint32_t probe = 0; probe < (uint32_t)TAB; probe++) {
                const uint32_t old = atomicCAS(&tab[s], EMPTY, id);
                if (old == EMPTY) return s;
                if (old == id) {
                    const uint32_t before = atomicAdd(&dup[s >> 1], 1u << ((s & 1) * 16));
                    if (((before >> ((s & 1) * 16)) & 0xFFFFu) == 0xFFFFu) over = true;
                    return 0xFFFFFFFFu;
                }
                s = (s + 1) & (uint32_t)(TAB - 1);
            }
            over = true;
            return 0xFFFFFFFFu;
        };
        uint32_t own[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) { const uint32_t i = t * T + threadIdx.x; own[t] = 0xFFFFFFFFu; if ((uint32_t)(t * T) < nk && i < nk) own[t] = insert_id(cu.id[t]); }
        if (over) ovf[gl] = 1;
        __syncthreads();
        // ---- the exact (value, multiplicity) pairs
        uint32_t wloc = 0;
        auto entry = [&](uint32_t s) {
            const uint64_t v = pt_value(bk, tab[s], sh, lg);
            const uint32_t w = 1u + ((dup[s >> 1] >> ((s & 1) * 16)) & 0xFFFFu);
            wloc = w > wloc ? w : wloc;
            const double winv = w == 1 ? 1.0 : 1.0 / (double)w;
            const bool alive2 = !(winv > thr_d);
            Rng rg; rg.seed(v);
            const double x = texp_sample(pc, rg);
            const double h = 0.0 + winv * x;
            if (h > thr_d && !alive2) return;                        // the exact multiplicity: the false alarms of shared cells end here
            const uint32_t b = (uint32_t)rng_uint(rg, (uint64_t)m, zone);
            if (!(h > thr_d)) {
                const uint64_t hb = (uint64_t)__double_as_longlong(h);
                uint64_t *slot = qg + b;
                if (hb <= __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    // a possible winner: listed without waiting for the atomic's answer - the claim only takes candidates whose point equals the slot's final minimum
                    (void)__hip_atomic_fetch_min((unsigned long long *)slot, (unsigned long long)hb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t sp = BIG ? seg : my_nc + atomicAdd(&s_nc, 1u);
                    if (sp < seg) { const uint32_t o = blockIdx.x * seg + sp; L.cand_v[o] = v; L.cand_h[o] = hb; L.cand_gb[o] = (uint64_t)gl * m + b; }
                    else {
                        const uint32_t pos = atomicAdd(L.n_cand, 1u);
                        if (pos < L.ovf_cap) { const uint32_t o = L.cand_cap + pos; L.cand_v[o] = v; L.cand_h[o] = hb; L.cand_gb[o] = (uint64_t)gl * m + b; }
                    }
                }
            }
            if (alive2) {                                            // may still reach a slot in pass 2 (superset: thr >= the final max q)
                const uint32_t pos = atomicAdd(L.n_act, 1u);
                if (pos < L.act_cap) {
                    L.akey[pos] = v; L.agl[pos] = gl; L.acnt[pos] = w;
                    L.astate[pos] = rg.s0; L.astate[(uint64_t)L.act_cap + pos] = rg.s1; L.astate[2 * (uint64_t)L.act_cap + pos] = rg.s2; L.astate[3 * (uint64_t)L.act_cap + pos] = rg.s3;
                }
            }
        };
#pragma unroll
        for (int t = 0; t < NT; t++) if ((uint32_t)(t * T) < nk && own[t] != 0xFFFFFFFFu) entry(own[t]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)wloc, o); wloc = y > wloc ? y : wloc; }
        if (lane == 0 && wloc > 1 && wloc > __hip_atomic_load(&wmax[gl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&wmax[gl], wloc);
        __syncthreads();
        if (!BIG) { const uint32_t room = seg - my_nc, got = s_nc; my_nc += got < room ? got : room; }      // (what did not fit went to the shared region)
    }
    if (!BIG && threadIdx.x == 0) L.seg_n[blockIdx.x] = my_nc;
}

// one chunk of genomes [g0, g0 + ng) through the tiered form; lgs = log2(buckets) per genome (host plan). redo as run_prob_buckets.
static int run_prob_tiers(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, const uint64_t *rec_start, const uint64_t *rec_len, const uint64_t *upre,
                          const uint64_t *genome_rec_off, const uint64_t *gunits, uint64_t g0, uint32_t ng, const uint64_t *hk, const uint32_t *lgs, const ProbConst &pc,
                          void *sig_rows, std::vector<uint8_t> &redo, bool &no_room)
{
    const uint32_t m = p->sketch_size, k = p->k;
    no_room = false;
    const bool aa = p->data_t == GS_DATA_AA;
    const int sigbits = gs_value_bits(p);
    const uint64_t zone = uint_zone(m);
    int rc;
    const uint32_t vbits = aa ? 5 * k : 2 * k;
    uint64_t maxk = 0; uint32_t nbmax = 1, nbt = 0;
    for (uint32_t i = 0; i < ng; i++) maxk = std::max(maxk, hk[i]);
    // parts per genome (one for the chunk), a power of two (the filter kernel deals slices to its eight wavefronts): four where the chunk has the genomes to fill the
    // device with them (fewer, longer slices: 1.11e11 k-mers/s at 4 against 0.99e11 at 16 over 256 x 5 Mbp), more for a handful of genomes - down to 8 tiles per part
    uint32_t parts = 1;
    {
        const uint64_t tiles = maxk / ((uint64_t)PT_T * 32) + 1;
        while (parts < 4 && (uint64_t)parts * 2 * 8 <= tiles) parts *= 2;
        while (parts < 32 && (uint64_t)parts * 2 * 8 <= tiles && (uint64_t)ng * parts < 2 * (uint64_t)c->n_cu) parts *= 2;
    }
    if (getenv("GS_PROB_PARTS")) { parts = 1; const uint32_t want = (uint32_t)atoi(getenv("GS_PROB_PARTS")); while (parts * 2 <= want && parts < 32) parts *= 2; }
    // per genome: shift, flat bucket offset, slice capacity (mean + 5 sigma of a slice's Poisson-like fill), offset of its slices (in 4-byte ids)
    std::vector<uint32_t> info(3 * (size_t)ng + 1); std::vector<uint64_t> vbase(ng); std::vector<uint64_t> capbits(2 * (size_t)ng); std::vector<double> capd(ng);      // capbits: [ng] caps, then [ng] t_one (below)
    uint32_t *sh = info.data(), *boff = sh + ng, *cap = boff + ng + 1;
    uint64_t T32 = 0;
    const double cap_c = getenv("GS_PROB_CAP_C") ? atof(getenv("GS_PROB_CAP_C")) : 10.0;     // P(a genome fails the check) = e^-c
    for (uint32_t i = 0; i < ng; i++) {
        const uint32_t lg = lgs[i];
        sh[i] = vbits - lg; boff[i] = nbt; nbt += 1u << lg; nbmax = std::max(nbmax, 1u << lg);
        const double e = (double)hk[i] / ((double)(1u << lg) * parts);
        cap[i] = ((uint32_t)(e + 5.0 * sqrt(e) + 16.0) + 63u) & ~63u;       // (a multiple of 64: the bucket kernel reads a slice in whole wavefront trips)
        vbase[i] = T32; T32 += ((uint64_t)parts << lg) * cap[i];
        // speculative cap of max_b q[b]: the points of a genome form a process of rate N (its k-mers with multiplicity) over m slots
        const double t = (double)m / (double)hk[i] * (log((double)m) + cap_c);
        capd[i] = t > 0.0 ? t : 0x1.0p-1000;
        memcpy(&capbits[i], &capd[i], 8);
        // the cap as a threshold on the 52 uniform bits K of a first draw x0 = c1 K 2^-52 (k_prob_tier_filter): x0 > c cap for K > c t_one, with a margin of 2^-40 relative
        // on the safe side (+ 4: the roundings of this line and the truncation leave t_one >= the exact product + 1) - once per genome here, not per bucket and lane there
        const double t1d = capd[i] / pc.c1 * 0x1.0p52 * (1.0 + 0x1.0p-40) + 4.0;
        capbits[ng + i] = t1d < 0x1.0p52 ? (uint64_t)t1d : ((uint64_t)1 << 52);
    }
    boff[ng] = nbt;
    PoolBuf dinfo(c, 0), dvb(c, 2), cnt(c, 3), vals(c, 8), q(c, 9), qprev(c, 10), sig(c, 11), sigpass(c, 12), thr(c, 13), wmax(c, 14), qmax(c, 15), ctr(c, 7);
    PoolBuf cv(c, 16), chh(c, 17), cgb(c, 18), akey(c, 19), agl(c, 24), acnt(c, 25), astate(c, 26), ph(c, 27), pb(c, 37), ovf(c, 38), segn(c, 28);
    PoolBuf kept(c, 61), desc(c, 4), big(c, 5);
    const uint32_t cand_cap = (uint32_t)std::min<uint64_t>((uint64_t)ng * m * 16 + 65536, (uint64_t)1 << 30), ovf_cap = cand_cap / 4, act_cap = 1u << 24;
    // the list of kept ids: what the caps let through (x1 < cap as singletons) plus the false alarms of shared cells and the real repeats, with room to spare
    uint64_t kept_want = 1u << 20;
    for (uint32_t i = 0; i < ng; i++) kept_want += (uint64_t)((double)hk[i] * std::min(1.0, 1.5 * capd[i] + 0.05));
    const uint32_t kept_cap = (uint32_t)std::min<uint64_t>(kept_want, 0xFFFF0000u), big_cap = 1u << 16;
    const uint32_t pts_max = (uint32_t)c->n_cu * 32;              // blocks of the wavefront-per-bucket kernel at most (segment counts)
    if ((rc = dinfo.alloc(4 * info.size())) || (rc = dvb.alloc(8 * (size_t)ng)) || (rc = cnt.alloc((size_t)4 * nbt * parts)) || (rc = vals.alloc(4 * (size_t)T32 + 64)) ||
        (rc = q.alloc((size_t)8 * ng * m)) || (rc = qprev.alloc((size_t)8 * ng * m)) || (rc = sig.alloc((size_t)8 * ng * m)) || (rc = sigpass.alloc((size_t)8 * ng * m)) ||
        (rc = thr.alloc(16 * (size_t)ng)) || (rc = wmax.alloc(4 * (size_t)ng)) || (rc = qmax.alloc(8 * (size_t)ng)) || (rc = ctr.alloc(64)) ||
        (rc = cv.alloc((size_t)8 * (cand_cap + ovf_cap))) || (rc = chh.alloc((size_t)8 * (cand_cap + ovf_cap))) || (rc = cgb.alloc((size_t)8 * (cand_cap + ovf_cap))) ||
        (rc = ovf.alloc(4 * (size_t)ng)) || (rc = segn.alloc((size_t)4 * pts_max)) || (rc = kept.alloc(4 * (size_t)kept_cap + 64)) || (rc = desc.alloc(8 * (size_t)nbt)) ||
        (rc = big.alloc(4 * (size_t)big_cap)) ||
        (rc = akey.alloc((size_t)8 * act_cap)) || (rc = agl.alloc((size_t)4 * act_cap)) || (rc = acnt.alloc((size_t)4 * act_cap)) || (rc = astate.alloc((size_t)32 * act_cap))) {
        no_room = true;                                               // (the one failure the caller answers with the older forms; anything later is an error)
        return rc;
    }
    const uint32_t *d_sh = dinfo.as<uint32_t>(), *d_boff = d_sh + ng, *d_cap = d_boff + ng + 1;
    GS_HIP_CHECK(hipMemcpyAsync(dinfo.p, info.data(), 4 * info.size(), hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dvb.p, vbase.data(), 8 * (size_t)ng, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(thr.p, capbits.data(), 16 * (size_t)ng, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_prob_init, dim3(c->n_cu * 4), dim3(256), 0, c->stream, q.as<uint64_t>(), qprev.as<uint64_t>(), sig.as<uint64_t>(), sigpass.as<uint64_t>(), ng * (uint64_t)m);
    {
        std::vector<uint32_t> ones(ng, 1u);
        GS_HIP_CHECK(hipMemcpyAsync(wmax.p, ones.data(), 4 * (size_t)ng, hipMemcpyHostToDevice, c->stream));
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));               // (the host vectors above go out of use here)
    }
    GS_HIP_CHECK(hipMemsetAsync(ovf.p, 0, 4 * (size_t)ng, c->stream));
    GS_HIP_CHECK(hipMemsetAsync(ctr.p, 0, 64, c->stream));          // [2] n_cand, [3] n_act, [4] n_active genomes, [5] kept ids, [6] big buckets
    uint32_t *ctr32 = ctr.as<uint32_t>();
    {
        ProfScope ps(c, FAM_SKETCH);
        const size_t lds = ((size_t)3 * nbmax + 8 + (size_t)PT_T * 32) * 4;
        dim3 grid(parts, ng), block(PT_T);
#define GS_LAUNCH_PT1(AAV)                                                                                                  \
    do {                                                                                                                    \
        auto kern = k_prob_part1<AAV>;                                                                                      \
        GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipLaunchKernelGGL(kern, grid, block, lds, c->stream, seq, rec_start, rec_len, upre, genome_rec_off, gunits, g0, kq_of(p), vbits, d_sh, d_boff, dvb.as<uint64_t>(), d_cap, \
                           parts, vals.as<uint32_t>(), cnt.as<uint32_t>(), ovf.as<uint32_t>());                             \
    } while (0)
        if (aa) GS_LAUNCH_PT1(true);
        else if (getenv("GS_PROB_TWOWALK")) GS_LAUNCH_PT1(false);
        else {
            GS_HIP_CHECK(hipFuncSetAttribute((const void *)k_prob_part1_dna, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_prob_part1_dna, grid, block, lds, c->stream, seq, rec_start, rec_len, upre, genome_rec_off, gunits, g0, kq_of(p), vbits, d_sh, d_boff, dvb.as<uint64_t>(), d_cap,
                               parts, vals.as<uint32_t>(), cnt.as<uint32_t>(), ovf.as<uint32_t>());
        }
#undef GS_LAUNCH_PT1
        GS_HIP_CHECK(hipGetLastError());
        PbLists L{cv.as<uint64_t>(), chh.as<uint64_t>(), cgb.as<uint64_t>(), cand_cap, ovf_cap, ctr32 + 2, segn.as<uint32_t>(), akey.as<uint64_t>(), agl.as<uint32_t>(), acnt.as<uint32_t>(),
                  astate.as<uint64_t>(), act_cap, ctr32 + 3, nullptr};
        DevBuf profbuf; unsigned long long *prof = nullptr;
        if (getenv("GS_PROB_PROFILE")) { if ((rc = profbuf.alloc(128))) return rc; GS_HIP_CHECK(hipMemsetAsync(profbuf.p, 0, 128, c->stream)); prof = profbuf.as<unsigned long long>(); }
        uint32_t lg_max = 0; while ((1u << lg_max) < nbmax) lg_max++;
        const uint64_t n_items = (uint64_t)ng << lg_max;
        int f_cu = 4, w_cu = 16, b_cu = 2;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&f_cu, (const void *)k_prob_tier_filter, PT2_T, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&w_cu, (const void *)k_prob_tier_points<64, 1024, false>, 64, 0);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&b_cu, (const void *)k_prob_tier_points<512, 4096, true>, 512, 0);
        const uint32_t fwgs = (uint32_t)std::min<uint64_t>(n_items, (uint64_t)c->n_cu * std::min(std::max(f_cu, 1), 8));
        const uint32_t pwgs = (uint32_t)std::min<uint64_t>(n_items, std::min<uint64_t>((uint64_t)c->n_cu * std::min(std::max(w_cu, 1), 32), pts_max));
        hipLaunchKernelGGL(k_prob_tier_filter, dim3(fwgs), dim3(PT2_T), 0, c->stream, vals.as<uint32_t>(), dvb.as<uint64_t>(), d_cap, cnt.as<uint32_t>(), parts, vbits, d_sh, d_boff, ng, lg_max,
                           pc, thr.as<uint64_t>(), kept.as<uint32_t>(), kept_cap, ctr32 + 5, desc.as<uint2>(), ovf.as<uint32_t>(), prof);
        hipLaunchKernelGGL((k_prob_tier_points<64, 1024, false>), dim3(pwgs), dim3(64), 0, c->stream, kept.as<uint32_t>(), desc.as<uint2>(), vbits, d_sh, d_boff, ng, lg_max, m, zone, pc,
                           q.as<uint64_t>(), thr.as<uint64_t>(), wmax.as<uint32_t>(), L, ovf.as<uint32_t>(), big.as<uint32_t>(), ctr32 + 6, big_cap);
        hipLaunchKernelGGL((k_prob_tier_points<512, 4096, true>), dim3((uint32_t)c->n_cu * std::min(std::max(b_cu, 1), 2)), dim3(512), 0, c->stream, kept.as<uint32_t>(), desc.as<uint2>(), vbits,
                           d_sh, d_boff, ng, lg_max, m, zone, pc, q.as<uint64_t>(), thr.as<uint64_t>(), wmax.as<uint32_t>(), L, ovf.as<uint32_t>(), big.as<uint32_t>(), ctr32 + 6, big_cap);
        if (prof) {
            unsigned long long h[16]; uint32_t hcn[8];
            GS_HIP_CHECK(hipMemcpyAsync(h, prof, 128, hipMemcpyDeviceToHost, c->stream));
            GS_HIP_CHECK(hipMemcpyAsync(hcn, ctr.p, 32, hipMemcpyDeviceToHost, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            const double it = (double)std::max<unsigned long long>(h[7], 1);
            fprintf(stderr, "[GS_PROB_PROFILE] tiers: filter workgroup 0 of %u (%d per CU, %u parts; points: %u wavefronts, %d per CU): %llu buckets, keys/bucket %.0f, kept %.0f | cycles per bucket: load+zero %.0f, count %.0f, draw+queue %.0f, write %.0f | kept ids %u of %u, big buckets %u\n",
                    fwgs, f_cu, parts, pwgs, w_cu, h[7], h[8] / it, h[6] / it, h[0] / it, h[1] / it, h[2] / it, h[3] / it, hcn[5], kept_cap, hcn[6]);
        }
        hipLaunchKernelGGL(k_prob_claim_list, dim3(pwgs + 1), dim3(256), 0, c->stream, cv.as<uint64_t>(), chh.as<uint64_t>(), cgb.as<uint64_t>(), segn.as<uint32_t>(), cand_cap / pwgs, pwgs,
                           cand_cap, ctr32 + 2, ovf_cap, q.as<uint64_t>(), sigpass.as<uint64_t>());
        hipLaunchKernelGGL(k_prob_fold, dim3(ng), dim3(256), 0, c->stream, m, 1u, q.as<uint64_t>(), qprev.as<uint64_t>(), sig.as<uint64_t>(), sigpass.as<uint64_t>(), wmax.as<uint32_t>(),
                           qmax.as<double>(), ctr32 + 4);
        GS_HIP_CHECK(hipGetLastError());
    }
    uint32_t hc[8]; std::vector<uint32_t> hovf(ng); std::vector<double> hq(ng);
    GS_HIP_CHECK(hipMemcpyAsync(hc, ctr.p, 32, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(hovf.data(), ovf.p, 4 * (size_t)ng, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(hq.data(), qmax.p, 8 * (size_t)ng, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    redo.assign(ng, 0);
    if (hc[2] > ovf_cap || hc[3] > act_cap) { redo.assign(ng, 1); return GS_OK; }
    // the speculation is checked here: q only decreases in later passes, so a maximum under the cap after pass 1 stays there
    for (uint32_t i = 0; i < ng; i++) redo[i] = (hovf[i] || !(hq[i] <= capd[i])) ? 1 : 0;
    if (getenv("GS_PROB_VERBOSE"))
        for (uint32_t i = 0; i < ng; i++) if (redo[i]) fprintf(stderr, "[GS_PROB] tiered form: genome %llu flagged (%s; max slot minimum %g, cap %g)\n", (unsigned long long)(g0 + i),
                                                               hovf[i] ? "a slice, the kept-id queue or a table overflowed" : "cap not confirmed", hq[i], capd[i]);
    uint32_t na = hc[4];
    if ((rc = prob_retire_flagged(c, redo, ng, wmax.as<uint32_t>(), qmax.as<double>(), na))) return rc;
    const uint32_t n_list = hc[3];
    if (na && n_list) {
        if ((rc = ph.alloc((size_t)8 * n_list)) || (rc = pb.alloc((size_t)4 * n_list))) return rc;
        const uint32_t lg = std::max<uint32_t>(1, std::min<uint32_t>((n_list + 255) / 256, (uint32_t)c->n_cu * 16));
        for (uint32_t it = 2; na; it++) {
            GS_HIP_CHECK(hipMemsetAsync(ctr32 + 4, 0, 4, c->stream));
            hipLaunchKernelGGL(k_prob_point_act, dim3(lg), dim3(256), 0, c->stream, akey.as<uint64_t>(), agl.as<uint32_t>(), acnt.as<uint32_t>(), n_list, act_cap, m, zone, pc, it,
                               qmax.as<double>(), q.as<uint64_t>(), astate.as<uint64_t>(), ph.as<uint64_t>(), pb.as<uint32_t>());
            hipLaunchKernelGGL(k_prob_claim_act, dim3(lg), dim3(256), 0, c->stream, akey.as<uint64_t>(), agl.as<uint32_t>(), n_list, m, q.as<uint64_t>(), ph.as<uint64_t>(), pb.as<uint32_t>(),
                               sigpass.as<uint64_t>());
            hipLaunchKernelGGL(k_prob_fold, dim3(ng), dim3(256), 0, c->stream, m, it, q.as<uint64_t>(), qprev.as<uint64_t>(), sig.as<uint64_t>(), sigpass.as<uint64_t>(), wmax.as<uint32_t>(),
                               qmax.as<double>(), ctr32 + 4);
            GS_HIP_CHECK(hipGetLastError());
            GS_HIP_CHECK(hipMemcpyAsync(&na, ctr32 + 4, 4, hipMemcpyDeviceToHost, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (getenv("GS_PROB_VERBOSE") && (it < 8 || (it & (it - 1)) == 0)) fprintf(stderr, "[GS_PROB] pass %u done: %u genomes still active, %u elements on the active list\n", it, na, n_list);
        }
    }
    if (sigbits == 32) hipLaunchKernelGGL(k_prob_write<uint32_t>, dim3(c->n_cu * 4), dim3(256), 0, c->stream, q.as<uint64_t>(), sig.as<uint64_t>(), ng * (uint64_t)m, (uint32_t *)sig_rows);
    else hipLaunchKernelGGL(k_prob_write<uint64_t>, dim3(c->n_cu * 4), dim3(256), 0, c->stream, q.as<uint64_t>(), sig.as<uint64_t>(), ng * (uint64_t)m, (uint64_t *)sig_rows);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// prob driver: runs of genomes the tiered form suits go through it in chunks; what it flags, and the genomes it does not suit, go through the bucketed form
// (chunks again), and what that one flags or does not suit through the sorted form
static int run_prob(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, uint64_t seq_bytes, const uint64_t *rec_start, const uint64_t *rec_len,
                    uint64_t n_rec, const uint64_t *genome_rec_off, uint64_t n_genomes, void *sig_out)
{
    const uint32_t m = p->sketch_size, k = p->k;
    const char *e = getenv("GS_PROB_IMPL");
    if (e && !strcmp(e, "sort")) return run_prob_sorted(c, p, seq, seq_bytes, rec_start, rec_len, n_rec, genome_rec_off, n_genomes, sig_out);
    const bool tiers_on = !(e && !strcmp(e, "buckets"));
    ProbConst pc;
    pc.lambda = log((double)m / (double)(m - 1));
    pc.c1 = expm1(pc.lambda) / pc.lambda;
    pc.c2 = log(2.0 / (1.0 + exp(-pc.lambda))) / pc.lambda;
    pc.c3 = (1.0 - exp(-pc.lambda)) / pc.lambda;
    int rc;
    DevBuf upre, gunits, kpre, gkm;              // (own allocations: the sorted form called below uses the scratch-pool slots of the same names)
    if ((rc = upre.alloc(8 * (n_rec + 1))) || (rc = gunits.alloc(8 * n_genomes)) || (rc = kpre.alloc(8 * (n_rec + 1))) || (rc = gkm.alloc(8 * n_genomes))) return rc;
    const uint32_t gb = (uint32_t)((n_genomes + 3) / 4);
    hipLaunchKernelGGL(k_unit_prefix, dim3(gb), dim3(256), 0, c->stream, rec_start, rec_len, genome_rec_off, n_genomes, k, upre.as<uint64_t>(), gunits.as<uint64_t>());
    hipLaunchKernelGGL(k_kmer_prefix, dim3(gb), dim3(256), 0, c->stream, rec_len, genome_rec_off, n_genomes, k, kpre.as<uint64_t>(), gkm.as<uint64_t>());
    GS_HIP_CHECK(hipGetLastError());
    std::vector<uint64_t> hk(n_genomes);
    GS_HIP_CHECK(hipMemcpyAsync(hk.data(), gkm.p, 8 * n_genomes, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    const size_t row = (size_t)m * (gs_value_bits(p) / 8);
    auto suits = [&](uint64_t g) { return hk[g] >= (uint64_t)64 * m && hk[g] <= (uint64_t)PB_NBMAX * PB_AVG; };
    // tiered form: the in-bucket id must fit 4 bytes (>= vbits - 31 bucket bits), <= 2^PT_LGMAX buckets of PT_MINB .. 3 PT_AVG k-mers
    const uint32_t vbits = p->data_t == GS_DATA_AA ? 5 * k : 2 * k;
    std::vector<uint32_t> lgs(n_genomes, 0xFFFFFFFFu);
    const uint64_t pt_avg = getenv("GS_PROB_PT_AVG") ? (uint64_t)std::max(256, atoi(getenv("GS_PROB_PT_AVG"))) : (uint64_t)PT_AVG;
    if (tiers_on && vbits <= 31 + (uint32_t)PT_LGMAX)
        for (uint64_t g = 0; g < n_genomes; g++) {
            if (hk[g] < (uint64_t)64 * m) continue;
            uint32_t lg = vbits > 31 ? vbits - 31 : 0;
            while ((pt_avg << lg) < hk[g] && lg < (uint32_t)PT_LGMAX && lg + 1 < vbits) lg++;
            if ((hk[g] >> lg) >= (uint64_t)PT_MINB && (hk[g] >> lg) <= 3 * pt_avg) lgs[g] = lg;
        }
    auto suits_tiers = [&](uint64_t g) { return lgs[g] != 0xFFFFFFFFu; };
    const uint64_t max_items = (uint64_t)3 << 29;                 // ~1.6e9 k-mers per chunk (12.9 GB of bucketed values)
    // the tiered form's chunks: ~8 bytes of scratch per k-mer (slices of 4-byte ids with their slack, kept ids, lists), so twice the k-mers of a bucketed chunk where a quarter of
    // the free device memory holds them (2048 x 5 Mbp: 1.07e11 k-mers/s at 1.6e9 per chunk, 1.11e11 at 3.2e9, 1.13e11 at 6.4e9 - a chunk ends in a host round trip)
    uint64_t tier_items = 2 * max_items;
    {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) tier_items = std::min<uint64_t>(tier_items, std::max<uint64_t>(max_items / 4, (uint64_t)fr / 4 / 8));
        else (void)hipGetLastError();
        if (getenv("GS_PROB_CHUNK_KMERS")) tier_items = std::max<uint64_t>(1, (uint64_t)atoll(getenv("GS_PROB_CHUNK_KMERS")));
    }
    // [a, b) through the bucketed form where it suits, else (and what it flags) through the sorted form
    auto old_range = [&](uint64_t a, uint64_t b) -> int {
        for (uint64_t g0 = a; g0 < b;) {
            uint64_t g1 = g0 + 1;
            if (!suits(g0)) {
                while (g1 < b && !suits(g1)) g1++;
                GS_HIP_CHECK(hipStreamSynchronize(c->stream));
                if ((rc = run_prob_sorted(c, p, seq, seq_bytes, rec_start, rec_len, n_rec, genome_rec_off + g0, g1 - g0, (uint8_t *)sig_out + row * g0))) return rc;
                g0 = g1;
                continue;
            }
            uint64_t T = hk[g0];
            while (g1 < b && suits(g1) && g1 - g0 < 65535 && T + hk[g1] <= max_items) { T += hk[g1]; g1++; }
            std::vector<uint8_t> redo;
            if ((rc = run_prob_buckets(c, p, seq, rec_start, rec_len, upre.as<uint64_t>(), genome_rec_off, gunits.as<uint64_t>(), g0, (uint32_t)(g1 - g0), hk.data() + g0, pc,
                                       (uint8_t *)sig_out + row * g0, redo))) return rc;
            for (uint64_t g = g0; g < g1;) {                          // flagged genomes, in runs
                if (!redo[g - g0]) { g++; continue; }
                uint64_t h = g + 1;
                while (h < g1 && redo[h - g0]) h++;
                GS_HIP_CHECK(hipStreamSynchronize(c->stream));
                if ((rc = run_prob_sorted(c, p, seq, seq_bytes, rec_start, rec_len, n_rec, genome_rec_off + g, h - g, (uint8_t *)sig_out + row * g))) return rc;
                g = h;
            }
            g0 = g1;
        }
        return GS_OK;
    };
    for (uint64_t g0 = 0; g0 < n_genomes;) {
        uint64_t g1 = g0 + 1;
        if (!suits_tiers(g0)) {
            while (g1 < n_genomes && !suits_tiers(g1)) g1++;
            if ((rc = old_range(g0, g1))) return rc;
            g0 = g1;
            continue;
        }
        uint64_t T = hk[g0];
        while (g1 < n_genomes && suits_tiers(g1) && g1 - g0 < 65535 && T + hk[g1] <= tier_items) { T += hk[g1]; g1++; }
        std::vector<uint8_t> redo; bool no_room = false;
        rc = run_prob_tiers(c, p, seq, rec_start, rec_len, upre.as<uint64_t>(), genome_rec_off, gunits.as<uint64_t>(), g0, (uint32_t)(g1 - g0), hk.data() + g0, lgs.data() + g0, pc,
                            (uint8_t *)sig_out + row * g0, redo, no_room);
        if (rc && no_room) {                                      // its scratch did not fit (an index with its pair cache beside the sketcher): the chunk takes the older forms, which
            (void)hipGetLastError();                                  // work in smaller chunks and have their own ways down
            (void)hipStreamSynchronize(c->stream);
            if (getenv("GS_PROB_VERBOSE")) fprintf(stderr, "[GS_PROB] tiered form: no room for the scratch of genomes [%llu, %llu) (%s): the bucketed form takes them\n", (unsigned long long)g0,
                                                   (unsigned long long)g1, gs_last_error());
            redo.assign(g1 - g0, 1);
            rc = GS_OK;
        }
        if (rc) return rc;
        for (uint64_t g = g0; g < g1;) {                              // flagged genomes (slice / table overflow, cap not confirmed), in runs: the exact fallback
            if (!redo[g - g0]) { g++; continue; }
            uint64_t h = g + 1;
            while (h < g1 && redo[h - g0]) h++;
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (getenv("GS_PROB_VERBOSE")) fprintf(stderr, "[GS_PROB] tiered form flagged genomes [%llu, %llu): redone by the bucketed form\n", (unsigned long long)g, (unsigned long long)h);
            if ((rc = old_range(g, h))) return rc;
            g = h;
        }
        g0 = g1;
    }
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

static int run_prob_sorted(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, uint64_t seq_bytes, const uint64_t *rec_start, const uint64_t *rec_len,
                           uint64_t n_rec, const uint64_t *genome_rec_off, uint64_t n_genomes, void *sig_out)
{
    const uint32_t m = p->sketch_size, k = p->k;
    const bool aa = p->data_t == GS_DATA_AA;
    const uint32_t vbits = aa ? 5 * k : 2 * k;
    const int sigbits = gs_value_bits(p);
    const uint64_t zone = uint_zone(m);
    ProbConst pc;
    pc.lambda = log((double)m / (double)(m - 1));
    pc.c1 = expm1(pc.lambda) / pc.lambda;
    pc.c2 = log(2.0 / (1.0 + exp(-pc.lambda))) / pc.lambda;
    pc.c3 = (1.0 - exp(-pc.lambda)) / pc.lambda;
    int rc;
    PoolBuf upre(c, 28), gunits(c, 29), kpre(c, 30), gkm(c, 31);
    if ((rc = upre.alloc(8 * (n_rec + 1)))) return rc;
    if ((rc = gunits.alloc(8 * n_genomes))) return rc;
    if ((rc = kpre.alloc(8 * (n_rec + 1)))) return rc;
    if ((rc = gkm.alloc(8 * n_genomes))) return rc;
    const uint32_t gb = (uint32_t)((n_genomes + 3) / 4);                 // one wavefront per genome
    hipLaunchKernelGGL(k_unit_prefix, dim3(gb), dim3(256), 0, c->stream, rec_start, rec_len, genome_rec_off, n_genomes, k, upre.as<uint64_t>(), gunits.as<uint64_t>());
    hipLaunchKernelGGL(k_kmer_prefix, dim3(gb), dim3(256), 0, c->stream, rec_len, genome_rec_off, n_genomes, k, kpre.as<uint64_t>(), gkm.as<uint64_t>());
    GS_HIP_CHECK(hipGetLastError());
    std::vector<uint64_t> hk(n_genomes);
    GS_HIP_CHECK(hipMemcpyAsync(hk.data(), gkm.p, 8 * n_genomes, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    // chunks of genomes: the composite sort key needs log2(chunk) spare bits; memory bounds the k-mer count
    const uint64_t max_items = (uint64_t)3 << 28;                       // ~8e8 k-mers per chunk (6.4 GB of keys, twice)
    const uint64_t max_g = vbits >= 64 ? 1 : std::min<uint64_t>((uint64_t)1 << std::min<uint32_t>(64 - vbits, 16), 65535);
    const size_t row = (size_t)m * (sigbits / 8);
    for (uint64_t g0 = 0; g0 < n_genomes;) {
        uint64_t ng = 0, T = 0;
        std::vector<uint64_t> base;
        while (g0 + ng < n_genomes && ng < max_g && (ng == 0 || T + hk[g0 + ng] <= max_items)) { base.push_back(T); T += hk[g0 + ng]; ng++; }
        GS_REQUIRE(T < ((uint64_t)1 << 31), GS_ERR_UNSUPPORTED, "a single genome with more than 2^31 k-mers is not supported by the prob sketcher");
        PoolBuf dbase(c, 0), q(c, 1), qprev(c, 2), sig(c, 3), sigpass(c, 4), wmax(c, 5), qmax(c, 6), nact(c, 7);
        if ((rc = dbase.alloc(8 * ng))) return rc;
        if ((rc = q.alloc(8 * ng * m))) return rc;
        if ((rc = qprev.alloc(8 * ng * m))) return rc;
        if ((rc = sig.alloc(8 * ng * m))) return rc;
        if ((rc = sigpass.alloc(8 * ng * m))) return rc;
        if ((rc = wmax.alloc(4 * ng))) return rc;
        if ((rc = qmax.alloc(8 * ng))) return rc;
        if ((rc = nact.alloc(64))) return rc;
        GS_HIP_CHECK(hipMemcpyAsync(dbase.p, base.data(), 8 * ng, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_prob_init, dim3(c->n_cu * 4), dim3(256), 0, c->stream, q.as<uint64_t>(), qprev.as<uint64_t>(), sig.as<uint64_t>(), sigpass.as<uint64_t>(), ng * (uint64_t)m);
        GS_HIP_CHECK(hipMemsetAsync(wmax.p, 0, 4 * ng, c->stream));
        {   // qmax = +inf
            std::vector<double> inf(ng, INFINITY);
            GS_HIP_CHECK(hipMemcpyAsync(qmax.p, inf.data(), 8 * ng, hipMemcpyHostToDevice, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        }
        if (T > 0) {
            PoolBuf vals(c, 8), sorted(c, 9), ukey(c, 10), ucnt(c, 11), nruns(c, 12), tmp(c, 13), candh(c, 14), candb(c, 15);
            if ((rc = vals.alloc(8 * T))) return rc;
            if ((rc = sorted.alloc(8 * T))) return rc;
            const uint64_t avg_units = (aa ? seq_bytes / 32 : seq_bytes / 8) / n_genomes + 1;
            uint32_t parts = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(avg_units / SK_THREADS + 1, (4 * (uint64_t)c->n_cu + ng - 1) / ng));
            {
                ProfScope ps(c, FAM_SKETCH);
                dim3 grid(parts, (uint32_t)ng), block(SK_THREADS);
                if (aa) hipLaunchKernelGGL(k_emit_values<true>, grid, block, 0, c->stream, seq, rec_start, rec_len, upre.as<uint64_t>(), kpre.as<uint64_t>(), genome_rec_off, gunits.as<uint64_t>(), dbase.as<uint64_t>(), g0, kq_of(p), vbits, vals.as<uint64_t>());
                else hipLaunchKernelGGL(k_emit_values<false>, grid, block, 0, c->stream, seq, rec_start, rec_len, upre.as<uint64_t>(), kpre.as<uint64_t>(), genome_rec_off, gunits.as<uint64_t>(), dbase.as<uint64_t>(), g0, kq_of(p), vbits, vals.as<uint64_t>());
                GS_HIP_CHECK(hipGetLastError());
            }
            int endbit = 64;
            if (vbits < 64) { endbit = (int)vbits; uint64_t x = ng - 1; while (x) { endbit++; x >>= 1; } if (endbit > 64) endbit = 64; }
            // multiplicities = run lengths of the sorted (genome, value) keys: own LSD radix sort + run-length encoding (gs_radix.hip)
            PoolBuf pos(c, 47);
            if ((rc = tmp.alloc(radix_scratch_bytes(T)))) return rc;
            if ((rc = pos.alloc(4 * T))) return rc;
            uint64_t *srt = nullptr;
            if ((rc = radix_sort_u64(c, vals.as<uint64_t>(), sorted.as<uint64_t>(), T, endbit, tmp.p, &srt))) return rc;
            // distinct elements + multiplicities: the unique keys go to the buffer the sorted keys are NOT in, which the rest of the pass calls `vals`
            if (srt == vals.as<uint64_t>()) { std::swap(vals.p, sorted.p); std::swap(vals.bytes, sorted.bytes); }
            if ((rc = ucnt.alloc(4 * T))) return rc;
            if ((rc = nruns.alloc(64))) return rc;
            if ((rc = run_length_encode_u64(c, sorted.as<uint64_t>(), T, vals.as<uint64_t>(), ucnt.as<uint32_t>(), nruns.as<uint32_t>(), pos.as<uint32_t>(), tmp.p))) return rc;
            uint32_t ne32 = 0;
            GS_HIP_CHECK(hipMemcpyAsync(&ne32, nruns.p, 4, hipMemcpyDeviceToHost, c->stream));
            GS_HIP_CHECK(hipStreamSynchronize(c->stream));
            const uint64_t ne = ne32;
            sorted.release();
            if ((rc = candh.alloc(8 * ne))) return rc;
            if ((rc = candb.alloc(4 * ne))) return rc;
            const uint32_t eg = (uint32_t)std::min<uint64_t>((ne + 255) / 256, (uint64_t)c->n_cu * 16);
            const uint32_t ACT_CAP = 1u << 24;                       // 16 M live elements keep their generator state (0.5 GB)
            PoolBuf akey(c, 16), acnt(c, 17), astate(c, 18), nlist(c, 19);
            uint32_t n_list = 0; bool use_list = false;
            for (uint32_t it = 1;; it++) {
                GS_HIP_CHECK(hipMemsetAsync(nact.p, 0, 4, c->stream));
                if (!use_list) {
                    hipLaunchKernelGGL(k_prob_point, dim3(eg), dim3(256), 0, c->stream, vals.as<uint64_t>(), ucnt.as<uint32_t>(), ne, vbits, m, zone, pc, it, qmax.as<double>(),
                                       q.as<uint64_t>(), candh.as<uint64_t>(), candb.as<uint32_t>(), it == 1 ? wmax.as<uint32_t>() : nullptr);
                    hipLaunchKernelGGL(k_prob_claim, dim3(eg), dim3(256), 0, c->stream, vals.as<uint64_t>(), ne, vbits, m, q.as<uint64_t>(), candh.as<uint64_t>(), candb.as<uint32_t>(), sigpass.as<uint64_t>());
                } else {
                    const uint32_t lg = std::max<uint32_t>(1, std::min<uint32_t>((n_list + 255) / 256, (uint32_t)c->n_cu * 16));
                    hipLaunchKernelGGL(k_prob_point_list, dim3(lg), dim3(256), 0, c->stream, akey.as<uint64_t>(), acnt.as<uint32_t>(), n_list, ACT_CAP, vbits, m, zone, pc, it,
                                       qmax.as<double>(), q.as<uint64_t>(), astate.as<uint64_t>(), candh.as<uint64_t>(), candb.as<uint32_t>());
                    hipLaunchKernelGGL(k_prob_claim, dim3(lg), dim3(256), 0, c->stream, akey.as<uint64_t>(), (uint64_t)n_list, vbits, m, q.as<uint64_t>(), candh.as<uint64_t>(), candb.as<uint32_t>(), sigpass.as<uint64_t>());
                }
                hipLaunchKernelGGL(k_prob_fold, dim3((uint32_t)ng), dim3(256), 0, c->stream, m, it, q.as<uint64_t>(), qprev.as<uint64_t>(), sig.as<uint64_t>(), sigpass.as<uint64_t>(),
                                   wmax.as<uint32_t>(), qmax.as<double>(), nact.as<uint32_t>());
                GS_HIP_CHECK(hipGetLastError());
                uint32_t na = 0;
                GS_HIP_CHECK(hipMemcpyAsync(&na, nact.p, 4, hipMemcpyDeviceToHost, c->stream));
                GS_HIP_CHECK(hipStreamSynchronize(c->stream));
                if (na == 0) break;
                if (it == 1) {      // survivors of pass 1 -> compact list with generator state (falls back to replay when it overflows)
                    if ((rc = nlist.alloc(64))) return rc;
                    if ((rc = akey.alloc(8 * (size_t)ACT_CAP))) return rc;
                    if ((rc = acnt.alloc(4 * (size_t)ACT_CAP))) return rc;
                    if ((rc = astate.alloc(32 * (size_t)ACT_CAP))) return rc;
                    GS_HIP_CHECK(hipMemsetAsync(nlist.p, 0, 4, c->stream));
                    hipLaunchKernelGGL(k_prob_compact, dim3(eg), dim3(256), 0, c->stream, vals.as<uint64_t>(), ucnt.as<uint32_t>(), ne, vbits, m, zone, pc, qmax.as<double>(), ACT_CAP,
                                       nlist.as<uint32_t>(), akey.as<uint64_t>(), acnt.as<uint32_t>(), astate.as<uint64_t>());
                    GS_HIP_CHECK(hipGetLastError());
                    GS_HIP_CHECK(hipMemcpyAsync(&n_list, nlist.p, 4, hipMemcpyDeviceToHost, c->stream));
                    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
                    use_list = n_list <= ACT_CAP;
                }
            }
        }
        uint8_t *dst = (uint8_t *)sig_out + row * g0;
        if (sigbits == 32) hipLaunchKernelGGL(k_prob_write<uint32_t>, dim3(c->n_cu * 4), dim3(256), 0, c->stream, q.as<uint64_t>(), sig.as<uint64_t>(), ng * (uint64_t)m, (uint32_t *)dst);
        else hipLaunchKernelGGL(k_prob_write<uint64_t>, dim3(c->n_cu * 4), dim3(256), 0, c->stream, q.as<uint64_t>(), sig.as<uint64_t>(), ng * (uint64_t)m, (uint64_t *)dst);
        GS_HIP_CHECK(hipGetLastError());
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        g0 += ng;
    }
    return GS_OK;
}

// everything on the device; scratch owned by the call (freed after the stream drains)

// ======================================================================================================
// hll = SetSketch1 with u16 registers (SPEC 3.4; kmerutils HyperLogLogSketch<Kmer,u16> over probminhash's SetSketcher,
// /root/reference/src/dna/dnasketch.rs:541-574, src/aa/aasketch.rs:481-500). Registers are per-slot MAXIMA over (element, j) of
// k_j = trunc(1 - log_b x_j): order free, so the sequential lower-bound pruning of the paper becomes two passes per genome inside
// one workgroup (register table in LDS, ds_max_u32):
//   pass A  level j = 0 of every k-mer (one exponential draw, one uniform slot) under the workgroup's RUNNING minimum register as a
//           filter - a conservative integer threshold on the 52 uniform bits decides without evaluating a logarithm for all but a
//           few percent of the k-mers (U > a*m*b^-(K-2) implies k_0 < K because -ln(1-U) >= U; table ucut[K] from the host);
//   pass B  with the exact minimum K_low of pass A as the bound every k-mer is hashed again, the ~3 % whose first point beats K_low
//           walk on (j = 1, 2, ...: exponential spacings through a lazily materialised Fisher-Yates permutation) until k_j <= K_low.
// The permutation of a walking lane is a short list of displaced positions in registers (walks are a few steps once K_low is
// realistic); a genome whose walks outgrow it - few k-mers per register, K_low ~ 0, walks of ~m steps - is flagged and redone by
// the COLD instantiation, whose lanes keep stamped perm arrays in global scratch (same scheme as k_smh_cold_wg).
// ======================================================================================================
constexpr int HL_T = 512;          // lanes per workgroup
constexpr int HL_JR = 12;          // displaced positions a warm lane can remember
constexpr int HL_CT = 256;         // lanes per workgroup of the cold instantiation
// GTAB (register files beyond the LDS: sketch_size > ~40 000, the reference takes sketch_size as is, dnasketch.rs:541-574): the u32
// registers live in a per-workgroup global table and LDS holds a 2-byte FILTER per register - a value that is never above the
// register (the last k this workgroup sent there, refreshed from the table between chunks). A k that does not exceed the filter
// cannot raise the register and stops after one LDS read; the rest go to atomicMax in memory. Racy plain 16-bit stores can only leave
// the filter lower, i.e. more conservative: exact. Beyond ~80 000 registers the filter does not fit either (filt == nullptr) and every
// k above K_low goes to memory.
__device__ __forceinline__ uint32_t hll_xthr_bits(uint32_t klow, double inv_lnb)
{
    // k(x) = trunc(1 - ln x / ln b) <= klow  <=  x > b^-klow; the margin (2e-5 relative) covers the float rounding and spec_ln's last bits by far
    return __float_as_uint((float)(exp(-(double)klow / inv_lnb) * (1.0 + 2e-5)));
}
struct HllShared { uint32_t *tab; volatile uint32_t *ctl; uint16_t *filt; };   // ctl: [0] klow, [1] work item, [2] scratch min, [3] overflow flag, [4,5] ucut[klow], [7] float bits of a value safely ABOVE b^-klow (x >= it => k(x) <= klow without the logarithm)
template <bool COLD, bool GTAB>
struct HllEmit {
    HllShared S; uint32_t m; uint64_t zone_m; double inv_lnb, am;
    uint32_t *q, *perm; uint32_t *stamp; bool walk;
    // warm instantiation: per-wave LDS queue of the element hashes that pass the cheap cut (round 3, the MinEmitF scheme). ~3 % of the k-mers
    // go on to the logarithm / walk; left in place, nearly every wave iteration had a lane or two in that branch and all 64 paid for it
    // (rocprofv3: 733 VALU wave-instructions per 64 k-mers, 8x the OPH sketch). Queued, the slow part runs on full wavefronts.
    uint64_t *sq; uint32_t lane; mutable uint32_t qn; mutable uint64_t cutr;
    // Round 4: pass A also RECORDS the hashes that pass its cut (ctl[6] counts them, `surv` holds up to surv_cap of them per workgroup). The cut only
    // ever tightens (registers only grow), so every k-mer pass B would look at is among them: pass B walks this list instead of hashing the whole
    // genome a second time (a list that overflowed falls back to the second walk).
    uint64_t *surv; uint32_t surv_cap;
    __device__ __forceinline__ void record_wave(uint64_t h, uint32_t cnt) const          // the first `cnt` lanes of the wave append their h
    {
        if (!surv || walk) return;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd((uint32_t *)&S.ctl[6], cnt);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < cnt && base + lane < surv_cap) surv[base + lane] = h;
    }
    __device__ __forceinline__ void refresh() const { cutr = ((uint64_t)S.ctl[5] << 32) | S.ctl[4]; }
    __device__ __forceinline__ void operator()(uint64_t v, uint64_t, uint64_t) const { process(elem_hash<ALGO_HLL, 64>(v), true); }
    __device__ __forceinline__ void full(uint64_t v) const
    {
        const uint64_t h = elem_hash<ALGO_HLL, 64>(v);
        const uint64_t s0 = splitmix_mix(h + GS_GAMMA), s3 = splitmix_mix(h + 4 * GS_GAMMA);       // the first output needs two of the four state words
        const bool pass = ((rotl64(s0 + s3, 23) + s0) >> 12) <= cutr;
        const uint64_t bal = __ballot(pass);
        if (bal) {
            if (pass) sq[qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u))] = h;
            qn += (uint32_t)__popcll(bal);
            if (qn >= 64) {
                const uint64_t hq = sq[lane];
                record_wave(hq, 64u);
                process(hq, false);
                const uint32_t rest = qn - 64;
                const uint64_t mv = sq[64 + lane];
                if (lane < rest) sq[lane] = mv;
                qn = rest;
            }
        }
    }
    __device__ __forceinline__ void finish() const
    {
        if (sq && qn) { const uint64_t hq = lane < qn ? sq[lane] : 0; record_wave(hq, qn); if (lane < qn) process(hq, false); }
        qn = 0;
    }
    __device__ __forceinline__ void process(uint64_t h, bool record) const
    {
        Rng g; g.seed(h);
        const uint64_t u52 = g.next64() >> 12;
        const uint64_t cut = ((uint64_t)S.ctl[5] << 32) | S.ctl[4];
        if (u52 > cut) return;                                    // its first point cannot reach the lower bound: nor can any later one
        if (record && surv && !walk) { const uint32_t at = atomicAdd((uint32_t *)&S.ctl[6], 1u); if (at < surv_cap) surv[at] = h; }      // (lanes outside full waves: record boundaries)
        const uint32_t klow = S.ctl[0];
        double x = -spec_ln(1.0 - (double)u52 * 0x1.0p-52) / am;
        uint32_t k = hll_k(x, inv_lnb);
        if (k <= klow) return;
        uint32_t t = (uint32_t)rng_uint(g, (uint64_t)m, zone_m);
        reg_max(t, k);                                            // j = 0: p[0] after swap(p[0], p[t]) is t
        if (!walk) return;
        // j >= 1: lazily materialised permutation. Position 0 now holds t and position t holds 0.
        uint32_t tpos[HL_JR], tval[HL_JR]; uint32_t nt = 0;
        const uint32_t st = COLD ? (*stamp)++ : 0, l = threadIdx.x;
        if (COLD) { q[(uint64_t)0 * HL_CT + l] = st; perm[(uint64_t)0 * HL_CT + l] = t; q[(uint64_t)t * HL_CT + l] = st; perm[(uint64_t)t * HL_CT + l] = 0; }
        else if (t != 0) { tpos[0] = t; tval[0] = 0; nt = 1; }
        const double xthr = (double)__uint_as_float(S.ctl[7]);
        for (uint32_t j = 1; j < m; j++) {
            const double uj = g.u64f();
            const double den = GS_HLL_A * (double)(m - j);
            // (round 5) -ln(1 - u) >= u: the next point lies at or beyond x + u / den; when already that is safely past b^-klow the walk ends here, without
            // the two logarithms (the spacing and k of the point) - the case for ~94 % of the walkers' second points. Nothing reads the generator afterwards.
            if (x + uj / den >= xthr) break;
            const double te = -spec_ln(1.0 - uj);
            x = x + te / den;
            k = hll_k(x, inv_lnb);
            if (k <= klow) break;
            const uint64_t range = (uint64_t)(m - j);
            t = j + (uint32_t)rng_uint(g, range, uint_zone(range));
            uint32_t sl;
            if (COLD) {
                const uint64_t ij = (uint64_t)j * HL_CT + l, it = (uint64_t)t * HL_CT + l;
                if (q[ij] != st) { q[ij] = st; perm[ij] = j; }
                if (q[it] != st) { q[it] = st; perm[it] = t; }
                const uint32_t tmp = perm[ij]; perm[ij] = perm[it]; perm[it] = tmp;
                sl = perm[ij];
            } else {
                uint32_t pj = j, pt = t; int hit = -1;
#pragma unroll
                for (int i = 0; i < HL_JR; i++) if ((uint32_t)i < nt) { if (tpos[i] == j) pj = tval[i]; if (tpos[i] == t) { pt = tval[i]; hit = i; } }
                if (t != j) {
                    if (hit >= 0) {
#pragma unroll
                        for (int i = 0; i < HL_JR; i++) if (i == hit) tval[i] = pj;
                    } else {
                        if (nt == HL_JR) { S.ctl[3] = 1; return; }       // outgrown: the genome is redone by the cold instantiation
#pragma unroll
                        for (int i = 0; i < HL_JR; i++) if ((uint32_t)i == nt) { tpos[i] = t; tval[i] = pj; }
                        nt++;
                    }
                }
                sl = (t != j) ? pt : pj;
            }
            reg_max(sl, k);
        }
    }
    __device__ __forceinline__ void reg_max(uint32_t t, uint32_t k) const
    {
        if (!GTAB) { atomicMax(&S.tab[t], k); return; }
        if (S.filt) { if (k <= S.filt[t]) return; S.filt[t] = (uint16_t)k; }      // k <= q + 1 = 65535
        atomicMax(&S.tab[t], k);
    }
};
template <bool GTAB> __device__ __forceinline__ void emit_full_wave(const HllEmit<false, GTAB> &e, uint64_t v, uint64_t r, uint64_t p) { if (e.sq) e.full(v); else e(v, r, p); }
template <bool GTAB> __device__ __forceinline__ void emit_finish(const HllEmit<false, GTAB> &e) { e.finish(); }
// TIGHT (warm form only): 128 VGPRs at most. The 72 kB register table lets two workgroups share a CU, the 160 registers the compiler takes by itself (the
// walker's double-precision logarithms) let only ONE: 2 waves per SIMD, VALU issue 55 % busy, 40 % of the wave cycles waiting (profiles/r05_hll_pmc.txt).
// Capped, 31 registers of the walker spill and two workgroups are resident: 512 genomes of 1 / 2 / 3 / 5 / 20 Mbp in 6.7 / 7.0 / 8.3 / 9.8 / 25.2 ms against
// 8.0 / 8.4 / 9.4 / 11.9 / 30.6 (5 Mbp: 2.1 -> 2.6e11 k-mers/s, 3.0e11 at 2048 genomes). The uncapped instantiation stays for batches of very short
// genomes (< 16 k-mers per register: the walker is most of the work there) and for A/B (GS_HLL_TIGHT=0).
template <bool AA, bool COLD, bool GTAB, bool TIGHT = false>
__global__ __launch_bounds__(COLD ? HL_CT : HL_T, (!COLD && TIGHT) ? 4 : 1) void k_sketch_hll(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start,
        const uint64_t *__restrict__ rec_len, const uint64_t *__restrict__ rec_upre, const uint64_t *__restrict__ genome_rec_off,
        const uint64_t *__restrict__ gen_units, const uint32_t *__restrict__ list, uint32_t n_items, uint32_t k, uint32_t m, double inv_lnb,
        const uint64_t *__restrict__ ucut, uint32_t *__restrict__ lane_q, uint32_t *__restrict__ lane_perm, unsigned long long *__restrict__ counter,
        uint8_t *__restrict__ cold_flag, uint16_t *__restrict__ sig, uint32_t *__restrict__ gtab, int use_filter, uint32_t queue_off,
        uint64_t *__restrict__ surv_all, uint32_t surv_cap, uint32_t surv_min_chunks, float spec_c)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_hll[];
    HllShared S;
    if (!GTAB) { S.tab = (uint32_t *)s_hll; S.ctl = (volatile uint32_t *)(S.tab + m); S.filt = nullptr; }
    else { S.ctl = (volatile uint32_t *)s_hll; S.filt = use_filter ? (uint16_t *)(s_hll + 32) : nullptr; S.tab = gtab + (uint64_t)blockIdx.x * m; }
    const uint32_t T = COLD ? HL_CT : HL_T;
    uint32_t *q = COLD ? lane_q + (uint64_t)blockIdx.x * m * HL_CT : nullptr, *perm = COLD ? lane_perm + (uint64_t)blockIdx.x * m * HL_CT : nullptr;
    uint32_t stamp = 0;
    const uint64_t zone_m = uint_zone(m);
    const double am = GS_HLL_A * (double)m;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.ctl[1] = (uint32_t)atomicAdd(counter, 1ull);
        __syncthreads();
        const uint32_t c = S.ctl[1];
        if (c >= n_items) break;
        const uint64_t g = list ? list[c] : c;
        for (uint32_t i = threadIdx.x; i < m; i += T) { S.tab[i] = 0; if (GTAB && S.filt) S.filt[i] = 0; }
        if (threadIdx.x == 0) { S.ctl[0] = 0; S.ctl[3] = 0; S.ctl[4] = 0xFFFFFFFFu; S.ctl[5] = 0xFFFFFFFFu; S.ctl[6] = 0; S.ctl[7] = hll_xthr_bits(0, inv_lnb); }
        __syncthreads();
        const uint64_t r0 = genome_rec_off[g], r1 = genome_rec_off[g + 1], units = gen_units[g];
        const uint32_t nchunks = (uint32_t)std::max<uint64_t>(1, units / ((uint64_t)T * 8));
        bool outgrown = false;                                    // workgroup-uniform: some warm lane's walk outgrew its registers
        // Round 4: with survivor lists the genome is hashed ~1.1 times instead of twice: a WARM-UP over the first eighth of the chunks (pass -1: registers
        // only) gives pass A a realistic bound from its first k-mer on, so the hashes it records (the ones that pass its running cut: ~5 % then, a third
        // of the genome without the warm-up) fit the list, and pass B reads the list. Genomes of fewer than four chunks keep the two plain passes.
        // (measured, 512 genomes, m = 18000: 20 Mbp 67.0 -> 47.9 ms; 5 Mbp 23.7 -> 24.4 ms - there the time is the survivors' logarithms and walks, not
        // the filtering the lists save; 1 Mbp 12.0 -> 17.4 ms. Lists from `surv_min_chunks` chunks of 131 072 k-mers on: 64 = 8.4 Mbp.)
        uint64_t *surv_wg = (!COLD && surv_all && nchunks >= surv_min_chunks) ? surv_all + (uint64_t)blockIdx.x * surv_cap : nullptr;
        const uint32_t nwarm = nchunks / 8 ? nchunks / 8 : 1;
        // the minimum register of the workgroup's table -> S.ctl[2] (all lanes return it)
        auto min_register = [&]() -> uint32_t {
            __syncthreads();
            if (threadIdx.x == 0) S.ctl[2] = 0xFFFFFFFFu;
            __syncthreads();
            uint32_t lo = 0xFFFFFFFFu;
            for (uint32_t i = threadIdx.x; i < m; i += T) {
                // (a global table is only ever written by atomics performed in the L2: read it there too, not through the vector L1)
                const uint32_t r = GTAB ? __hip_atomic_load(&S.tab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : S.tab[i];
                if (GTAB && S.filt) S.filt[i] = (uint16_t)r;
                lo = min(lo, r);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) lo = min(lo, (uint32_t)__shfl_down((int)lo, o));
            if ((threadIdx.x & 63) == 0) atomicMin((uint32_t *)&S.ctl[2], lo);
            __syncthreads();
            return S.ctl[2];
        };
        // Round 5, ONE speculative pass instead of two (the bound of k_sketch_min's filtered emitter): every register ends as the maximum of
        // k = trunc(1 - log_b x) over one Exp(a)-distributed point per distinct k-mer, so the smallest register of N k-mers sits near
        // K = 1 - log_b((ln m + G) / (a N)), G Gumbel. With a guess K_g taken at G = c the genome is walked ONCE - level 0 and the walks of the ~m (ln m + c) / N
        // k-mers whose first point can beat K_g, everything at or below K_g dropped - and the guess is CHECKED: if no register ended below K_g, nothing
        // that was dropped could have raised one (max semantics) and the table is the exact one; otherwise (c = 7: ~1 genome in 1000; genomes of many
        // repeated k-mers) the table is kept - registers only grow - and the two exact passes below run over it.
        bool spec_done = false;
        if (!COLD && spec_c != 0.0f && units > 0) {
            const float xg = (__logf((float)m) + spec_c) / ((float)GS_HLL_A * (float)units * 32.0f);
            uint32_t kg = 0;
            if (xg > 0.0f && xg < 1.0f) { const float y = 1.0f - __logf(xg) * (float)inv_lnb; kg = y >= 4.0f ? (uint32_t)fminf(y, (float)GS_HLL_Q) - 1u : 0u; }
            if (kg >= 2) {
                if (threadIdx.x == 0) { const uint64_t cu = ucut[kg]; S.ctl[0] = kg; S.ctl[4] = (uint32_t)cu; S.ctl[5] = (uint32_t)(cu >> 32); S.ctl[7] = hll_xthr_bits(kg, inv_lnb); }
                __syncthreads();
                uint64_t *sq = queue_off ? (uint64_t *)(s_hll + queue_off) + (threadIdx.x >> 6) * 128 : nullptr;
                HllEmit<COLD, GTAB> emit{S, m, zone_m, inv_lnb, am, q, perm, &stamp, true, sq, threadIdx.x & 63, 0u, ~(uint64_t)0, nullptr, 0u};
                for (uint32_t ch = 0; ch < nchunks; ch++) {
                    emit.refresh();
                    walk_genome<AA>(seq, rec_start, rec_len, rec_upre, r0, r1, units, k, ch, nchunks, emit);
                    if (__syncthreads_or(S.ctl[3] != 0)) { outgrown = true; break; }
                }
                if (!outgrown) {
                    if (min_register() >= kg) spec_done = true;
                    else {
                        if (threadIdx.x == 0) { S.ctl[0] = 0; S.ctl[4] = 0xFFFFFFFFu; S.ctl[5] = 0xFFFFFFFFu; S.ctl[7] = hll_xthr_bits(0, inv_lnb); }
                        __syncthreads();
                    }
                }
            }
        }
        for (int pass = surv_wg ? -1 : 0; pass < 2 && !outgrown && !spec_done; pass++) {
            // (queue_off != 0: the warm instantiation has room in LDS for its survivor queues, 128 hashes per wave)
            uint64_t *sq = (!COLD && queue_off) ? (uint64_t *)(s_hll + queue_off) + (threadIdx.x >> 6) * 128 : nullptr;
            uint64_t *surv = pass == 0 ? surv_wg : nullptr;
            HllEmit<COLD, GTAB> emit{S, m, zone_m, inv_lnb, am, q, perm, &stamp, pass == 1, sq, threadIdx.x & 63, 0u, ~(uint64_t)0, surv, surv_cap};
            if (!COLD && pass == 1 && surv_wg && S.ctl[6] <= surv_cap) {
                // pass B over the recorded survivors of pass A (S.ctl[6] is stable here: pass A ended with barriers)
                const uint32_t nl = S.ctl[6];
                for (uint32_t i = threadIdx.x; i < nl; i += T) emit.process(surv_wg[i], false);
                if (__syncthreads_or(S.ctl[3] != 0)) outgrown = true;
                break;
            }
            for (uint32_t ch = 0; ch < (pass < 0 ? nwarm : nchunks); ch++) {
                emit.refresh();
                walk_genome<AA>(seq, rec_start, rec_len, rec_upre, r0, r1, units, k, ch, nchunks, emit);
                if (pass <= 0 || ch + 1 == nchunks) {
                    // refresh the lower bound: minimum register (pass A: running, between chunks; after pass A: exact)
                    const uint32_t kl_ = min_register();
                    if (threadIdx.x == 0) { const uint32_t kl = kl_; const uint64_t cu = ucut[kl]; S.ctl[0] = kl; S.ctl[4] = (uint32_t)cu; S.ctl[5] = (uint32_t)(cu >> 32); S.ctl[7] = hll_xthr_bits(kl, inv_lnb); }
                    __syncthreads();
                }
                // the overflow flag is raised by single lanes in mid-walk: every wave must take the SAME decision here (a wave that read
                // it a moment earlier than the others would leave the loop alone and pair its barriers with the wrong ones), so the flag
                // goes through a barrier-wide OR
                if (!COLD && __syncthreads_or(S.ctl[3] != 0)) { outgrown = true; break; }
            }
        }
        __syncthreads();
        if (!COLD && outgrown) { if (threadIdx.x == 0) cold_flag[g] = 1; continue; }
        for (uint32_t i = threadIdx.x; i < m; i += T)
            sig[g * (uint64_t)m + i] = (uint16_t)(GTAB ? __hip_atomic_load(&S.tab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : S.tab[i]);
    }
}

static int run_hll(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, const uint64_t *rec_start, const uint64_t *rec_len,
                   const uint64_t *rec_upre, const uint64_t *genome_rec_off, const uint64_t *gen_units, uint64_t n_genomes, uint64_t total_units, uint16_t *sig_out)
{
    const uint32_t m = p->sketch_size;
    const bool aa = p->data_t == GS_DATA_AA;
    const double inv_lnb = 1.0 / spec_ln(GS_HLL_B);
    // conservative skip thresholds on the 52 uniform bits, one per possible lower bound K (see the kernel comment)
    std::vector<uint64_t> ucut((size_t)GS_HLL_Q + 2);
    for (uint32_t K = 0; K <= GS_HLL_Q + 1; K++) {
        double f = K < 2 ? 1.0 : GS_HLL_A * (double)m * pow(GS_HLL_B, -(double)(K - 2));
        if (!(f < 1.0)) f = 1.0;
        ucut[K] = (uint64_t)floor(f * 0x1.0p52);                  // u52 < 2^52 always: f = 1 never skips
    }
    PoolBuf dcut(c, 24), cold(c, 25), cnt(c, 38);
    int rc;
    if ((rc = dcut.alloc(8 * ucut.size()))) return rc;
    if ((rc = cold.alloc(n_genomes))) return rc;
    if ((rc = cnt.alloc(64))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(dcut.p, ucut.data(), 8 * ucut.size(), hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemsetAsync(cold.p, 0, n_genomes, c->stream));
    GS_HIP_CHECK(hipMemsetAsync(cnt.p, 0, 8, c->stream));
    // register file: u32 in LDS while it fits; beyond that a per-workgroup global table behind a 2-byte LDS filter (while THAT fits)
    const size_t lds_cap = 160 * 1024 - 256;
    const bool gtab = 4 * ((size_t)m + 8) > lds_cap;
    const int use_filter = gtab && 32 + 2 * (size_t)m <= lds_cap;
    const size_t lds0 = !gtab ? 4 * ((size_t)m + 8) : (use_filter ? 32 + 2 * (size_t)m : 32);
    // survivor queues of the warm kernel (8 kB): only where they do not cost a resident workgroup
    const size_t lds_q = ((lds0 + 15) & ~(size_t)15) + (size_t)(HL_T / 64) * 128 * 8, half_cu = (160 * 1024) / 2 / 1280 * 1280;
    const bool use_q = !(getenv("GS_HLL_QUEUE") && !atoi(getenv("GS_HLL_QUEUE"))) && (lds_q <= half_cu || (lds0 > half_cu && lds_q <= lds_cap));
    const uint32_t queue_off = use_q ? (uint32_t)((lds0 + 15) & ~(size_t)15) : 0u;
    const size_t lds = use_q ? lds_q : lds0;
    const uint32_t wgs = (uint32_t)std::min<uint64_t>(n_genomes, (uint64_t)c->n_cu * 2);
    PoolBuf gt(c, 19), sv(c, 46);
    if (gtab && (rc = gt.alloc((size_t)4 * m * std::max<uint32_t>(wgs, (uint32_t)c->n_cu)))) return rc;
    // survivor lists of pass A: 2^20 hashes (8 MB) per workgroup - ~8 % of the k-mers of a 5 Mbp genome survive the running cut; a longer genome
    // overflows its list and takes the second walk as before (GS_HLL_SURVIVORS=n: lists of n hashes, 0 = always the second walk)
    uint32_t surv_cap = getenv("GS_HLL_SURVIVORS") ? (uint32_t)std::max(0, std::min(1 << 24, atoi(getenv("GS_HLL_SURVIVORS")))) : (1u << 20);
    // GS_HLL_SPEC: the c of the speculative single pass (k_sketch_hll; default 8, 0 = the two exact passes only)
    const float spec_c = getenv("GS_HLL_SPEC") ? (float)atof(getenv("GS_HLL_SPEC")) : 8.0f;   // (one genome per workgroup and launch tail: a failed guess doubles that workgroup's time)
    const uint32_t surv_minch = getenv("GS_HLL_SURVIVORS_MINCHUNKS") ? (uint32_t)std::max(4, atoi(getenv("GS_HLL_SURVIVORS_MINCHUNKS"))) : 64u;
    //  the lists are only allocated when some genome of the batch can reach `surv_minch` chunks - no genome holds more units than the whole
    // batch does -, and a batch whose lists do not fit takes the plain second walk instead of failing the call
    if (total_units / ((uint64_t)HL_T * 8) < surv_minch) surv_cap = 0;
    if (surv_cap && sv.alloc((size_t)8 * surv_cap * wgs) != GS_OK) { (void)hipGetLastError(); surv_cap = 0; }
    {
        ProfScope ps(c, FAM_SKETCH);
    // (GS_HLL_TIGHT=0/1 overrides the choice for A/B)
    const double per_reg = 32.0 * (double)total_units / (double)std::max<uint64_t>(n_genomes, 1) / (double)m;       // (a unit is 32 symbols)
    const bool tight = getenv("GS_HLL_TIGHT") ? atoi(getenv("GS_HLL_TIGHT")) != 0 : per_reg >= 16.0;       // (measured from 55 k-mers per register up: always)
#define GS_LAUNCH_HLL(AAV, GV) do { if (tight) GS_LAUNCH_HLL_T(AAV, GV, true); else GS_LAUNCH_HLL_T(AAV, GV, false); } while (0)
#define GS_LAUNCH_HLL_T(AAV, GV, TV)                                                                                           \
    do {                                                                                                                       \
        auto kern = k_sketch_hll<AAV, false, GV, TV>;                                                                          \
        if (lds > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(HL_T), lds, c->stream, seq, rec_start, rec_len, rec_upre, genome_rec_off, gen_units, (const uint32_t *)nullptr, \
                           (uint32_t)n_genomes, kq_of(p), m, inv_lnb, dcut.as<uint64_t>(), (uint32_t *)nullptr, (uint32_t *)nullptr, cnt.as<unsigned long long>(), cold.as<uint8_t>(), sig_out, \
                           gt.as<uint32_t>(), use_filter, queue_off, surv_cap ? sv.as<uint64_t>() : (uint64_t *)nullptr, surv_cap, surv_minch, spec_c); \
    } while (0)
        if (aa) { if (gtab) GS_LAUNCH_HLL(true, true); else GS_LAUNCH_HLL(true, false); }
        else { if (gtab) GS_LAUNCH_HLL(false, true); else GS_LAUNCH_HLL(false, false); }
#undef GS_LAUNCH_HLL
#undef GS_LAUNCH_HLL_T
    }
    GS_HIP_CHECK(hipGetLastError());
    std::vector<uint8_t> h(n_genomes);
    GS_HIP_CHECK(hipMemcpyAsync(h.data(), cold.p, n_genomes, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    std::vector<uint32_t> list;
    for (uint64_t g = 0; g < n_genomes; g++) if (h[g]) list.push_back((uint32_t)g);
    if (list.empty()) return GS_OK;
    // cold genomes (few k-mers per register): exact walks with the permutation in per-lane global scratch
    const uint32_t nc = (uint32_t)list.size();
    // one workgroup needs 2 x 4 x m x HL_CT bytes of per-lane scratch (205 MB at m = 100 000): as many workgroups as a quarter of the free memory
    // (at most 16 GB) holds - they pull the cold genomes from a shared counter, fewer workgroups only take more turns
    uint32_t cw = std::min<uint32_t>(nc, (uint32_t)c->n_cu);
    {
        size_t fr = 0, tot = 0;
        uint64_t budget = (uint64_t)4 << 30;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget = std::min<uint64_t>((uint64_t)16 << 30, std::max<uint64_t>((uint64_t)fr / 4, (uint64_t)64 << 20));
        cw = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cw, budget / ((uint64_t)8 * m * HL_CT)));
    }
    PoolBuf dl(c, 26), lq(c, 27), lp(c, 37);
    if ((rc = dl.alloc(4 * (size_t)nc))) return rc;
    if ((rc = lq.alloc((size_t)4 * cw * m * HL_CT))) return rc;
    if ((rc = lp.alloc((size_t)4 * cw * m * HL_CT))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(dl.p, list.data(), 4 * (size_t)nc, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemsetAsync(lq.p, 0xFF, (size_t)4 * cw * m * HL_CT, c->stream));
    GS_HIP_CHECK(hipMemsetAsync(cnt.p, 0, 8, c->stream));
#define GS_LAUNCH_HLLC(AAV, GV)                                                                                                \
    do {                                                                                                                       \
        auto kern = k_sketch_hll<AAV, true, GV>;                                                                               \
        if (lds > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, dim3(cw), dim3(HL_CT), lds, c->stream, seq, rec_start, rec_len, rec_upre, genome_rec_off, gen_units, dl.as<uint32_t>(), nc, kq_of(p), m, \
                           inv_lnb, dcut.as<uint64_t>(), lq.as<uint32_t>(), lp.as<uint32_t>(), cnt.as<unsigned long long>(), cold.as<uint8_t>(), sig_out, \
                           gt.as<uint32_t>(), use_filter, 0u, (uint64_t *)nullptr, 0u, 0u, 0.0f);                              \
    } while (0)
    if (aa) { if (gtab) GS_LAUNCH_HLLC(true, true); else GS_LAUNCH_HLLC(true, false); }
    else { if (gtab) GS_LAUNCH_HLLC(false, true); else GS_LAUNCH_HLLC(false, false); }
#undef GS_LAUNCH_HLLC
    GS_HIP_CHECK(hipGetLastError());
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

int sketch_dev_impl(gs_ctx *c, const gs_sketch_params *p, const void *seq, uint64_t seq_bytes, const uint64_t *rec_start,
                    const uint64_t *rec_len, uint64_t n_rec, const uint64_t *genome_rec_off, uint64_t n_genomes, void *sig_out, bool sync_at_end)
{
    int rc = gs_check_params(p);
    if (rc) return rc;
    GS_REQUIRE(c && (n_genomes == 0 || (seq && rec_start && rec_len && genome_rec_off && sig_out)), GS_ERR_INVALID, "null argument");
    if (n_genomes == 0) return GS_OK;
    GS_REQUIRE(n_genomes < ((uint64_t)1 << 31), GS_ERR_INVALID, "too many genomes in one batch");
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t m = p->sketch_size;
    if (p->algo == GS_ALGO_OPTDENS || p->algo == GS_ALGO_REVOPTDENS) {
        PoolBuf upre(c, 20), gunits(c, 21), table(c, 22), win(c, 23);
        rc = upre.alloc(8 * (n_rec + 1)); if (rc) return rc;
        rc = gunits.alloc(8 * n_genomes); if (rc) return rc;
        rc = table.alloc((size_t)n_genomes * m * 4); if (rc) return rc;
        if (p->algo == GS_ALGO_REVOPTDENS) { rc = win.alloc((size_t)n_genomes * m * 4); if (rc) return rc; }
        hipLaunchKernelGGL(k_unit_prefix, dim3((uint32_t)((n_genomes + 3) / 4)), dim3(256), 0, c->stream, rec_start, rec_len,
                           genome_rec_off, n_genomes, p->k, upre.as<uint64_t>(), gunits.as<uint64_t>());
        GS_HIP_CHECK(hipGetLastError());
        uint64_t avg_units = (p->data_t == GS_DATA_AA ? seq_bytes / 32 : seq_bytes / 8) / n_genomes + 1;
        rc = launch_oph(c, p, (const uint8_t *)seq, rec_start, rec_len, upre.as<uint64_t>(), genome_rec_off, gunits.as<uint64_t>(),
                        n_genomes, avg_units, table.as<uint32_t>(), win.as<uint32_t>(), (float *)sig_out);
        if (rc) return rc;
        // (the scratch slots are only reused by later calls on this context, i.e. behind these kernels on its stream: a caller that pipelines
        // several sketches - gs_index_sketch_and_search_dev - may leave the wait to its own events)
        if (sync_at_end) GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        return GS_OK;
    }
    if (p->algo == GS_ALGO_SUPER || p->algo == GS_ALGO_SUPER2) {
        PoolBuf upre(c, 20), gunits(c, 21);
        rc = upre.alloc(8 * (n_rec + 1)); if (rc) return rc;
        rc = gunits.alloc(8 * n_genomes); if (rc) return rc;
        hipLaunchKernelGGL(k_unit_prefix, dim3((uint32_t)((n_genomes + 3) / 4)), dim3(256), 0, c->stream, rec_start, rec_len,
                           genome_rec_off, n_genomes, p->k, upre.as<uint64_t>(), gunits.as<uint64_t>());
        GS_HIP_CHECK(hipGetLastError());
        const uint64_t avg_units = (p->data_t == GS_DATA_AA ? seq_bytes / 32 : seq_bytes / 8) / n_genomes + 1;
        const uint8_t *sq = (const uint8_t *)seq;
        const uint64_t *up = upre.as<uint64_t>(), *gu = gunits.as<uint64_t>();
        const int vb = gs_value_bits(p);
        if (p->algo == GS_ALGO_SUPER) rc = run_smh<ALGO_SUPER, 64, uint32_t>(c, p, sq, rec_start, rec_len, up, genome_rec_off, gu, n_genomes, avg_units, sig_out);
        else if (vb == 32) rc = run_smh<ALGO_SUPER2, 32, uint32_t>(c, p, sq, rec_start, rec_len, up, genome_rec_off, gu, n_genomes, avg_units, sig_out);
        else rc = run_smh<ALGO_SUPER2, 64, uint64_t>(c, p, sq, rec_start, rec_len, up, genome_rec_off, gu, n_genomes, avg_units, sig_out);
        if (rc) return rc;
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        return GS_OK;
    }
    if (p->algo == GS_ALGO_HLL) {
        PoolBuf upre(c, 20), gunits(c, 21);
        rc = upre.alloc(8 * (n_rec + 1)); if (rc) return rc;
        rc = gunits.alloc(8 * n_genomes); if (rc) return rc;
        hipLaunchKernelGGL(k_unit_prefix, dim3((uint32_t)((n_genomes + 3) / 4)), dim3(256), 0, c->stream, rec_start, rec_len,
                           genome_rec_off, n_genomes, p->k, upre.as<uint64_t>(), gunits.as<uint64_t>());
        GS_HIP_CHECK(hipGetLastError());
        rc = run_hll(c, p, (const uint8_t *)seq, rec_start, rec_len, upre.as<uint64_t>(), genome_rec_off, gunits.as<uint64_t>(), n_genomes, (p->data_t == GS_DATA_AA ? seq_bytes / 32 : seq_bytes / 8) + n_rec + 1, (uint16_t *)sig_out);
        if (rc) return rc;
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        return GS_OK;
    }
    if (p->algo == GS_ALGO_PROB3A) {
        rc = run_prob(c, p, (const uint8_t *)seq, seq_bytes, rec_start, rec_len, n_rec, genome_rec_off, n_genomes, sig_out);
        if (rc) return rc;
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));
        return GS_OK;
    }
    GS_REQUIRE(false, GS_ERR_UNSUPPORTED, "sketch algo %u is not implemented on the device", p->algo);
}

}  // namespace gs

extern "C" {

int gs_sketch_batch_dev(gs_ctx *c, const gs_sketch_params *p, const void *seq_dev, uint64_t seq_bytes, const uint64_t *rec_start_dev,
                        const uint64_t *rec_len_dev, uint64_t n_rec, const uint64_t *genome_rec_off_dev, uint64_t n_genomes,
                        void *sig_out_dev)
{
    return gs::sketch_dev_impl(c, p, seq_dev, seq_bytes, rec_start_dev, rec_len_dev, n_rec, genome_rec_off_dev, n_genomes, sig_out_dev, true);
}

int gs_sketch_batch(gs_ctx *c, const gs_sketch_params *p, const void *seq, uint64_t seq_bytes, const uint64_t *rec_start,
                    const uint64_t *rec_len, uint64_t n_rec, const uint64_t *genome_rec_off, uint64_t n_genomes, void *sig_out)
{
    int rc = gs_check_params(p);
    if (rc) return rc;
    GS_REQUIRE(c && (n_genomes == 0 || (rec_start && rec_len && genome_rec_off && sig_out)), GS_ERR_INVALID, "null argument");
    if (n_genomes == 0) return GS_OK;
    GS_REQUIRE(seq || seq_bytes == 0, GS_ERR_INVALID, "null sequence buffer");
    // every record must lie inside the buffer
    const uint64_t sym_cap = p->data_t == GS_DATA_AA ? seq_bytes : seq_bytes * 4;
    for (uint64_t r = 0; r < n_rec; r++)
        GS_REQUIRE(rec_start[r] + rec_len[r] <= sym_cap, GS_ERR_INVALID, "record %llu exceeds the sequence buffer", (unsigned long long)r);
    GS_REQUIRE(genome_rec_off[n_genomes] <= n_rec, GS_ERR_INVALID, "genome_rec_off exceeds n_rec");
    // (a worker thread of the host runs on its own stream and scratch, see gs_internal.hpp; a device failure there is repeated on the main context)
    return gs::on_worker(c, [&](gs_ctx *c) -> int {
    int rc;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    // staging from the context's grow-only pool: a hipMalloc / hipFree pair per call costs more than a small sketch and, worse, hipFree waits for
    // every stream of the device - which would serialise the worker contexts of other host threads
    gs::PoolBuf dseq(c, 48), drs(c, 49), drl(c, 50), dgo(c, 51), dsig(c, 52);
    const uint64_t padded = gs::round_up(seq_bytes, 32) + 32;
    const size_t sigbytes = (size_t)n_genomes * p->sketch_size * gs_sig_elem_bytes(p);
    if ((rc = dseq.alloc(padded))) return rc;
    if ((rc = drs.alloc(8 * (n_rec + 1)))) return rc;
    if ((rc = drl.alloc(8 * (n_rec + 1)))) return rc;
    if ((rc = dgo.alloc(8 * (n_genomes + 1)))) return rc;
    if ((rc = dsig.alloc(sigbytes))) return rc;
    GS_HIP_CHECK(hipMemsetAsync((uint8_t *)dseq.p + (padded - 64), 0, 64, c->stream));
    if (seq_bytes) GS_HIP_CHECK(hipMemcpyAsync(dseq.p, seq, seq_bytes, hipMemcpyHostToDevice, c->stream));
    if (n_rec) {
        GS_HIP_CHECK(hipMemcpyAsync(drs.p, rec_start, 8 * n_rec, hipMemcpyHostToDevice, c->stream));
        GS_HIP_CHECK(hipMemcpyAsync(drl.p, rec_len, 8 * n_rec, hipMemcpyHostToDevice, c->stream));
    }
    GS_HIP_CHECK(hipMemcpyAsync(dgo.p, genome_rec_off, 8 * (n_genomes + 1), hipMemcpyHostToDevice, c->stream));
    rc = gs::sketch_dev_impl(c, p, dseq.p, padded, drs.as<uint64_t>(), drl.as<uint64_t>(), n_rec, dgo.as<uint64_t>(), n_genomes, dsig.p, true);
    if (rc) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(sig_out, dsig.p, sigbytes, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
    });
}

}  // extern "C"
