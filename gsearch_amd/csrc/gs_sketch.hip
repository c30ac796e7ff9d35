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
#include <vector>
#include "gs_internal.hpp"
#include "gs_spec.hpp"

namespace gs {

#define GS_EMPTY32 0xFFFFFFFFu
static constexpr int SK_THREADS = 512;

__constant__ uint8_t c_aa_code[32] = {
    // index = ASCII & 31 : @ A B C D E F G H I J K L M N O P Q R S T U V W X Y Z ...
    0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 0, 8, 9, 10, 11, 0, 12, 13, 14, 15, 16, 0, 17, 18, 0, 19, 0, 0, 0, 0, 0, 0};

// per-record unit counts -> exclusive prefix inside each genome (one thread per genome; record lists are short)
__global__ void k_unit_prefix(const uint64_t *rec_start, const uint64_t *rec_len, const uint64_t *genome_rec_off,
                              uint64_t n_genomes, uint32_t k, uint64_t *rec_upre, uint64_t *gen_units)
{
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_genomes) return;
    uint64_t acc = 0;
    for (uint64_t r = genome_rec_off[g]; r < genome_rec_off[g + 1]; r++) {
        rec_upre[r] = acc;
        uint64_t len = rec_len[r];
        if (len >= k) { uint64_t s = rec_start[r]; acc += ((s + len - 1) >> 5) - (s >> 5) + 1; }
    }
    gen_units[g] = acc;
}

struct OphEmit {
    uint32_t *table; uint32_t m; uint64_t zone;
    __device__ __forceinline__ void operator()(uint64_t v) const
    {
        uint32_t r, b;
        oph_draw(fx64(v), m, zone, r, b);
        atomicMin(&table[b], r);
    }
};

// The streaming part shared by every sketcher: walk the units of genome g assigned to this workgroup and
// call emit(v) for each valid canonical k-mer value.
template <bool AA, class Emit>
__device__ __forceinline__ void walk_genome(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start,
                                            const uint64_t *__restrict__ rec_len, const uint64_t *__restrict__ rec_upre,
                                            uint64_t r0, uint64_t r1, uint64_t units, uint32_t k, uint32_t part,
                                            uint32_t parts, const Emit &emit)
{
    const uint64_t mask = AA ? (((uint64_t)1 << (5 * k)) - 1) : (k == 32 ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1));
    const uint32_t rcshift = 2 * (k - 1);
    for (uint64_t f = (uint64_t)part * blockDim.x + threadIdx.x; f < units; f += (uint64_t)parts * blockDim.x) {
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
                uint64_t p = __builtin_bswap64(w64[u - 1]) << (2 * (32 - (k - 1)));
                for (uint32_t j = 0; j + 1 < k; j++) {
                    uint64_t c = p >> 62; p <<= 2;
                    fwd = ((fwd << 2) | c) & mask;
                    rc = (rc >> 2) | ((3 - c) << rcshift);
                }
            }
#pragma unroll 2
            for (uint32_t j = 0; j < 32; j++) {
                uint64_t c = w >> 62; w <<= 2;
                fwd = ((fwd << 2) | c) & mask;
                rc = (rc >> 2) | ((3 - c) << rcshift);
                uint64_t a = a0 + j;
                if (a >= first_valid && a < re) emit((fwd < rc ? fwd : rc) & mask);
            }
        } else {
            const uint64_t *w64 = (const uint64_t *)seq;
            uint64_t val = 0;
            if (a0 > rb && k > 1) {
                // previous k-1 residues: bytes a0-(k-1) .. a0-1 (k-1 <= 11 -> inside the previous two 8-byte words)
                for (uint32_t j = 0; j + 1 < k; j++) {
                    uint64_t a = a0 - (k - 1) + j;
                    uint8_t ch = seq[a];
                    val = ((val << 5) | c_aa_code[ch & 31]) & mask;
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                uint64_t x = w64[u * 4 + q];
#pragma unroll 2
                for (uint32_t j = 0; j < 8; j++) {
                    uint32_t ch = (uint32_t)(x & 0xFF); x >>= 8;
                    val = ((val << 5) | c_aa_code[ch & 31]) & mask;
                    uint64_t a = a0 + q * 8 + j;
                    if (a >= first_valid && a < re) emit(val);
                }
            }
        }
    }
}

// ---- optdens / revoptdens main kernel (SPEC 3.1) --------------------------------------------------
template <bool AA, bool LDS_TABLE>
__global__ __launch_bounds__(SK_THREADS) void k_sketch_oph(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ rec_start,
                                                            const uint64_t *__restrict__ rec_len, const uint64_t *__restrict__ rec_upre,
                                                            const uint64_t *__restrict__ genome_rec_off, const uint64_t *__restrict__ gen_units,
                                                            uint32_t k, uint32_t m, uint64_t zone, uint32_t *__restrict__ table_out)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_table[];
    const uint64_t g = blockIdx.y;
    const uint32_t part = blockIdx.x, parts = gridDim.x;
    uint32_t *gtab = table_out + g * (uint64_t)m;
    uint32_t *table = LDS_TABLE ? s_table : gtab;
    if (LDS_TABLE) {
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) s_table[i] = GS_EMPTY32;
        __syncthreads();
    }
    OphEmit emit{table, m, zone};
    walk_genome<AA>(seq, rec_start, rec_len, rec_upre, genome_rec_off[g], genome_rec_off[g + 1], gen_units[g], k, part, parts, emit);
    if (LDS_TABLE) {
        __syncthreads();
        if (parts == 1) { for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) gtab[i] = s_table[i]; }
        else { for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) { uint32_t v = s_table[i]; if (v != GS_EMPTY32) atomicMin(&gtab[i], v); } }
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

static int launch_oph(gs_ctx *c, const gs_sketch_params *p, const uint8_t *seq, const uint64_t *rec_start, const uint64_t *rec_len,
                      const uint64_t *rec_upre, const uint64_t *genome_rec_off, const uint64_t *gen_units, uint64_t n_genomes,
                      uint64_t avg_units, uint32_t *table, uint32_t *win, float *sig)
{
    const uint32_t m = p->sketch_size;
    const uint64_t zone = uint_zone(m);
    const size_t lds = (size_t)m * 4;
    const bool use_lds = lds <= 160 * 1024 - 256;
    uint32_t parts = 1;
    if (n_genomes < (uint64_t)2 * c->n_cu) {
        parts = (uint32_t)((2 * (uint64_t)c->n_cu + n_genomes - 1) / n_genomes);
        uint64_t maxp = avg_units / SK_THREADS + 1;       // at least one full sweep per part
        if (parts > maxp) parts = (uint32_t)maxp;
        if (parts < 1) parts = 1;
    }
    if (!use_lds || parts > 1) GS_HIP_CHECK(hipMemsetAsync(table, 0xFF, (size_t)n_genomes * m * 4, c->stream));
    const bool aa = p->data_t == GS_DATA_AA;
    // grid.y is limited to 65535: chunk the genome dimension
    for (uint64_t g0 = 0; g0 < n_genomes; g0 += 65535) {
        uint64_t ng = n_genomes - g0 < 65535 ? n_genomes - g0 : 65535;
        dim3 grid(parts, (uint32_t)ng), block(SK_THREADS);
        const uint64_t *gro = genome_rec_off + g0; const uint64_t *gu = gen_units + g0;
        uint32_t *tab = table + g0 * m;
        ProfScope ps(c, FAM_SKETCH);
#define GS_LAUNCH_OPH(AAV, LDSV)                                                                                        \
    do {                                                                                                                \
        auto kern = k_sketch_oph<AAV, LDSV>;                                                                            \
        if (LDSV) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, block, LDSV ? lds : 0, c->stream, seq, rec_start, rec_len, rec_upre, gro, gu, p->k, m, zone, tab); \
    } while (0)
        if (aa) { if (use_lds) GS_LAUNCH_OPH(true, true); else GS_LAUNCH_OPH(true, false); }
        else    { if (use_lds) GS_LAUNCH_OPH(false, true); else GS_LAUNCH_OPH(false, false); }
#undef GS_LAUNCH_OPH
        GS_HIP_CHECK(hipGetLastError());
    }
    if (p->algo == GS_ALGO_OPTDENS)
        hipLaunchKernelGGL(k_oph_finish<ALGO_OPTDENS>, dim3((uint32_t)n_genomes), dim3(256), 0, c->stream, table, m, zone, win, sig);
    else
        hipLaunchKernelGGL(k_oph_finish<ALGO_REVOPTDENS>, dim3((uint32_t)n_genomes), dim3(256), 0, c->stream, table, m, zone, win, sig);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

// everything on the device; scratch owned by the call (freed after the stream drains)
static int sketch_dev_impl(gs_ctx *c, const gs_sketch_params *p, const void *seq, uint64_t seq_bytes, const uint64_t *rec_start,
                           const uint64_t *rec_len, uint64_t n_rec, const uint64_t *genome_rec_off, uint64_t n_genomes, void *sig_out)
{
    int rc = gs_check_params(p);
    if (rc) return rc;
    GS_REQUIRE(c && (n_genomes == 0 || (seq && rec_start && rec_len && genome_rec_off && sig_out)), GS_ERR_INVALID, "null argument");
    if (n_genomes == 0) return GS_OK;
    GS_REQUIRE(n_genomes < ((uint64_t)1 << 31), GS_ERR_INVALID, "too many genomes in one batch");
    GS_HIP_CHECK(hipSetDevice(c->device));
    const uint32_t m = p->sketch_size;
    if (p->algo == GS_ALGO_OPTDENS || p->algo == GS_ALGO_REVOPTDENS) {
        DevBuf upre, gunits, table, win;
        rc = upre.alloc(8 * (n_rec + 1)); if (rc) return rc;
        rc = gunits.alloc(8 * n_genomes); if (rc) return rc;
        rc = table.alloc((size_t)n_genomes * m * 4); if (rc) return rc;
        if (p->algo == GS_ALGO_REVOPTDENS) { rc = win.alloc((size_t)n_genomes * m * 4); if (rc) return rc; }
        hipLaunchKernelGGL(k_unit_prefix, dim3((uint32_t)((n_genomes + 255) / 256)), dim3(256), 0, c->stream, rec_start, rec_len,
                           genome_rec_off, n_genomes, p->k, upre.as<uint64_t>(), gunits.as<uint64_t>());
        GS_HIP_CHECK(hipGetLastError());
        uint64_t avg_units = (p->data_t == GS_DATA_AA ? seq_bytes / 32 : seq_bytes / 8) / n_genomes + 1;
        rc = launch_oph(c, p, (const uint8_t *)seq, rec_start, rec_len, upre.as<uint64_t>(), genome_rec_off, gunits.as<uint64_t>(),
                        n_genomes, avg_units, table.as<uint32_t>(), win.as<uint32_t>(), (float *)sig_out);
        if (rc) return rc;
        GS_HIP_CHECK(hipStreamSynchronize(c->stream));   // scratch lifetime
        return GS_OK;
    }
    GS_REQUIRE(false, GS_ERR_UNSUPPORTED, "sketch algo %u is not implemented on the device yet", p->algo);
}

}  // namespace gs

extern "C" {

int gs_sketch_batch_dev(gs_ctx *c, const gs_sketch_params *p, const void *seq_dev, uint64_t seq_bytes, const uint64_t *rec_start_dev,
                        const uint64_t *rec_len_dev, uint64_t n_rec, const uint64_t *genome_rec_off_dev, uint64_t n_genomes,
                        void *sig_out_dev)
{
    return gs::sketch_dev_impl(c, p, seq_dev, seq_bytes, rec_start_dev, rec_len_dev, n_rec, genome_rec_off_dev, n_genomes, sig_out_dev);
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
    GS_HIP_CHECK(hipSetDevice(c->device));
    gs::DevBuf dseq, drs, drl, dgo, dsig;
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
    rc = gs::sketch_dev_impl(c, p, dseq.p, padded, drs.as<uint64_t>(), drl.as<uint64_t>(), n_rec, dgo.as<uint64_t>(), n_genomes, dsig.p);
    if (rc) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(sig_out, dsig.p, sigbytes, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

}  // extern "C"
