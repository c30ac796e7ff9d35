// gs_ingest.hip — FASTA ingest on the host side of the boundary.
//
// Replaces, for already-decompressed text, what the reference does per file in
//   /root/reference/src/dna/dnafiles.rs:43-193 (needletail record iteration, `capsid` filter :67,
//   Sequence::encode_and_add :70-71,148-149 — non-ACGT dropped, case folded, one Sequence per record).
// The host only finds record boundaries (memchr-speed); the per-byte work — filtering, case folding and
// 2-bit packing — runs on the device: newlines are just more non-ACGT bytes, so the raw text of a record goes
// through one stream-compaction kernel (per-chunk valid counts -> offsets -> ballot-prefix pack).
#include <string.h>
#include <vector>
#include "gs_internal.hpp"

namespace gs {

constexpr int PK_CHUNK = 4096;      // text bytes per workgroup
constexpr int PK_T = 256;

__device__ __forceinline__ int dna_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}
// chunk c of the batch = bytes [cb[c], ce[c]) of the text (never spans two records)
__global__ __launch_bounds__(PK_T) void k_pack_count(const uint8_t *__restrict__ text, const uint64_t *__restrict__ cb, const uint64_t *__restrict__ ce,
                                                      uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_n;
    const uint64_t c = blockIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t loc = 0;
    for (uint64_t i = cb[c] + threadIdx.x; i < ce[c]; i += PK_T) loc += dna_code(text[i]) >= 0;
    for (int o = 32; o > 0; o >>= 1) loc += __shfl_down(loc, o);
    if ((threadIdx.x & 63) == 0 && loc) atomicAdd(&s_n, loc);
    __syncthreads();
    if (threadIdx.x == 0) counts[c] = s_n;
}
// pack chunk c at base coordinate out_base[c]: codes are assembled into 32-bit words in LDS, interior words are stored,
// the (at most two) words shared with the neighbouring chunks are OR-ed atomically into the zeroed output
__global__ __launch_bounds__(PK_T) void k_pack_write(const uint8_t *__restrict__ text, const uint64_t *__restrict__ cb, const uint64_t *__restrict__ ce,
                                                      const uint64_t *__restrict__ out_base, uint32_t *__restrict__ packed)
{
    __shared__ uint32_t s_words[PK_CHUNK / 16 + 2];
    __shared__ uint32_t s_wave[PK_T / 64];
    __shared__ uint32_t s_run;
    const uint64_t c = blockIdx.x, b0 = cb[c], b1 = ce[c], ob = out_base[c];
    const uint32_t shift = (uint32_t)(ob & 15);                 // bases already present in the first output word
    for (uint32_t i = threadIdx.x; i < PK_CHUNK / 16 + 2; i += PK_T) s_words[i] = 0;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint64_t i0 = b0; i0 < b1; i0 += PK_T) {
        const uint64_t i = i0 + threadIdx.x;
        const int code = i < b1 ? dna_code(text[i]) : -1;
        const uint64_t bal = __ballot(code >= 0);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = s_run;
        for (uint32_t w = 0; w < wv; w++) off += s_wave[w];
        if (code >= 0) {
            const uint32_t idx = shift + off + (uint32_t)__popcll(bal & ((1ull << lane) - 1));   // base index relative to the first word
            // byte-major, first base in the top bits of its byte (SPEC 1.1): base b of a little-endian u32 word -> bits 8*(b/4) + 6-2*(b%4)
            const uint32_t b = idx & 15;
            atomicOr(&s_words[idx >> 4], (uint32_t)code << (8 * (b >> 2) + 6 - 2 * (b & 3)));
        }
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t w = 0; w < PK_T / 64; w++) t += s_wave[w]; s_run += t; }
        __syncthreads();
    }
    const uint32_t nbases = s_run;
    if (nbases == 0) return;
    const uint32_t nwords = (shift + nbases + 15) >> 4;
    uint32_t *dst = packed + (ob >> 4);
    for (uint32_t w = threadIdx.x; w < nwords; w += PK_T) {
        const uint32_t v = s_words[w];
        if (w == 0 || w == nwords - 1) { if (v) atomicOr(&dst[w], v); }
        else dst[w] = v;
    }
}

// amino acids: the same two-pass compaction with one byte per kept residue (aafiles.rs:11-28 filter_out_non_aa: letters outside the
// 20-letter alphabet - '*', X, B, Z, ..., and here the newlines of the raw text - are dropped; either case is kept as read)
__device__ __forceinline__ bool aa_valid(uint8_t c)
{
    const uint32_t u = c & 0xDFu;                                // fold case
    if (u < 'A' || u > 'Z') return false;
    // A C D E F G H I K L M N P Q R S T V W Y  (not B J O U X Z)
    return (0x016FBDFDu >> (u - 'A')) & 1u;
}
__global__ __launch_bounds__(PK_T) void k_aa_count(const uint8_t *__restrict__ text, const uint64_t *__restrict__ cb, const uint64_t *__restrict__ ce,
                                                    uint32_t *__restrict__ counts)
{
    __shared__ uint32_t s_n;
    const uint64_t c = blockIdx.x;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    uint32_t loc = 0;
    for (uint64_t i = cb[c] + threadIdx.x; i < ce[c]; i += PK_T) loc += aa_valid(text[i]);
    for (int o = 32; o > 0; o >>= 1) loc += __shfl_down(loc, o);
    if ((threadIdx.x & 63) == 0 && loc) atomicAdd(&s_n, loc);
    __syncthreads();
    if (threadIdx.x == 0) counts[c] = s_n;
}
__global__ __launch_bounds__(PK_T) void k_aa_write(const uint8_t *__restrict__ text, const uint64_t *__restrict__ cb, const uint64_t *__restrict__ ce,
                                                    const uint64_t *__restrict__ out_base, uint8_t *__restrict__ out)
{
    __shared__ uint32_t s_wave[PK_T / 64];
    __shared__ uint32_t s_run;
    const uint64_t c = blockIdx.x, b0 = cb[c], b1 = ce[c], ob = out_base[c];
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint64_t i0 = b0; i0 < b1; i0 += PK_T) {
        const uint64_t i = i0 + threadIdx.x;
        const uint8_t ch = i < b1 ? text[i] : 0;
        const bool ok = i < b1 && aa_valid(ch);
        const uint64_t bal = __ballot(ok);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t off = s_run;
        for (uint32_t w = 0; w < wv; w++) off += s_wave[w];
        if (ok) out[ob + off + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = ch;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (uint32_t w = 0; w < PK_T / 64; w++) t += s_wave[w]; s_run += t; }
        __syncthreads();
    }
}

// shared driver of the DNA pack and the AA filter. contiguous = the records follow one another without alignment gaps and form ONE
// output record (--block: process_file_in_one_block appends every record to one Sequence, k-mers span the joins, dnafiles.rs:200-262)
int ingest_records_dev(gs_ctx *c, bool aa, bool contiguous, const void *text_dev, uint64_t n_bytes, const uint64_t *seq_begin, const uint64_t *seq_end,
                       uint64_t n_rec, void *out_dev, uint64_t out_base0, uint64_t *rec_start_out, uint64_t *rec_len_out, uint64_t *out_end)
{
    if (out_end) *out_end = out_base0;
    if (n_rec == 0) return GS_OK;
    GS_REQUIRE(text_dev && seq_begin && seq_end && out_dev, GS_ERR_INVALID, "null argument");
    std::vector<uint64_t> cb, ce, first_chunk(n_rec + 1);
    for (uint64_t r = 0; r < n_rec; r++) {
        GS_REQUIRE(seq_begin[r] <= seq_end[r] && seq_end[r] <= n_bytes, GS_ERR_INVALID, "record %llu outside the text", (unsigned long long)r);
        first_chunk[r] = cb.size();
        for (uint64_t b = seq_begin[r]; b < seq_end[r]; b += PK_CHUNK) { cb.push_back(b); ce.push_back(std::min<uint64_t>(b + PK_CHUNK, seq_end[r])); }
    }
    first_chunk[n_rec] = cb.size();
    const uint64_t nch = cb.size();
    for (uint64_t r = 0; r < n_rec; r++) { rec_start_out[r] = out_base0; rec_len_out[r] = 0; }
    if (nch == 0) return GS_OK;
    PoolBuf dcb(c, 10), dce(c, 11), dcnt(c, 12), dbase(c, 13);
    int rc;
    if ((rc = dcb.alloc(8 * nch))) return rc;
    if ((rc = dce.alloc(8 * nch))) return rc;
    if ((rc = dcnt.alloc(4 * nch))) return rc;
    if ((rc = dbase.alloc(8 * nch))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(dcb.p, cb.data(), 8 * nch, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dce.p, ce.data(), 8 * nch, hipMemcpyHostToDevice, c->stream));
    if (aa) hipLaunchKernelGGL(k_aa_count, dim3((uint32_t)nch), dim3(PK_T), 0, c->stream, (const uint8_t *)text_dev, dcb.as<uint64_t>(), dce.as<uint64_t>(), dcnt.as<uint32_t>());
    else hipLaunchKernelGGL(k_pack_count, dim3((uint32_t)nch), dim3(PK_T), 0, c->stream, (const uint8_t *)text_dev, dcb.as<uint64_t>(), dce.as<uint64_t>(), dcnt.as<uint32_t>());
    GS_HIP_CHECK(hipGetLastError());
    std::vector<uint32_t> cnt(nch);
    GS_HIP_CHECK(hipMemcpyAsync(cnt.data(), dcnt.p, 4 * nch, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(gs::stream_wait(c));
    std::vector<uint64_t> base(nch);
    uint64_t pos = out_base0;
    for (uint64_t r = 0; r < n_rec; r++) {
        if (!contiguous && !aa) pos = (pos + 31) / 32 * 32;       // DNA records start on a packed-word boundary
        rec_start_out[r] = pos;
        for (uint64_t ch = first_chunk[r]; ch < first_chunk[r + 1]; ch++) { base[ch] = pos; pos += cnt[ch]; }
        rec_len_out[r] = pos - rec_start_out[r];
    }
    if (contiguous) { rec_len_out[0] = pos - rec_start_out[0]; for (uint64_t r = 1; r < n_rec; r++) { rec_start_out[r] = pos; rec_len_out[r] = 0; } }
    if (out_end) *out_end = pos;
    GS_HIP_CHECK(hipMemcpyAsync(dbase.p, base.data(), 8 * nch, hipMemcpyHostToDevice, c->stream));
    if (aa) hipLaunchKernelGGL(k_aa_write, dim3((uint32_t)nch), dim3(PK_T), 0, c->stream, (const uint8_t *)text_dev, dcb.as<uint64_t>(), dce.as<uint64_t>(), dbase.as<uint64_t>(), (uint8_t *)out_dev);
    else hipLaunchKernelGGL(k_pack_write, dim3((uint32_t)nch), dim3(PK_T), 0, c->stream, (const uint8_t *)text_dev, dcb.as<uint64_t>(), dce.as<uint64_t>(), dbase.as<uint64_t>(), (uint32_t *)out_dev);
    GS_HIP_CHECK(hipGetLastError());
    GS_HIP_CHECK(gs::stream_wait(c));
    return GS_OK;
}

}  // namespace gs

extern "C" {

/* host: record boundaries of a FASTA text. Record r: sequence text = bytes [seq_begin[r], seq_end[r]) (may contain newlines),
 * id = the header's first word (reporting only). Records whose header line contains "capsid" anywhere are skipped when
 * skip_capsid != 0 (needletail id() = whole header; dnafiles.rs:62-67, aafiles.rs:78,133,190,256). */
int gs_fasta_scan(const char *buf, uint64_t n, int skip_capsid, uint64_t cap, uint64_t *seq_begin, uint64_t *seq_end, uint64_t *id_begin,
                  uint32_t *id_len, uint64_t *n_rec_out)
{
    GS_REQUIRE(buf || n == 0, GS_ERR_INVALID, "null buffer");
    GS_REQUIRE(n_rec_out, GS_ERR_INVALID, "null n_rec_out");
    uint64_t nr = 0, i = 0;
    while (i < n && buf[i] != '>') { const char *nl = (const char *)memchr(buf + i, '\n', n - i); if (!nl) { i = n; break; } i = (uint64_t)(nl - buf) + 1; }
    while (i < n) {
        // header line
        const uint64_t h0 = i + 1;
        const char *nl = (const char *)memchr(buf + i, '\n', n - i);
        const uint64_t hend = nl ? (uint64_t)(nl - buf) : n;
        uint64_t idl = 0;
        while (h0 + idl < hend && buf[h0 + idl] != ' ' && buf[h0 + idl] != '\t' && buf[h0 + idl] != '\r') idl++;
        const uint64_t s0 = nl ? hend + 1 : n;
        // sequence text runs to the next line that starts with '>'
        uint64_t j = s0;
        while (j < n) {
            if (buf[j] == '>') break;
            const char *nl2 = (const char *)memchr(buf + j, '\n', n - j);
            if (!nl2) { j = n; break; }
            j = (uint64_t)(nl2 - buf) + 1;
        }
        bool skip = false;
        if (skip_capsid) {
            static const char pat[] = "capsid";
            // needletail's id() is the WHOLE header line (description included, line ending stripped); dnafiles.rs:62-67 tests
            // strid.contains("capsid") on it, so ">NC_1.1 Foo virus capsid protein" is dropped too. The first word is only reported.
            uint64_t hl = hend > h0 ? hend - h0 : 0;
            if (hl && buf[h0 + hl - 1] == '\r') hl--;
            for (uint64_t t = 0; t + 6 <= hl && !skip; t++) skip = !memcmp(buf + h0 + t, pat, 6);
        }
        if (!skip) {
            if (nr < cap) { if (seq_begin) seq_begin[nr] = s0; if (seq_end) seq_end[nr] = j; if (id_begin) id_begin[nr] = h0; if (id_len) id_len[nr] = (uint32_t)idl; }
            nr++;
        }
        i = j;
    }
    *n_rec_out = nr;
    return GS_OK;
}

/* device: pack the sequence text of n_rec records to 2 bits. text_dev: the raw text (n_bytes); seq_begin/seq_end: HOST arrays from
 * gs_fasta_scan (offsets into the text); packed_dev: ZEROED device buffer of at least n_bytes/4 + 8*n_rec + 64 bytes; every record
 * starts on a 32-base boundary. rec_start_out / rec_len_out (HOST) receive the base coordinates for gs_sketch_batch_dev. */
int gs_pack_fasta_dev(gs_ctx *c, const void *text_dev, uint64_t n_bytes, const uint64_t *seq_begin, const uint64_t *seq_end, uint64_t n_rec,
                      void *packed_dev, uint64_t *rec_start_out, uint64_t *rec_len_out)
{
    GS_REQUIRE(c && rec_start_out && rec_len_out, GS_ERR_INVALID, "null argument");
    if (n_rec == 0) return GS_OK;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    return gs::ingest_records_dev(c, false, false, text_dev, n_bytes, seq_begin, seq_end, n_rec, packed_dev, 0, rec_start_out, rec_len_out, nullptr);
}

/* device: drop everything outside the 20-letter amino-acid alphabet from the text of n_rec records (filter_out_non_aa, aafiles.rs:11-28;
 * newlines go with it). out_dev: >= n_bytes bytes; rec_start_out / rec_len_out (HOST): residue coordinates for gs_sketch_batch_dev. */
int gs_filter_aa_dev(gs_ctx *c, const void *text_dev, uint64_t n_bytes, const uint64_t *seq_begin, const uint64_t *seq_end, uint64_t n_rec,
                     void *out_dev, uint64_t *rec_start_out, uint64_t *rec_len_out)
{
    GS_REQUIRE(c && rec_start_out && rec_len_out, GS_ERR_INVALID, "null argument");
    if (n_rec == 0) return GS_OK;
    GS_CTX_LOCK(c);
    GS_HIP_CHECK(hipSetDevice(c->device));
    return gs::ingest_records_dev(c, true, false, text_dev, n_bytes, seq_begin, seq_end, n_rec, out_dev, 0, rec_start_out, rec_len_out, nullptr);
}

}  // extern "C"
