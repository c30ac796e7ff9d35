// gs_inflate.hip — gzip members inflated ON the device ("gunzip << PCIe"; the reference reads .gz through
// needletail's flate2 reader on --pio host threads, files.rs:258-341, dnafiles.rs:114-193).
//
// A host core inflates ~0.5 GB/s of FASTA text (libdeflate: 9.8 ms per 5 Mbp genome), so 16 cores feed the sketch kernels ~1000 genomes/s
// against the 88 000/s they sustain from HBM. DEFLATE is serial WITHIN a stream, but a request is thousands of independent streams: here
// one wave decodes one member (RFC 1951: stored / fixed / dynamic blocks), 4 waves per CU, 1024 streams in flight per device.
//   * the 32 KB history window is a circular buffer in LDS; every completed 16 KB half is written to HBM coalesced (16 B per lane);
//   * the wave runs the bit reader, the Huffman look-ups (10-bit litlen / 8-bit distance root tables in LDS, canonical bit-by-bit
//     walk for the rare longer codes) uniformly; the compressed words come from a 64-lane register cache (one coalesced 256 B load per 64
//     words, fetched with v_readlane), never from a dependent global load;
//   * a match is copied by the lanes in parallel (64 bytes per step, period-`dist` addressing when the source overlaps the target);
//   * table construction (code counts, canonical codes, root-table fill) is lane-parallel over the symbols (ballot ranks).
// CRC-32 of the produced text is a second, fully parallel kernel (raw CRCs of 256-byte pieces folded in a tree with x^(8 len) multipliers
// in GF(2)[x]/P; the host conditions and compares with the member's trailer). Anything this path does not take (multi-member files,
// reserved header flags, a member whose trailer disagrees) is handed back to the host decoder by the caller: same bytes either way.
#include "gs_internal.hpp"
#include "gs_inflate.hpp"
#include <string.h>
#include <algorithm>

namespace gs {

constexpr uint32_t INF_WSIZE = 32768, INF_WMASK = INF_WSIZE - 1, INF_HALF = 16384;
constexpr uint32_t INF_LROOT = 10, INF_DROOT = 8, INF_CROOT = 7;
constexpr uint32_t INF_LONG = 0x30;          // root entry of a prefix shared by codes longer than the root

__device__ __forceinline__ uint32_t inf_litlen_entry(uint32_t s, uint32_t L)
{
    if (s < 256) return L | (s << 8);
    if (s == 256) return L | (2u << 4);
    if (s > 285) return 0;
    uint32_t base, eb;
    if (s < 265) { base = s - 254; eb = 0; }
    else if (s == 285) { base = 258; eb = 0; }
    else { eb = (s - 261) >> 2; base = 3 + ((4 + ((s - 261) & 3)) << eb); }
    return L | (1u << 4) | (base << 8) | (eb << 24);
}
__device__ __forceinline__ uint32_t inf_dist_entry(uint32_t s, uint32_t L)
{
    if (s > 29) return 0;
    uint32_t base, eb;
    if (s < 4) { base = 1 + s; eb = 0; }
    else { eb = (s >> 1) - 1; base = 1 + ((2 + (s & 1)) << eb); }
    return L | (1u << 4) | (base << 8) | (eb << 24);
}
template <int KIND> __device__ __forceinline__ uint32_t inf_entry(uint32_t s, uint32_t L)
{
    return KIND == 0 ? inf_litlen_entry(s, L) : KIND == 1 ? inf_dist_entry(s, L) : (L | (s << 8));
}

// LDS of one stream (one wave per workgroup): 39.6 KB -> four streams per CU. File scope so that the table builder can be a real function:
// inlined five times into the decode loop it made one 45 KB instruction stream whose taken branches each cost the (only) wave of the SIMD a
// fetch stall.
struct InfLds {
    uint32_t llut[1u << INF_LROOT], dlut[1u << INF_DROOT], clut[1u << INF_CROOT];
    uint8_t lens[320], clens[20];
    uint16_t lcnt[16], dcnt[16], ccnt[16], nextc[16], offs[16], lsorted[288], dsorted[32], csorted[20];
};
static __shared__ __attribute__((aligned(16))) InfLds g_inf;
// history window of the LDS form, circular (the GWIN form of k_inflate reads its history back from the text it has written: no window)
static __shared__ __attribute__((aligned(16))) uint8_t g_win[INF_WSIZE];

// Canonical Huffman tables of one alphabet from its code lengths (RFC 1951 3.2.2): KIND 0 = literal/length (g_inf.lens[first ..]), 1 = distance,
// 2 = code-length alphabet (g_inf.clens). Fills the root table (code length in bits 0-3, kind in 4-5, value from bit 8, extra-bit count from
// bit 24; 0 = no such code), the per-length counts and the symbols sorted by (length, symbol) for the bit-by-bit walk of codes longer than the
// root. Returns 0, or 1 when the lengths over-subscribe the code space or leave it incomplete (zlib's rule).
template <int KIND>
__device__ __noinline__ int inf_build(uint32_t first, uint32_t nsym)
{
    const uint32_t lane = threadIdx.x;
    constexpr uint32_t root = KIND == 0 ? INF_LROOT : KIND == 1 ? INF_DROOT : INF_CROOT;
    const uint8_t *lens = KIND == 2 ? g_inf.clens : g_inf.lens + first;
    uint32_t *lut = KIND == 0 ? g_inf.llut : KIND == 1 ? g_inf.dlut : g_inf.clut;
    uint16_t *cnt = KIND == 0 ? g_inf.lcnt : KIND == 1 ? g_inf.dcnt : g_inf.ccnt;
    uint16_t *sorted = KIND == 0 ? g_inf.lsorted : KIND == 1 ? g_inf.dsorted : g_inf.csorted;
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    for (uint32_t i = lane; i < (1u << root); i += 64) lut[i] = 0;
    uint32_t c[16];
#pragma unroll
    for (int l = 0; l < 16; l++) c[l] = 0;
    for (uint32_t b = 0; b < nsym; b += 64) {
        const uint32_t L = b + lane < nsym ? lens[b + lane] : 0;
#pragma unroll
        for (int l = 1; l < 16; l++) c[l] += (uint32_t)__popcll(__ballot(L == (uint32_t)l));
    }
    int left = 1;
    uint32_t code = 0, off = 0;
#pragma unroll
    for (int l = 1; l < 16; l++) {
        left = left * 2 - (int)c[l];
        code = (code + c[l - 1]) << 1;
        if (lane == 0) { cnt[l] = (uint16_t)c[l]; g_inf.nextc[l] = (uint16_t)code; g_inf.offs[l] = (uint16_t)off; }
        off += c[l];
    }
    if (lane == 0) cnt[0] = 0;
    if (left < 0) return 1;
    // an INCOMPLETE set is refused too, as zlib (inftrees.c) and libdeflate do - the member then goes to the host decoder, which reports the corrupt
    // stream: whether a malformed file is accepted must not depend on which pipeline the dealer handed it to. zlib's one exception is kept: a
    // literal/length or distance alphabet with a single code of length 1, or no code at all
    int maxlen = 0;
#pragma unroll
    for (int l = 1; l < 16; l++) if (c[l]) maxlen = l;
    if (left > 0 && (KIND == 2 || maxlen > 1)) return 1;
    __syncthreads();
    uint32_t run[16];
#pragma unroll
    for (int l = 0; l < 16; l++) run[l] = 0;
    for (uint32_t b = 0; b < nsym; b += 64) {
        const uint32_t s = b + lane;
        const uint32_t L = s < nsym ? lens[s] : 0;
        uint32_t rank = 0;
#pragma unroll
        for (int l = 1; l < 16; l++) {
            const uint64_t m = __ballot(L == (uint32_t)l);
            if (L == (uint32_t)l) rank = run[l] + (uint32_t)__popcll(m & lt);
            run[l] += (uint32_t)__popcll(m);
        }
        if (L) {
            const uint32_t cd = (uint32_t)g_inf.nextc[L] + rank;
            sorted[(uint32_t)g_inf.offs[L] + rank] = (uint16_t)s;
            const uint32_t rev = __brev(cd) >> (32 - L);
            if (L <= root) { const uint32_t e = inf_entry<KIND>(s, L); for (uint32_t i = rev; i < (1u << root); i += 1u << L) lut[i] = e; }
            else lut[rev & ((1u << root) - 1)] = INF_LONG;
        }
    }
    __syncthreads();
    return 0;
}

// bit-by-bit canonical decode (codes longer than the root table): the symbol's table entry, 0 = no such code
template <int KIND>
__device__ __noinline__ uint32_t inf_walk(uint64_t buf)
{
    const uint16_t *cnt = KIND == 0 ? g_inf.lcnt : g_inf.dcnt;
    const uint16_t *sorted = KIND == 0 ? g_inf.lsorted : g_inf.dsorted;
    uint32_t code = 0, first = 0, index = 0;
    for (uint32_t l = 1; l <= 15; l++) {
        code |= (uint32_t)(buf >> (l - 1)) & 1u;
        const uint32_t c = cnt[l];
        if (code < first + c) return inf_entry<KIND>(sorted[index + (code - first)], l);
        index += c; first = (first + c) << 1; code <<= 1;
    }
    return 0;
}

// the bit reader: buf holds cnt valid bits (LSB first); the compressed words come from the lanes' registers (cur = the 64-word block that
// holds word wi, nxt = the block after it), one coalesced load per 64 words
struct InfBits { uint64_t buf; uint32_t cnt, wi, cur, nxt; };
__device__ __forceinline__ uint32_t inf_uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t inf_uni64(uint64_t x) { return (uint64_t)inf_uni((uint32_t)x) | (uint64_t)inf_uni((uint32_t)(x >> 32)) << 32; }
#define INF_REFILL(B)                                                                                                          \
    do {                                                                                                                       \
        if ((B).cnt <= 32) {                                                                                                   \
            const uint32_t w_ = (uint32_t)__builtin_amdgcn_readlane((int)(B).cur, (int)((B).wi & 63u));                        \
            (B).buf |= (uint64_t)w_ << (B).cnt; (B).cnt += 32; (B).wi++;                                                       \
            if (((B).wi & 63u) == 0) { (B).cur = (B).nxt; (B).nxt = (B).wi + 64 + lane < nwords ? comp[(B).wi + 64 + lane] : 0; } \
        }                                                                                                                      \
    } while (0)
#define INF_TAKE(B, nb_) do { const uint32_t t_ = (nb_); (B).buf >>= t_; (B).cnt -= t_; } while (0)

// header of a dynamic block (RFC 1951 3.2.7): the code lengths of both alphabets, then their tables. status in *st
__device__ __noinline__ InfBits inf_dynamic_header(InfBits B, const uint32_t *__restrict__ comp, uint32_t nwords, uint32_t *st)
{
    const uint32_t lane = threadIdx.x;
    B.buf = inf_uni64(B.buf); B.cnt = inf_uni(B.cnt); B.wi = inf_uni(B.wi);
    *st = INF_OK;
    INF_REFILL(B);
    const uint32_t hlit = ((uint32_t)B.buf & 31) + 257, hdist = ((uint32_t)(B.buf >> 5) & 31) + 1, hclen = ((uint32_t)(B.buf >> 10) & 15) + 4;
    INF_TAKE(B, 14);
    if (hlit > 286 || hdist > 30) { *st = INF_E_CODE; return B; }
    if (lane < 20) g_inf.clens[lane] = 0;
    __syncthreads();
    for (uint32_t i = 0; i < hclen; i++) {
        INF_REFILL(B);
        // order of the code-length code lengths: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
        const uint32_t ord = i < 3 ? 16 + i : i == 3 ? 0 : (i & 1) ? 7 - ((i - 5) >> 1) : 8 + ((i - 4) >> 1);
        if (lane == 0) g_inf.clens[ord] = (uint8_t)((uint32_t)B.buf & 7);
        INF_TAKE(B, 3);
    }
    __syncthreads();
    if (inf_build<2>(0, 19)) { *st = INF_E_OVERSUB; return B; }
    const uint32_t total = hlit + hdist;
    for (uint32_t i = 0; i < total;) {
        INF_REFILL(B);
        const uint32_t e = inf_uni(g_inf.clut[(uint32_t)B.buf & ((1u << INF_CROOT) - 1)]);
        const uint32_t nb = e & 15;
        if (!nb) { *st = INF_E_CODE; return B; }
        INF_TAKE(B, nb);
        const uint32_t sym = e >> 8;
        if (sym < 16) { if (lane == 0) g_inf.lens[i] = (uint8_t)sym; i++; continue; }
        uint32_t rep, v = 0;
        if (sym == 16) { if (i == 0) { *st = INF_E_REPEAT; return B; } v = g_inf.lens[i - 1]; rep = 3 + ((uint32_t)B.buf & 3); INF_TAKE(B, 2); }
        else if (sym == 17) { rep = 3 + ((uint32_t)B.buf & 7); INF_TAKE(B, 3); }
        else { rep = 11 + ((uint32_t)B.buf & 127); INF_TAKE(B, 7); }
        if (i + rep > total) { *st = INF_E_REPEAT; return B; }
        for (uint32_t j = lane; j < rep; j += 64) g_inf.lens[i + j] = (uint8_t)v;
        i += rep;
    }
    __syncthreads();
    if (g_inf.lens[256] == 0) { *st = INF_E_CODE; return B; }                 // no end-of-block code
    if (inf_build<0>(0, hlit) || inf_build<1>(hlit, hdist)) *st = INF_E_OVERSUB;
    return B;
}

// GWIN = false: the window in LDS (39.6 KB per member: four members per CU, ~1500 cycles per symbol). GWIN = true: no window - literals and
// copies go straight to the text in HBM and a match reads its source back from there (through L2, past the non-coherent L1): ~2x the
// latency per symbol, but 7.6 KB of LDS per member, so five times the members per CU interleave on the same issue slots.
template <bool GWIN>
__global__ __launch_bounds__(64) void k_inflate(const uint32_t *__restrict__ comp, const InflateStream *__restrict__ sts, uint32_t n, uint8_t *__restrict__ out_all,
                                                InflateResult *__restrict__ res)
{
    const uint32_t lane = threadIdx.x;
    const InflateStream st = sts[blockIdx.x];
    uint8_t *out = out_all + st.out_off;
    uint8_t *win = g_win;
    const uint32_t cap = (uint32_t)st.out_cap;
    const uint64_t in_end = st.in_off + st.in_len;                       // absolute byte offsets into comp
    const uint32_t nwords = (uint32_t)((in_end + 3) >> 2);
    InfBits B;
    B.wi = (uint32_t)(st.in_off >> 2);
    { const uint32_t b0 = B.wi & ~63u; B.cur = b0 + lane < nwords ? comp[b0 + lane] : 0; B.nxt = b0 + 64 + lane < nwords ? comp[b0 + 64 + lane] : 0; }
    B.buf = 0; B.cnt = 0;
    INF_REFILL(B);
    INF_TAKE(B, (uint32_t)(st.in_off & 3) * 8);
    uint32_t pos = 0, status = INF_OK, blocks = 0;
    uint32_t synced = 0;                  // GWIN: every byte of the text below this position is known to have reached L2
    auto flush_half = [&](uint32_t base) {
        if (GWIN) return;
#pragma unroll 4
        for (uint32_t k = 0; k < INF_HALF / 1024; k++) {
            const uint32_t off = k * 1024 + lane * 16;
            *(uint4 *)(out + base + off) = *(const uint4 *)&win[(base + off) & INF_WMASK];
        }
    };
    for (bool last = false; !last && status == INF_OK;) {
        INF_REFILL(B);
        last = B.buf & 1;
        const uint32_t btype = (uint32_t)(B.buf >> 1) & 3;
        INF_TAKE(B, 3);
        blocks++;
        if (btype == 3) { status = INF_E_BTYPE; break; }
        if (btype == 0) {                                     // stored
            INF_TAKE(B, B.cnt & 7);
            INF_REFILL(B);
            const uint32_t len = (uint32_t)B.buf & 0xFFFFu, nlen = (uint32_t)(B.buf >> 16) & 0xFFFFu;
            INF_TAKE(B, 32);
            if ((len ^ 0xFFFFu) != nlen) { status = INF_E_STORED; break; }
            if (len > cap - pos) { status = INF_E_OUTPUT; break; }
            for (uint32_t i = 0; i < len; i++) {
                INF_REFILL(B);
                if (lane == 0) { if (GWIN) out[pos] = (uint8_t)B.buf; else win[pos & INF_WMASK] = (uint8_t)B.buf; }
                INF_TAKE(B, 8);
                pos++;
                if ((pos & (INF_HALF - 1)) == 0) flush_half(pos - INF_HALF);
            }
            continue;
        }
        if (btype == 1) {                                     // fixed codes (RFC 1951 3.2.6)
            for (uint32_t i = lane; i < 288; i += 64) g_inf.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            if (lane < 32) g_inf.lens[288 + lane] = 5;
            __syncthreads();
            (void)inf_build<0>(0, 288);
            (void)inf_build<1>(288, 32);
        } else {
            uint32_t hst;
            B = inf_dynamic_header(B, comp, nwords, &hst);
            if (hst != INF_OK) { status = hst; break; }
        }
        B.buf = inf_uni64(B.buf); B.cnt = inf_uni(B.cnt); B.wi = inf_uni(B.wi);
        // The symbols of the block. Per match the wave makes two dependent LDS round trips (length code, distance code); the copy's own
        // round trip overlaps the NEXT symbol's table look-up, which is issued before the copied bytes are waited for.
        INF_REFILL(B);
        uint32_t e = inf_uni(g_inf.llut[(uint32_t)B.buf & ((1u << INF_LROOT) - 1)]);
        for (;;) {
            // (the state is wave-uniform by construction; saying so once per symbol keeps the whole chain on the scalar unit)
            e = inf_uni(e); B.buf = inf_uni64(B.buf); B.cnt = inf_uni(B.cnt); B.wi = inf_uni(B.wi); pos = inf_uni(pos);
            if (e == INF_LONG) e = inf_uni(inf_walk<0>(B.buf));
            uint32_t nb = e & 15;
            if (!nb) { status = INF_E_CODE; break; }
            if (B.wi > nwords + 2) { status = INF_E_INPUT; break; }      // past the end of the member: the zeros fed from there on may decode for ever
            INF_TAKE(B, nb);
            const uint32_t kind = (e >> 4) & 3;
            if (kind == 0) {
                if (pos >= cap) { status = INF_E_OUTPUT; break; }
                if (lane == 0) { if (GWIN) out[pos] = (uint8_t)(e >> 8); else win[pos & INF_WMASK] = (uint8_t)(e >> 8); }
                pos++;
                INF_REFILL(B);
                e = inf_uni(g_inf.llut[(uint32_t)B.buf & ((1u << INF_LROOT) - 1)]);
                if ((pos & (INF_HALF - 1)) == 0) flush_half(pos - INF_HALF);
                continue;
            }
            if (kind == 2) break;
            uint32_t eb = (e >> 24) & 15;
            const uint32_t len = ((e >> 8) & 0xFFFFu) + ((uint32_t)B.buf & ((1u << eb) - 1));
            INF_TAKE(B, eb);
            INF_REFILL(B);
            uint32_t d = inf_uni(g_inf.dlut[(uint32_t)B.buf & ((1u << INF_DROOT) - 1)]);
            if (d == INF_LONG) d = inf_uni(inf_walk<1>(B.buf));
            nb = d & 15;
            if (!nb) { status = INF_E_CODE; break; }
            INF_TAKE(B, nb);
            eb = (d >> 24) & 15;
            const uint32_t dist = ((d >> 8) & 0xFFFFu) + ((uint32_t)B.buf & ((1u << eb) - 1));
            INF_TAKE(B, eb);
            if (dist > pos) { status = INF_E_DIST; break; }
            if (len > cap - pos) { status = INF_E_OUTPUT; break; }
            INF_REFILL(B);
            const uint32_t e_next = g_inf.llut[(uint32_t)B.buf & ((1u << INF_LROOT) - 1)];      // in flight under the copy
            const uint32_t src0 = pos - dist;
            if (GWIN) {
                // the source may be bytes this wave stored a moment ago: a store is acknowledged (vmcnt) once L2 has it, and the load below
                // goes to L2 (agent scope: not served by the CU's L1, which does not see L2 writes). Loads may overtake this wave's own stores,
                // so every byte at or beyond `synced` (the text position at the last wait) counts as possibly in flight: wait exactly when the
                // source reaches into that range
                if (src0 + (dist >= len ? len : dist) > synced) { __builtin_amdgcn_s_waitcnt(0x0F70); synced = pos; }          // vmcnt(0)
                for (uint32_t i = lane; i < len; i += 64) {
                    const uint8_t b = __hip_atomic_load(out + src0 + (dist >= len ? i : i % dist), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    out[pos + i] = b;
                }
            } else if (dist >= len) {
                for (uint32_t i = lane; i < len; i += 64) win[(pos + i) & INF_WMASK] = win[(src0 + i) & INF_WMASK];
            } else {                                          // the source runs into the target: the last `dist` bytes repeat
                for (uint32_t i = lane; i < len; i += 64) win[(pos + i) & INF_WMASK] = win[(src0 + i % dist) & INF_WMASK];
            }
            const uint32_t np = pos + len;
            if ((pos ^ np) & ~(INF_HALF - 1)) flush_half((np & ~(INF_HALF - 1)) - INF_HALF);
            pos = np;
            e = inf_uni(e_next);
        }
    }
    // the tail of the text (the window half that never filled)
    if (!GWIN && status == INF_OK) {
        const uint32_t base = pos & ~(INF_HALF - 1), rem = pos - base;
        for (uint32_t off = lane * 16; off < rem; off += 1024) *(uint4 *)(out + base + off) = *(const uint4 *)&win[(base + off) & INF_WMASK];
    }
    INF_TAKE(B, B.cnt & 7);
    const uint64_t used = (uint64_t)B.wi * 4 - B.cnt / 8 - st.in_off;
    if (status == INF_OK && used > st.in_len) status = INF_E_INPUT;
    if (lane == 0) { InflateResult r; r.status = status; r.blocks = blocks; r.in_used = used; r.out_len = pos; res[blockIdx.x] = r; }
}

// ---- the window-less form with several matches in flight (round 4) ---------------------------------------------------------------------
// What bounded k_inflate<true>: a member's 32 KB of history lives in the text it has written, 20 members per CU keep ~20 MB of history per
// XCD alive against 4 MB of L2, so the source bytes of a match come back from the Infinity Cache / HBM (~1.5-2 us) - and the wave waited
// for every one of them before it stored the copy and went on (gzip -6 of DNA is one match per ~5 bytes: ~1 460 cycles per symbol and wave
// at the 1.5 waves per SIMD of a 6 x CUs group). Decoding does not depend on the copied bytes, only later copies may: here the wave decodes a
// batch of up to P matches, stores the copies of the PREVIOUS batch after one wait and then issues this batch's source loads, which have the
// next round of decoding to come back. A match whose source reaches into text that is not known to have arrived in L2 (`synced`: the first
// destination of the batch in flight - everything below it had been stored before the last wait) drains both batches first; so do matches
// longer than one load per lane (64 bytes). Measured: 587 -> 383 ms per 1536 x 5 Mbp members, most of it from the instructions this
// restructuring took off the per-symbol chain - the kernel executes 70 scalar + 14 vector instructions per symbol.
template <int KIND>
__device__ __forceinline__ uint32_t inf_walk_inl(uint64_t buf)
{
    const uint16_t *cnt = KIND == 0 ? g_inf.lcnt : g_inf.dcnt;
    const uint16_t *sorted = KIND == 0 ? g_inf.lsorted : g_inf.dsorted;
    uint32_t code = 0, first = 0, index = 0, r = 0;
#pragma unroll 1
    for (uint32_t l = 1; l <= 15; l++) {
        code |= (uint32_t)(buf >> (l - 1)) & 1u;
        const uint32_t c = cnt[l];
        if (code < first + c) { r = inf_entry<KIND>(sorted[index + (code - first)], l); break; }
        index += c; first = (first + c) << 1; code <<= 1;
    }
    return r;
}
// The root tables of a block in REGISTERS: entry i lives in lane i & 63 of register i >> 6 (16 registers for the 10-bit literal/length table, 4
// of 16 for the 8-bit distance table). A look-up is an indexed register read with a wave-uniform index plus v_readlane - a few cycles - where the LDS
// read + wait + v_readfirstlane of the other forms is ~140 (by itself this changed nothing measurable: 434.7 vs 434.7 ms - what it buys is one LDS
// instruction per nine symbols instead of three per symbol on a chain that is bound by its instruction count).
typedef uint32_t inf_u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ uint32_t inf_lut16(const inf_u32x16 &t, uint32_t idx) { const uint32_t v = t[idx >> 6]; return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)(idx & 63u)); }
template <int P>
__global__ __launch_bounds__(64) void k_inflate_pipe(const uint32_t *__restrict__ comp, const InflateStream *__restrict__ sts, uint32_t n, uint8_t *__restrict__ out_all,
                                                     InflateResult *__restrict__ res)
{
    const uint32_t lane = threadIdx.x;
    const InflateStream st = sts[blockIdx.x];
    uint8_t *out = out_all + st.out_off;
    const uint32_t cap = (uint32_t)st.out_cap;
    const uint64_t in_end = st.in_off + st.in_len;
    const uint32_t nwords = (uint32_t)((in_end + 3) >> 2);
    InfBits B;
    B.wi = (uint32_t)(st.in_off >> 2);
    { const uint32_t b0 = B.wi & ~63u; B.cur = b0 + lane < nwords ? comp[b0 + lane] : 0; B.nxt = b0 + 64 + lane < nwords ? comp[b0 + 64 + lane] : 0; }
    B.buf = 0; B.cnt = 0;
    INF_REFILL(B);
    INF_TAKE(B, (uint32_t)(st.in_off & 3) * 8);
    uint32_t pos = 0, status = INF_OK, blocks = 0;
    uint32_t synced = 0;                  // every byte of the text below this position has reached L2
    // the history is read through a buffer resource (raw byte loads with the sc1 bit; the compiler counts them in vmcnt, so the wait sits where a
    // loaded byte is first used - an agent-scope atomic byte load is widened and masked right behind the load, which waits for it there)
    const __amdgpu_buffer_rsrc_t hist = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)cap, 0x00020000);
    for (bool last = false; !last && status == INF_OK;) {
        INF_REFILL(B);
        last = B.buf & 1;
        const uint32_t btype = (uint32_t)(B.buf >> 1) & 3;
        INF_TAKE(B, 3);
        blocks++;
        if (btype == 3) { status = INF_E_BTYPE; break; }
        if (btype == 0) {                                     // stored
            INF_TAKE(B, B.cnt & 7);
            INF_REFILL(B);
            const uint32_t len = (uint32_t)B.buf & 0xFFFFu, nlen = (uint32_t)(B.buf >> 16) & 0xFFFFu;
            INF_TAKE(B, 32);
            if ((len ^ 0xFFFFu) != nlen) { status = INF_E_STORED; break; }
            if (len > cap - pos) { status = INF_E_OUTPUT; break; }
            for (uint32_t i = 0; i < len; i++) {
                INF_REFILL(B);
                if (lane == 0) out[pos] = (uint8_t)B.buf;
                INF_TAKE(B, 8);
                pos++;
            }
            continue;
        }
        if (btype == 1) {                                     // fixed codes (RFC 1951 3.2.6)
            for (uint32_t i = lane; i < 288; i += 64) g_inf.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            if (lane < 32) g_inf.lens[288 + lane] = 5;
            __syncthreads();
            (void)inf_build<0>(0, 288);
            (void)inf_build<1>(288, 32);
        } else {
            uint32_t hst;
            B = inf_dynamic_header(B, comp, nwords, &hst);
            hst = inf_uni(hst);                                // (it comes back through memory: without this the compiler takes `status`, and every
            if (hst != INF_OK) { status = hst; break; }       //  variable assigned under a test of it, for lane-dependent)
        }
        B.buf = inf_uni64(B.buf); B.cnt = inf_uni(B.cnt); B.wi = inf_uni(B.wi);
        INF_REFILL(B);
        inf_u32x16 LT, DT;                                   // (DT: four registers used; a 16-vector so that the look-up is an indexed read, not a select chain)
#pragma unroll
        for (int r = 0; r < 16; r++) LT[r] = g_inf.llut[r * 64 + lane];
#pragma unroll
        for (int r = 0; r < 16; r++) DT[r] = r < 4 ? g_inf.dlut[(r & 3) * 64 + lane] : 0u;
        uint32_t e = inf_lut16(LT, (uint32_t)B.buf & ((1u << INF_LROOT) - 1));
        bool eob = false;
        uint32_t pv[P], pd[P], pl[P], nl = 0;                 // the batch whose source loads are in flight: loaded byte (per lane), destination, length
#pragma unroll
        for (int j = 0; j < P; j++) { pv[j] = 0; pd[j] = 0; pl[j] = 0; }
        while (!eob && status == INF_OK) {
            uint32_t qa[P], qd[P], ql[P];                     // the batch being decoded: source offset (per lane), destination, length
            uint32_t nq = 0, slow_len = 0, slow_dist = 0;
            // (unrolled over the slots of the batch: a slot's registers are named at compile time. A plain loop that described the matches in the
            // lanes of a few registers and built addresses / issued loads from v_readlane afterwards was a third of the code and 25 % SLOWER: a
            // member's decode is one wave's dependent instruction chain, and every instruction added to it shows)
#pragma unroll
            for (int j = 0; j < P; j++) { qa[j] = 0; qd[j] = 0; ql[j] = 0; }
#pragma unroll
            for (int j = 0; j < P; j++) {                     // (every event - end of block, error, a match for the slow path - leaves the batch)
                uint32_t kind = 0;
                // (no end-of-input test per symbol as in k_inflate: past the member's end the reader feeds zeros, and every symbol either advances `pos`,
                // which `cap` bounds, or ends the block - and a block header of zeros is a stored block whose LEN / NLEN do not check. The overrun is
                // reported from the bytes used, behind the loop.)
                for (;;) {                                    // literals, until something else turns up
                    e = inf_uni(e); B.buf = inf_uni64(B.buf); B.cnt = inf_uni(B.cnt); B.wi = inf_uni(B.wi); pos = inf_uni(pos);
                    uint32_t nb = e & 15;
                    if (!nb) {                                // rare: a code longer than the root (INF_LONG has no length), or no such code
                        if (e == INF_LONG) e = inf_uni(inf_walk_inl<0>(B.buf));
                        nb = e & 15;
                        if (!nb) { status = INF_E_CODE; break; }
                    }
                    INF_TAKE(B, nb);
                    kind = (e >> 4) & 3;
                    if (kind) break;
                    if (pos >= cap) { status = INF_E_OUTPUT; break; }
                    if (lane == 0) out[pos] = (uint8_t)(e >> 8);
                    pos++;
                    INF_REFILL(B);
                    e = inf_lut16(LT, inf_uni((uint32_t)B.buf) & ((1u << INF_LROOT) - 1));
                }
                if (status != INF_OK) break;
                if (kind == 2) { eob = true; break; }
                uint32_t eb = (e >> 24) & 15;
                const uint32_t len = ((e >> 8) & 0xFFFFu) + ((uint32_t)B.buf & ((1u << eb) - 1));
                INF_TAKE(B, eb);
                INF_REFILL(B);
                uint32_t d = inf_lut16(DT, inf_uni((uint32_t)B.buf) & ((1u << INF_DROOT) - 1));
                uint32_t nbd = d & 15;
                if (!nbd) {
                    if (d == INF_LONG) d = inf_uni(inf_walk_inl<1>(B.buf));
                    nbd = d & 15;
                    if (!nbd) { status = INF_E_CODE; break; }
                }
                INF_TAKE(B, nbd);
                eb = (d >> 24) & 15;
                const uint32_t dist = ((d >> 8) & 0xFFFFu) + ((uint32_t)B.buf & ((1u << eb) - 1));
                INF_TAKE(B, eb);
                if (dist > pos) { status = INF_E_DIST; break; }
                if (len > cap - pos) { status = INF_E_OUTPUT; break; }
                INF_REFILL(B);
                e = inf_lut16(LT, inf_uni((uint32_t)B.buf) & ((1u << INF_LROOT) - 1));
                const uint32_t src0 = pos - dist;
                if (len <= 64 && src0 + (dist >= len ? len : dist) <= inf_uni(synced)) {
                    uint32_t idx = lane;
                    if (dist < len) idx = lane % dist;        // the source runs into the target: the last `dist` bytes repeat
                    qa[j] = src0 + idx; qd[j] = pos; ql[j] = len; nq = (uint32_t)j + 1;
                    pos += len;
                } else { slow_len = len; slow_dist = dist; break; }
            }
            // the copies of the batch decoded one round ago: their loads have had this round's decoding to come back (one wait, said once in front
            // of all the stores: the compiler would put its own vmcnt(0) in front of every one, behind the store before it; it also covers every
            // store issued before it) ...
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
            for (int j = 0; j < P; j++) if ((uint32_t)j < nl && lane < pl[j]) out[pd[j] + lane] = (uint8_t)pv[j];
            // ... and the loads of this round's batch go out together. When they do, everything stored before the wait above has reached L2: all
            // text except the destinations of the batch just stored and of this one - so the NEXT round's sources have to lie below this batch
#pragma unroll
            for (int j = 0; j < P; j++) {
                if ((uint32_t)j < nq && lane < ql[j]) pv[j] = __builtin_amdgcn_raw_buffer_load_b8(hist, qa[j], 0, 16);      // (sc1: from L2, not from this CU's L1)
                pd[j] = qd[j]; pl[j] = ql[j];
            }
            nl = nq;
            synced = nq ? qd[0] : pos;
            if (slow_len || eob || status != INF_OK) {       // drain: a block ends, or a match wants text that is still on its way
                __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                for (int j = 0; j < P; j++) if ((uint32_t)j < nl && lane < pl[j]) out[pd[j] + lane] = (uint8_t)pv[j];
                nl = 0;
                if (slow_len) {
                    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): everything stored so far is in L2
                    const uint32_t src0 = pos - slow_dist;
                    for (uint32_t i = lane; i < slow_len; i += 64) {
                        const uint8_t b = __hip_atomic_load(out + src0 + (slow_dist >= slow_len ? i : i % slow_dist), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        out[pos + i] = b;
                    }
                    pos += slow_len;
                }
                synced = pos;
            }
        }
    }
    INF_TAKE(B, B.cnt & 7);
    const uint64_t used = (uint64_t)B.wi * 4 - B.cnt / 8 - st.in_off;
    if (status == INF_OK && used > st.in_len) status = INF_E_INPUT;
    if (lane == 0) { InflateResult r; r.status = status; r.blocks = blocks; r.in_used = used; r.out_len = pos; res[blockIdx.x] = r; }
}
#undef INF_REFILL
#undef INF_TAKE

// ---- CRC-32 (IEEE 802.3, reflected) of the produced texts -------------------------------------------------------------------------
constexpr uint32_t CRC_POLY = 0xEDB88320u, CRC_PIECE = 256, CRC_T = 256, CRC_CHUNK = CRC_PIECE * CRC_T;
// product of two polynomials mod P, reflected representation (bit 31 = x^0)
__host__ __device__ inline uint32_t crc_mul(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
static uint32_t crc_xpow8(uint64_t nbytes)      // x^(8 nbytes) mod P
{
    uint32_t r = 0x80000000u, sq = 0x00800000u;  // 1 ; x^8
    for (; nbytes; nbytes >>= 1) { if (nbytes & 1) r = crc_mul(r, sq); sq = crc_mul(sq, sq); }
    return r;
}
struct CrcChunk { uint64_t text_off; int64_t first; uint64_t n; };   // chunk = virtual bytes [first, first + CRC_CHUNK) of a text of n bytes (first < 0: leading zeros)
// raw CRC (register starts at 0, no final xor) of each 64 KB chunk; leading virtual zeros leave the register at 0, so the FIRST chunk of a text is
// the short one and every fold uses a constant multiplier. The chunk is staged through LDS (coalesced byte loads -> rows of 256 bytes, padded
// to 260 so that the per-thread walks hit different banks): 256 threads each walking their own 256 bytes straight out of global memory
// touch 64 cache lines per load instruction and thrash the 32 KB L1 (measured: 93 GB/s).
constexpr uint32_t CRC_ROW = CRC_PIECE + 4;
__global__ __launch_bounds__(CRC_T) void k_crc32_chunks(const uint8_t *__restrict__ text, const CrcChunk *__restrict__ chunks, const uint32_t *__restrict__ kpow, uint32_t *__restrict__ out)
{
    __shared__ uint32_t tab[256];
    __shared__ uint32_t part[CRC_T];
    __shared__ __attribute__((aligned(16))) uint8_t rows[CRC_T * CRC_ROW];
    const uint32_t t = threadIdx.x;
    { uint32_t c = t; for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ CRC_POLY : c >> 1; tab[t] = c; }
    const CrcChunk ch = chunks[blockIdx.x];
    const uint8_t *p = text + ch.text_off;
#pragma unroll 8
    for (uint32_t r = 0; r < CRC_T; r++) {
        const int64_t v = ch.first + (int64_t)(r * CRC_PIECE + t);
        rows[r * CRC_ROW + t] = v >= 0 ? p[v] : (uint8_t)0;
    }
    __syncthreads();
    uint32_t c = 0;
    const uint32_t *mine = (const uint32_t *)&rows[t * CRC_ROW];
#pragma unroll 4
    for (uint32_t j = 0; j < CRC_PIECE / 4; j++) {
        const uint32_t w = mine[j];
        c = tab[(c ^ w) & 0xFF] ^ (c >> 8);
        c = tab[(c ^ (w >> 8)) & 0xFF] ^ (c >> 8);
        c = tab[(c ^ (w >> 16)) & 0xFF] ^ (c >> 8);
        c = tab[(c ^ (w >> 24)) & 0xFF] ^ (c >> 8);
    }
    part[t] = c;
    __syncthreads();
    for (uint32_t j = 0, s2 = 1; s2 < CRC_T; j++, s2 <<= 1) {
        if ((t & (2 * s2 - 1)) == 0) part[t] = crc_mul(part[t], kpow[j]) ^ part[t + s2];
        __syncthreads();
    }
    if (t == 0) out[blockIdx.x] = part[0];
}

// The LDS-window form writes its text out in 16-byte vectors (whole ones, also at the end of the text): fine for texts that start on a 64-byte
// boundary with padding behind them, wrong for the members of a bgzip file, which lie end to end - those take the form with byte-exact stores
static bool inflate_needs_byte_stores(const InflateStream *st, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++) if (st[i].out_off & 15) return true;
    return false;
}
// GS_INFLATE_WINDOW = lds | global | pipe picks the form (measurement aid / tests); by default up to four members per CU take the LDS-window form
// (lowest latency per symbol), more take the window-less form with several matches in flight, whose members interleave five times as densely
constexpr int INF_PIPE = 8;
static void inflate_launch_form(gs_ctx *c, hipStream_t stream, const uint32_t *comp, const InflateStream *streams_host, const InflateStream *ds, uint32_t n, uint8_t *out,
                                InflateResult *dr)
{
    const char *w = getenv("GS_INFLATE_WINDOW");
    int form = n > 4u * (uint32_t)c->n_cu ? 2 : 0;                      // 0 = LDS window, 1 = window-less (one match at a time), 2 = window-less, pipelined
    if (w && !strcmp(w, "lds")) form = 0;
    else if (w && !strcmp(w, "global")) form = 1;
    else if (w && !strcmp(w, "pipe")) form = 2;
    if (form == 0 && inflate_needs_byte_stores(streams_host, n)) form = 2;
    if (form == 2) hipLaunchKernelGGL(k_inflate_pipe<INF_PIPE>, dim3(n), dim3(64), 0, stream, comp, ds, n, out, dr);
    else if (form == 1) hipLaunchKernelGGL(k_inflate<true>, dim3(n), dim3(64), 0, stream, comp, ds, n, out, dr);
    else hipLaunchKernelGGL(k_inflate<false>, dim3(n), dim3(64), 0, stream, comp, ds, n, out, dr);
}
int inflate_streams_dev(gs_ctx *c, const void *comp_dev, const InflateStream *streams, uint32_t n, void *out_dev, InflateResult *results)
{
    if (n == 0) return GS_OK;
    PoolBuf ds(c, 40), dr(c, 41);
    int rc;
    if ((rc = ds.alloc(sizeof(InflateStream) * n)) || (rc = dr.alloc(sizeof(InflateResult) * n))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(ds.p, streams, sizeof(InflateStream) * n, hipMemcpyHostToDevice, c->stream));
    inflate_launch_form(c, c->stream, (const uint32_t *)comp_dev, streams, ds.as<InflateStream>(), n, (uint8_t *)out_dev, dr.as<InflateResult>());
    GS_HIP_CHECK(hipGetLastError());
    GS_HIP_CHECK(hipMemcpyAsync(results, dr.p, sizeof(InflateResult) * n, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(gs::stream_wait(c));
    return GS_OK;
}

// the same launch, asynchronous on `stream`: descriptors and results live in PINNED host memory (a copy from / to pageable memory would make the
// call wait for the stream), nothing is waited for - the caller records an event behind it
int inflate_streams_launch(gs_ctx *c, hipStream_t stream, const void *comp_dev, const InflateStream *streams_pinned, uint32_t n, void *out_dev, void *ds_dev, void *dr_dev,
                           InflateResult *results_pinned)
{
    if (n == 0) return GS_OK;
    GS_HIP_CHECK(hipMemcpyAsync(ds_dev, streams_pinned, sizeof(InflateStream) * n, hipMemcpyHostToDevice, stream));
    inflate_launch_form(c, stream, (const uint32_t *)comp_dev, streams_pinned, (const InflateStream *)ds_dev, n, (uint8_t *)out_dev, (InflateResult *)dr_dev);
    GS_HIP_CHECK(hipGetLastError());
    GS_HIP_CHECK(hipMemcpyAsync(results_pinned, dr_dev, sizeof(InflateResult) * n, hipMemcpyDeviceToHost, stream));
    return GS_OK;
}

int crc32_texts_dev(gs_ctx *c, const void *text_dev, const uint64_t *text_off, const uint64_t *text_len, uint32_t n, uint32_t *crc_out)
{
    std::vector<CrcChunk> chunks; std::vector<uint64_t> first_chunk(n + 1);
    for (uint32_t f = 0; f < n; f++) {
        first_chunk[f] = chunks.size();
        const uint64_t nch = (text_len[f] + CRC_CHUNK - 1) / CRC_CHUNK;
        for (uint64_t k = 0; k < nch; k++) chunks.push_back({text_off[f], (int64_t)text_len[f] - (int64_t)((nch - k) * CRC_CHUNK), text_len[f]});
    }
    first_chunk[n] = chunks.size();
    std::vector<uint32_t> raw(chunks.size());
    if (!chunks.empty()) {
        uint32_t kpow[8];
        for (int j = 0; j < 8; j++) kpow[j] = crc_xpow8((uint64_t)CRC_PIECE << j);
        PoolBuf dc(c, 42), dk(c, 43), dout(c, 44);
        int rc;
        if ((rc = dc.alloc(sizeof(CrcChunk) * chunks.size())) || (rc = dk.alloc(sizeof kpow)) || (rc = dout.alloc(4 * chunks.size()))) return rc;
        GS_HIP_CHECK(hipMemcpyAsync(dc.p, chunks.data(), sizeof(CrcChunk) * chunks.size(), hipMemcpyHostToDevice, c->stream));
        GS_HIP_CHECK(hipMemcpyAsync(dk.p, kpow, sizeof kpow, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_crc32_chunks, dim3((uint32_t)chunks.size()), dim3(CRC_T), 0, c->stream, (const uint8_t *)text_dev, dc.as<CrcChunk>(), dk.as<uint32_t>(), dout.as<uint32_t>());
        GS_HIP_CHECK(hipGetLastError());
        GS_HIP_CHECK(hipMemcpyAsync(raw.data(), dout.p, 4 * chunks.size(), hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(gs::stream_wait(c));
    }
    const uint32_t kchunk = crc_xpow8(CRC_CHUNK);
    for (uint32_t f = 0; f < n; f++) {
        uint32_t r = 0;
        for (uint64_t k = first_chunk[f]; k < first_chunk[f + 1]; k++) r = crc_mul(r, kchunk) ^ raw[k];
        // register preset to all ones = + 0xFFFFFFFF * x^(8 n); final complement
        crc_out[f] = r ^ crc_mul(0xFFFFFFFFu, crc_xpow8(text_len[f])) ^ 0xFFFFFFFFu;
    }
    return GS_OK;
}

// ---- record boundaries of texts that exist only on the device (the counterpart of gs_fasta_scan, gs_ingest.hip) -------------------------
constexpr uint32_t SC_T = 256, SC_CHUNK = 16384;
struct ScanChunk { uint64_t begin, end, file_begin; };
// '>' at the start of a line = a record (a header line); positions appended in any order, the host sorts them
__global__ __launch_bounds__(SC_T) void k_fasta_starts(const uint8_t *__restrict__ text, const ScanChunk *__restrict__ chunks, uint64_t *__restrict__ starts, uint32_t cap,
                                                       uint32_t *__restrict__ count)
{
    const ScanChunk ch = chunks[blockIdx.x];
    for (uint64_t i = ch.begin + threadIdx.x; i < ch.end; i += SC_T) {
        if (text[i] == '>' && (i == ch.file_begin || text[i - 1] == '\n')) {
            const uint32_t k = atomicAdd(count, 1u);
            if (k < cap) starts[k] = i;
        }
    }
}
// per record: end of its header line (the '\n', or the end of the file) and whether the line contains "capsid" (dnafiles.rs:62-67)
__global__ __launch_bounds__(SC_T) void k_fasta_headers(const uint8_t *__restrict__ text, const uint64_t *__restrict__ starts, const uint64_t *__restrict__ file_end, uint64_t n,
                                                        uint64_t *__restrict__ hend, uint8_t *__restrict__ capsid)
{
    const uint64_t r = (uint64_t)blockIdx.x * SC_T + threadIdx.x;
    if (r >= n) return;
    const uint64_t h0 = starts[r] + 1, fe = file_end[r];
    uint64_t i = h0;
    uint64_t w = 0;                                  // the last six bytes of the line so far
    bool hit = false;
    while (i < fe && text[i] != '\n') {
        w = ((w << 8) | text[i]) & 0xFFFFFFFFFFFFull;
        hit |= w == 0x636170736964ull;                // "capsid": a trailing '\r' never completes it, so CRLF needs no special case
        i++;
    }
    hend[r] = i;
    capsid[r] = hit;
}

// sequence ranges [sb, se) of the records of n texts (text f = bytes [off[f], off[f] + len[f]) of text_dev), capsid records dropped: the same
// answer as gs_fasta_scan(text, skip_capsid = 1) on each text, offsets absolute in text_dev
int fasta_scan_dev(gs_ctx *c, const void *text_dev, const uint64_t *off, const uint64_t *len, uint32_t n, std::vector<std::vector<uint64_t>> &sb,
                   std::vector<std::vector<uint64_t>> &se)
{
    sb.assign(n, {}); se.assign(n, {});
    std::vector<ScanChunk> chunks;
    uint64_t total = 0;
    for (uint32_t f = 0; f < n; f++) {
        total += len[f];
        for (uint64_t b = 0; b < len[f]; b += SC_CHUNK) chunks.push_back({off[f] + b, off[f] + std::min<uint64_t>(b + SC_CHUNK, len[f]), off[f]});
    }
    if (chunks.empty()) return GS_OK;
    PoolBuf dch(c, 42), dst(c, 43), dcn(c, 44);
    int rc;
    if ((rc = dch.alloc(sizeof(ScanChunk) * chunks.size())) || (rc = dcn.alloc(16))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(dch.p, chunks.data(), sizeof(ScanChunk) * chunks.size(), hipMemcpyHostToDevice, c->stream));
    uint32_t cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(65536, total / 256), (uint64_t)1 << 30), cnt = 0;
    for (int pass = 0; pass < 2; pass++) {
        if ((rc = dst.alloc(8 * (size_t)cap))) return rc;
        GS_HIP_CHECK(hipMemsetAsync(dcn.p, 0, 16, c->stream));
        hipLaunchKernelGGL(k_fasta_starts, dim3((uint32_t)chunks.size()), dim3(SC_T), 0, c->stream, (const uint8_t *)text_dev, dch.as<ScanChunk>(), dst.as<uint64_t>(), cap, dcn.as<uint32_t>());
        GS_HIP_CHECK(hipGetLastError());
        GS_HIP_CHECK(hipMemcpyAsync(&cnt, dcn.p, 4, hipMemcpyDeviceToHost, c->stream));
        GS_HIP_CHECK(gs::stream_wait(c));
        if (cnt <= cap) break;
        cap = cnt;                                   // more records than guessed (short protein records): once more with room for all
    }
    if (cnt == 0) return GS_OK;
    std::vector<uint64_t> starts(cnt), fend(cnt), hend(cnt);
    std::vector<uint8_t> cap_flag(cnt);
    GS_HIP_CHECK(hipMemcpyAsync(starts.data(), dst.p, 8 * (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(gs::stream_wait(c));
    std::sort(starts.begin(), starts.end());
    {   // the file of every record (texts are disjoint and in increasing offset order is NOT assumed: binary search over sorted file ranges)
        std::vector<uint32_t> order(n);
        for (uint32_t f = 0; f < n; f++) order[f] = f;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return off[a] < off[b]; });
        size_t k = 0;
        for (uint32_t o = 0; o < n && k < cnt; o++) {
            const uint32_t f = order[o];
            while (k < cnt && starts[k] < off[f] + len[f]) { fend[k] = off[f] + len[f]; k++; }
        }
    }
    PoolBuf dfe(c, 40), dhe(c, 41), dcf(c, 45);
    if ((rc = dfe.alloc(8 * (size_t)cnt)) || (rc = dhe.alloc(8 * (size_t)cnt)) || (rc = dcf.alloc(cnt))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(dst.p, starts.data(), 8 * (size_t)cnt, hipMemcpyHostToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(dfe.p, fend.data(), 8 * (size_t)cnt, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_fasta_headers, dim3((cnt + SC_T - 1) / SC_T), dim3(SC_T), 0, c->stream, (const uint8_t *)text_dev, dst.as<uint64_t>(), dfe.as<uint64_t>(), (uint64_t)cnt,
                       dhe.as<uint64_t>(), dcf.as<uint8_t>());
    GS_HIP_CHECK(hipGetLastError());
    GS_HIP_CHECK(hipMemcpyAsync(hend.data(), dhe.p, 8 * (size_t)cnt, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync(cap_flag.data(), dcf.p, cnt, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(gs::stream_wait(c));
    // records in text order per file
    std::vector<std::pair<uint64_t, uint32_t>> ranges(n);
    for (uint32_t f = 0; f < n; f++) ranges[f] = {off[f], f};
    std::sort(ranges.begin(), ranges.end());
    size_t k = 0;
    for (uint32_t o = 0; o < n; o++) {
        const uint32_t f = ranges[o].second;
        const uint64_t fe = off[f] + len[f];
        while (k < cnt && starts[k] < fe) {
            const uint64_t next = (k + 1 < cnt && starts[k + 1] < fe) ? starts[k + 1] : fe;
            if (!cap_flag[k]) { sb[f].push_back(hend[k] < fe ? hend[k] + 1 : fe); se[f].push_back(next); }
            k++;
        }
    }
    return GS_OK;
}

// RFC 1952 member header: returns its length, 0 when the member is not one this path takes (not gzip/deflate, reserved flags, truncated)
size_t gzip_header_len(const uint8_t *p, size_t n)
{
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    const uint8_t flg = p[3];
    size_t h = 10;
    if (flg & 4) { if (h + 2 > n) return 0; h += 2 + ((size_t)p[h] | (size_t)p[h + 1] << 8); }
    if (flg & 8) { while (h < n && p[h]) h++; h++; }
    if (flg & 16) { while (h < n && p[h]) h++; h++; }
    if (flg & 2) h += 2;
    return h + 8 <= n ? h : 0;
}

}  // namespace gs

extern "C" {

/* Inflate n single-member gzip buffers on the device and check each against its trailer (CRC-32, ISIZE): the building block of
 * gs_sketch_files' .gz path, exposed for parity tests against zlib (files.rs:258-341 reads .gz through needletail / flate2).
 * in[i]/in_len[i]: HOST gzip bytes; out[i]/out_cap[i]: HOST buffers for the text; out_len[i]: bytes produced; status[i]: 0 = ok,
 * 1..8 = malformed deflate data (INF_E_*), 100 = header not taken, 101 = trailing bytes after the member (multi-member file),
 * 102 = ISIZE mismatch, 103 = CRC mismatch, 104 = out_cap too small. Returns GS_OK when the call itself worked. */
int gs_gunzip_batch(gs_ctx *c, const uint8_t *const *in, const uint64_t *in_len, uint64_t n, uint8_t *const *out, const uint64_t *out_cap, uint64_t *out_len,
                    int *status)
{
    GS_REQUIRE(c && (n == 0 || (in && in_len && out && out_cap && out_len && status)), GS_ERR_INVALID, "null argument");
    GS_REQUIRE(n < ((uint64_t)1 << 24), GS_ERR_INVALID, "too many members");
    if (n == 0) return GS_OK;
    GS_CTX_LOCK(c);
    std::vector<gs::InflateStream> st; std::vector<uint64_t> who, coff(n, 0), isize(n, 0);
    uint64_t ctot = 0, otot = 0;
    for (uint64_t i = 0; i < n; i++) {
        out_len[i] = 0; status[i] = 100;
        const size_t h = in[i] ? gs::gzip_header_len(in[i], in_len[i]) : 0;
        if (!h) continue;
        const uint8_t *t = in[i] + in_len[i] - 4;
        isize[i] = (uint64_t)t[0] | (uint64_t)t[1] << 8 | (uint64_t)t[2] << 16 | (uint64_t)t[3] << 24;
        if (isize[i] > out_cap[i] || isize[i] >= ((uint64_t)1 << 30)) { status[i] = 104; continue; }
        GS_REQUIRE(ctot + in_len[i] < ((uint64_t)12 << 30), GS_ERR_INVALID, "gs_gunzip_batch: more than 12 GiB of members in one call");
        coff[i] = ctot;
        st.push_back({ctot + h, in_len[i] - h, otot, isize[i]});
        who.push_back(i);
        ctot += gs::round_up(in_len[i], 64); otot += gs::round_up(isize[i], 64);
    }
    if (st.empty()) return GS_OK;
    gs::PoolBuf dcomp(c, 45), dtext(c, 46);
    int rc;
    if ((rc = dcomp.alloc(ctot + 64)) || (rc = dtext.alloc(otot + 64))) return rc;
    for (uint64_t k = 0; k < who.size(); k++) GS_HIP_CHECK(hipMemcpyAsync((uint8_t *)dcomp.p + coff[who[k]], in[who[k]], in_len[who[k]], hipMemcpyHostToDevice, c->stream));
    std::vector<gs::InflateResult> res(st.size());
    if ((rc = gs::inflate_streams_dev(c, dcomp.p, st.data(), (uint32_t)st.size(), dtext.p, res.data()))) return rc;
    std::vector<uint64_t> toff(st.size()), tlen(st.size());
    for (uint64_t k = 0; k < st.size(); k++) { toff[k] = st[k].out_off; tlen[k] = res[k].status == 0 ? res[k].out_len : 0; }
    std::vector<uint32_t> crc(st.size());
    if ((rc = gs::crc32_texts_dev(c, dtext.p, toff.data(), tlen.data(), (uint32_t)st.size(), crc.data()))) return rc;
    for (uint64_t k = 0; k < st.size(); k++) {
        const uint64_t i = who[k];
        const uint8_t *t = in[i] + in_len[i] - 8;
        const uint32_t want = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
        if (res[k].status) status[i] = (int)res[k].status;
        else if (res[k].in_used + 8 != st[k].in_len) status[i] = 101;
        else if (res[k].out_len != isize[i]) status[i] = 102;
        else if (crc[k] != want) status[i] = 103;
        else status[i] = 0;
        out_len[i] = res[k].out_len;
        if (res[k].out_len) GS_HIP_CHECK(hipMemcpyAsync(out[i], (const uint8_t *)dtext.p + st[k].out_off, res[k].out_len, hipMemcpyDeviceToHost, c->stream));
    }
    GS_HIP_CHECK(gs::stream_wait(c));
    return GS_OK;
}

}  // extern "C"
