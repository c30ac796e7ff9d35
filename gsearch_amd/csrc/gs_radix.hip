// gs_radix.hip — stable LSD radix sort of 64-bit keys and run-length encoding of a sorted array, for the sorted (fallback) form of the
// ProbMinHash3a sketcher (gs_sketch.hip run_prob_sorted: the genomes the bucketed form does not suit - fewer than 64 k-mers per slot, more than
// 25 M k-mers, an overfull bucket). Multiplicities w(v) of SPEC 3.3 are the run lengths of the sorted (genome, value) keys
// (/root/reference/src/dna/dnasketch.rs:499-518 counts them in a hash map before ProbMinHash3a::hash_weighted...).
// Written for wave64: one wavefront owns a tile of RS_TILE consecutive keys in both passes, so stability needs no cross-wave ordering -
// the tile is walked 64 keys at a time, lanes of equal digit find one another with eight ballots, and the tile's 256 running offsets sit in LDS.
#include <algorithm>
#include "gs_internal.hpp"

namespace gs {

constexpr uint32_t RS_TILE = 16384;        // keys per wavefront (the per-tile digit counts are scanned by one workgroup: fewer, larger tiles)

// per tile: how many keys carry each value of the 8-bit digit at `shift`   (hist[d * ntiles + tile])
__global__ __launch_bounds__(64) void k_radix_hist(const uint64_t *__restrict__ in, uint64_t n, uint32_t shift, uint32_t ntiles, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) h[i] = 0;
    __syncthreads();
    const uint64_t t0 = (uint64_t)blockIdx.x * RS_TILE, t1 = t0 + RS_TILE < n ? t0 + RS_TILE : n;
    for (uint64_t i = t0 + lane; i < t1; i += 64) atomicAdd(&h[(uint32_t)(in[i] >> shift) & 255u], 1u);
    __syncthreads();
    for (uint32_t i = lane; i < 256; i += 64) hist[(uint64_t)i * ntiles + blockIdx.x] = h[i];
}
// exclusive prefix sum of `cnt` 32-bit values in place, by ONE workgroup (lanes take contiguous stretches; the arrays here are a few million
// entries: 256 digits x tiles, or one count per tile). total_out (optional) receives the sum.
__global__ __launch_bounds__(1024) void k_scan_u32(uint32_t *__restrict__ a, uint64_t cnt, uint32_t *__restrict__ total_out)
{
    __shared__ uint32_t part[1024];
    const uint64_t per = ((cnt + 1023) / 1024 + 3) & ~(uint64_t)3, b = (uint64_t)threadIdx.x * per, e = b + per < cnt ? b + per : cnt;     // stretches of whole uint4s
    uint32_t s = 0;
    uint64_t i = b;
    for (; i + 4 <= e; i += 4) { const uint4 v = *(const uint4 *)(a + i); s += v.x + v.y + v.z + v.w; }
    for (; i < e; i++) s += a[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 1024; o <<= 1) {            // Hillis-Steele over the 1024 partial sums
        const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
    for (i = b; i + 4 <= e; i += 4) {
        uint4 v = *(const uint4 *)(a + i);
        const uint32_t x0 = run, x1 = x0 + v.x, x2 = x1 + v.y, x3 = x2 + v.z;
        run = x3 + v.w;
        *(uint4 *)(a + i) = make_uint4(x0, x1, x2, x3);
    }
    for (; i < e; i++) { const uint32_t v = a[i]; a[i] = run; run += v; }
    if (total_out && threadIdx.x == 1023) *total_out = part[1023];
}
// lanes of the wavefront that hold the same 8-bit digit as this one
__device__ __forceinline__ uint64_t same_digit_lanes(uint32_t d, bool valid)
{
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const uint64_t bal = __ballot((d >> b) & 1u);
        m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    return m;
}
__global__ __launch_bounds__(64) void k_radix_scatter(const uint64_t *__restrict__ in, uint64_t n, uint32_t shift, uint32_t ntiles, const uint32_t *__restrict__ offs,
                                                      uint64_t *__restrict__ out)
{
    __shared__ uint32_t o[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) o[i] = offs[(uint64_t)i * ntiles + blockIdx.x];
    __syncthreads();
    const uint64_t t0 = (uint64_t)blockIdx.x * RS_TILE, t1 = t0 + RS_TILE < n ? t0 + RS_TILE : n;
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    for (uint64_t s = t0; s < t1; s += 64) {
        const uint64_t i = s + lane;
        const bool valid = i < t1;
        const uint64_t key = valid ? in[i] : 0;
        const uint32_t d = (uint32_t)(key >> shift) & 255u;
        const uint64_t grp = same_digit_lanes(d, valid);
        const uint32_t base = valid ? o[d] : 0u;
        __syncthreads();                                         // every lane has read its digit's offset before any leader moves it on
        if (valid) {
            const uint32_t rank = (uint32_t)__popcll(grp & lt);
            if (rank == 0) o[d] = base + (uint32_t)__popcll(grp);
            out[base + rank] = key;
        }
        __syncthreads();
    }
}
// sorted -> (unique keys, run lengths): per tile the number of run heads, scanned, then heads write their key and position
__global__ __launch_bounds__(64) void k_rle_count(const uint64_t *__restrict__ a, uint64_t n, uint32_t *__restrict__ tile_heads)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * RS_TILE, t1 = t0 + RS_TILE < n ? t0 + RS_TILE : n;
    uint32_t c = 0;
    for (uint64_t i = t0 + lane; i < t1; i += 64) c += (i == 0 || a[i] != a[i - 1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if (lane == 0) tile_heads[blockIdx.x] = c;
}
__global__ __launch_bounds__(64) void k_rle_write(const uint64_t *__restrict__ a, uint64_t n, const uint32_t *__restrict__ tile_base, uint64_t *__restrict__ uniq, uint32_t *__restrict__ pos)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * RS_TILE, t1 = t0 + RS_TILE < n ? t0 + RS_TILE : n;
    const uint64_t lt = ((uint64_t)1 << lane) - 1;
    uint32_t run = tile_base[blockIdx.x];
    for (uint64_t s = t0; s < t1; s += 64) {
        const uint64_t i = s + lane;
        const bool head = i < t1 && (i == 0 || a[i] != a[i - 1]);
        const uint64_t bal = __ballot(head);
        if (head) { const uint32_t r = run + (uint32_t)__popcll(bal & lt); uniq[r] = a[i]; pos[r] = (uint32_t)i; }
        run += (uint32_t)__popcll(bal);
    }
}
__global__ void k_rle_lengths(const uint32_t *__restrict__ pos, const uint32_t *__restrict__ nruns, uint64_t n, uint32_t *__restrict__ len)
{
    const uint32_t nr = *nruns;
    for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nr; r += (uint64_t)gridDim.x * blockDim.x)
        len[r] = (r + 1 < nr ? pos[r + 1] : (uint32_t)n) - pos[r];
}

size_t radix_scratch_bytes(uint64_t n) { const uint64_t nt = (n + RS_TILE - 1) / RS_TILE; return (size_t)(4 * (256 * nt + 64)); }

// keys[0..n) sorted ascending on bits [0, endbit) (stable); `alt` is a second buffer of n keys; *sorted_out = whichever of the two holds the result
int radix_sort_u64(gs_ctx *c, uint64_t *keys, uint64_t *alt, uint64_t n, int endbit, void *scratch, uint64_t **sorted_out)
{
    *sorted_out = keys;
    if (n < 2) return GS_OK;
    GS_REQUIRE(n < ((uint64_t)1 << 32), GS_ERR_UNSUPPORTED, "radix_sort_u64: more than 2^32 keys");
    const uint32_t nt = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    uint32_t *hist = (uint32_t *)scratch;
    uint64_t *src = keys, *dst = alt;
    for (int shift = 0; shift < endbit; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(nt), dim3(64), 0, c->stream, src, n, (uint32_t)shift, nt, hist);
        hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, c->stream, hist, (uint64_t)256 * nt, (uint32_t *)nullptr);
        hipLaunchKernelGGL(k_radix_scatter, dim3(nt), dim3(64), 0, c->stream, src, n, (uint32_t)shift, nt, hist, dst);
        GS_HIP_CHECK(hipGetLastError());
        std::swap(src, dst);
    }
    *sorted_out = src;
    return GS_OK;
}
// sorted[0..n) -> uniq[0..r), len[0..r), *nruns_dev = r. `pos`: n x 4 bytes of scratch (head positions); scratch as for the sort.
int run_length_encode_u64(gs_ctx *c, const uint64_t *sorted, uint64_t n, uint64_t *uniq, uint32_t *len, uint32_t *nruns_dev, uint32_t *pos, void *scratch)
{
    GS_REQUIRE(n >= 1 && n < ((uint64_t)1 << 32), GS_ERR_UNSUPPORTED, "run_length_encode_u64: bad length");
    const uint32_t nt = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    uint32_t *heads = (uint32_t *)scratch;
    hipLaunchKernelGGL(k_rle_count, dim3(nt), dim3(64), 0, c->stream, sorted, n, heads);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(1024), 0, c->stream, heads, (uint64_t)nt, nruns_dev);
    hipLaunchKernelGGL(k_rle_write, dim3(nt), dim3(64), 0, c->stream, sorted, n, heads, uniq, pos);
    hipLaunchKernelGGL(k_rle_lengths, dim3(c->n_cu * 8), dim3(256), 0, c->stream, pos, nruns_dev, n, len);
    GS_HIP_CHECK(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
