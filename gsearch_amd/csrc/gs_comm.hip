// gs_comm.hip — multi-GPU exchange of the path behind the C ABI, for hosts without torch (the reference's host is Rust):
// one process per GPU, one RCCL communicator over xGMI, and the only collective the path has - the all-gather of the per-rank
// top-k blocks (conceptual ancestor: the per-shard loop + merge of /root/reference/scripts/multiple_search.sh:71-107).
// RCCL is loaded lazily (dlopen) when the first communicator is made: the sketch / distance / index entry points do not depend on
// it, and a process that already carries an RCCL (e.g. through torch.distributed) is not disturbed unless it asks for this one.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <stdlib.h>
#include <mutex>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include "gs_internal.hpp"

namespace gs {

// the few RCCL entry points used, with the types of /opt/rocm/include/rccl/rccl.h (NCCL_UNIQUE_ID_BYTES = 128, ncclChar = 0)
struct NcclId { char internal[128]; };
typedef int (*fn_get_id)(NcclId *);
typedef int (*fn_init_rank)(void **comm, int nranks, NcclId id, int rank);
typedef int (*fn_destroy)(void *comm);
typedef int (*fn_allgather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t stream);
typedef const char *(*fn_errstr)(int);
struct Rccl {
    void *h = nullptr; fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_destroy destroy = nullptr; fn_allgather allgather = nullptr;
    fn_errstr errstr = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;
static int rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.h) return GS_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    // an RCCL the process already carries (torch.distributed's) is taken as it is - one instance per process -, else the system's is loaded
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD); if (h) break; }
    if (!h) for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    GS_REQUIRE(h, GS_ERR_UNSUPPORTED, "RCCL (librccl.so) cannot be loaded: %s", dlerror());
    Rccl r; r.h = h;
    r.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId"); r.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy"); r.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    GS_REQUIRE(r.get_id && r.init_rank && r.destroy && r.allgather, GS_ERR_UNSUPPORTED, "librccl.so lacks an expected entry point");
    if (getenv("GS_COMM_VERBOSE")) { Dl_info di; if (dladdr((void *)r.allgather, &di) && di.dli_fname) fprintf(stderr, "[GS_COMM] RCCL entry points from %s\n", di.dli_fname); }
    g_rccl = r;
    return GS_OK;
}
#define GS_NCCL_CHECK(expr)                                                                                             \
    do {                                                                                                                \
        int _r = (expr);                                                                                                \
        if (_r != 0) { gs::set_error("%s failed: %s", #expr, gs::g_rccl.errstr ? gs::g_rccl.errstr(_r) : "rccl error"); return GS_ERR_HIP; } \
    } while (0)

}  // namespace gs

struct gs_comm {
    gs_ctx *ctx = nullptr; void *comm = nullptr; int n_ranks = 1, rank = 0;
    gs::DevBuf send, recv;
    uint64_t *pending_counts = nullptr;      // counts + bad-shape word of the last exchange (in recv), for gs_comm_wait
};

extern "C" {

int gs_comm_unique_id(void *id_out_128)
{
    GS_REQUIRE(id_out_128, GS_ERR_INVALID, "null argument");
    int rc = gs::rccl_load(); if (rc) return rc;
    gs::NcclId id;
    GS_NCCL_CHECK(gs::g_rccl.get_id(&id));
    memcpy(id_out_128, &id, 128);
    return GS_OK;
}

int gs_comm_create(gs_ctx *c, int n_ranks, int rank, const void *id_128, gs_comm **out)
{
    GS_REQUIRE(c && out && id_128 && n_ranks >= 1 && rank >= 0 && rank < n_ranks, GS_ERR_INVALID, "bad argument");
    int rc = gs::rccl_load(); if (rc) return rc;
    GS_HIP_CHECK(hipSetDevice(c->device));
    gs::NcclId id; memcpy(&id, id_128, 128);
    gs_comm *m = new gs_comm();
    m->ctx = c; m->n_ranks = n_ranks; m->rank = rank;
    int r = gs::g_rccl.init_rank(&m->comm, n_ranks, id, rank);
    if (r != 0) { gs::set_error("ncclCommInitRank failed: %s", gs::g_rccl.errstr ? gs::g_rccl.errstr(r) : "rccl error"); delete m; return GS_ERR_HIP; }
    *out = m;
    return GS_OK;
}

void gs_comm_destroy(gs_comm *m)
{
    if (!m) return;
    if (m->comm && gs::g_rccl.destroy) { (void)hipSetDevice(m->ctx->device); (void)hipStreamSynchronize(m->ctx->stream); (void)gs::g_rccl.destroy(m->comm); }
    delete m;
}
int gs_comm_rank(const gs_comm *m) { return m ? m->rank : -1; }
int gs_comm_size(const gs_comm *m) { return m ? m->n_ranks : 0; }

/* ---- block layout of the exchange (host helpers: a host that ships the blocks by its own means - MPI, sockets, the gloo tests - packs and unpacks
 * with these; the device path below uses the same layout). One block per rank, gs_topk_block_bytes(nq_max, knbn) bytes whatever the rank holds:
 *   [0,8) nq_local  [8,12) knbn  [12,16) magic "GSTK"  | nq_max x knbn ids (u64) | nq_max x knbn distances (f32) | padding to 16 bytes
 * Ranks may hold DIFFERENT numbers of queries (contiguous shards differ by one; a rank may hold none): rows beyond nq_local are padding. */
#define GS_TOPK_MAGIC 0x4B545347u
uint64_t gs_topk_block_bytes(uint64_t nq_max, uint32_t knbn) { return (16 + nq_max * knbn * 12 + 15) & ~(uint64_t)15; }
int gs_topk_pack(const uint64_t *ids, const float *dist, uint64_t nq_local, uint64_t nq_max, uint32_t knbn, void *block_out)
{
    GS_REQUIRE(block_out && nq_local <= nq_max && (nq_local == 0 || (ids && dist)), GS_ERR_INVALID, "gs_topk_pack: bad argument");
    uint8_t *b = (uint8_t *)block_out;
    memset(b, 0, gs_topk_block_bytes(nq_max, knbn));
    const uint32_t magic = GS_TOPK_MAGIC;
    memcpy(b, &nq_local, 8); memcpy(b + 8, &knbn, 4); memcpy(b + 12, &magic, 4);
    if (nq_local) { memcpy(b + 16, ids, nq_local * knbn * 8); memcpy(b + 16 + nq_max * knbn * 8, dist, nq_local * knbn * 4); }
    return GS_OK;
}
int gs_topk_unpack(const void *blocks, int n_ranks, uint64_t nq_max, uint32_t knbn, uint64_t *all_ids, float *all_dist, uint64_t *counts_out)
{
    GS_REQUIRE(blocks && n_ranks >= 1, GS_ERR_INVALID, "gs_topk_unpack: bad argument");
    const uint64_t bb = gs_topk_block_bytes(nq_max, knbn);
    uint64_t row = 0;
    for (int r = 0; r < n_ranks; r++) {
        const uint8_t *b = (const uint8_t *)blocks + (uint64_t)r * bb;
        uint64_t nq; uint32_t k, magic;
        memcpy(&nq, b, 8); memcpy(&k, b + 8, 4); memcpy(&magic, b + 12, 4);
        GS_REQUIRE(magic == GS_TOPK_MAGIC && k == knbn && nq <= nq_max, GS_ERR_INVALID, "gs_topk_unpack: block of rank %d is not a top-k block of this shape", r);
        if (counts_out) counts_out[r] = nq;
        if (nq && all_ids) memcpy(all_ids + row * knbn, b + 16, nq * knbn * 8);
        if (nq && all_dist) memcpy(all_dist + row * knbn, b + 16 + nq_max * knbn * 8, nq * knbn * 4);
        row += nq;
    }
    return GS_OK;
}

}  // extern "C"

namespace gs {
__global__ void k_topk_pack(const uint64_t *__restrict__ ids, const float *__restrict__ dist, uint64_t nq_local, uint64_t nq_max, uint32_t knbn, uint8_t *__restrict__ block)
{
    const uint64_t n = nq_local * knbn, i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { ((uint64_t *)block)[0] = nq_local; ((uint32_t *)block)[2] = knbn; ((uint32_t *)block)[3] = GS_TOPK_MAGIC; }
    if (i < n) { ((uint64_t *)(block + 16))[i] = ids[i]; ((float *)(block + 16 + nq_max * knbn * 8))[i] = dist[i]; }
}
// every rank's rows to their place of the compact (sum of counts) x knbn answer, rank order; counts[r] for the host. Bad headers raise *bad.
__global__ void k_topk_unpack(const uint8_t *__restrict__ blocks, uint64_t block_bytes, int n_ranks, uint64_t nq_max, uint32_t knbn, uint64_t *__restrict__ all_ids,
                              float *__restrict__ all_dist, uint64_t *__restrict__ counts, uint32_t *__restrict__ bad)
{
    const int r = blockIdx.y;
    const uint8_t *b = blocks + (uint64_t)r * block_bytes;
    // EVERY header is checked before anything is scattered: a block of another shape would otherwise move the rows of the ranks behind it
    // past the end of all_ids / all_dist before the host ever sees the flag. A bad header anywhere: nobody writes.
    uint64_t row0 = 0; bool any_bad = false;
    for (int q = 0; q < n_ranks; q++) {
        const uint8_t *h = blocks + (uint64_t)q * block_bytes;
        const uint64_t nqq = *(const uint64_t *)h;
        any_bad |= ((const uint32_t *)h)[2] != knbn || ((const uint32_t *)h)[3] != GS_TOPK_MAGIC || nqq > nq_max;
        if (q < r) row0 += nqq;
    }
    if (any_bad) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(bad, 1u); return; }
    const uint64_t nq = *(const uint64_t *)b;
    if (threadIdx.x == 0 && blockIdx.x == 0) counts[r] = nq;
    const uint64_t n = nq * knbn;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        all_ids[row0 * knbn + i] = ((const uint64_t *)(b + 16))[i];
        all_dist[row0 * knbn + i] = ((const float *)(b + 16 + nq_max * knbn * 8))[i];
    }
}
// DB-sharded alternative (scripts/multiple_search.sh:71-107: every shard answers ALL queries, the answers are merged): per query the knbn_out best of
// the S lists of knbn_in under (distance, id). One workgroup per query; keys in LDS, every key ranks itself against the others (S * knbn_in <= a few
// hundred). Distances are non-negative floats (+inf in unused slots), so their bit patterns order like the values.
constexpr int MG_T = 256;
__global__ __launch_bounds__(MG_T) void k_topk_merge(const uint64_t *__restrict__ ids, const float *__restrict__ dist, uint32_t S, uint64_t nq, uint32_t kin, const uint64_t *__restrict__ id_off,
                                                     uint32_t kout, uint64_t *__restrict__ out_ids, float *__restrict__ out_dist)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_mg[];
    const uint32_t N = S * kin;
    uint64_t *sid = (uint64_t *)s_mg; uint32_t *sd = (uint32_t *)(sid + N);
    const uint64_t q = blockIdx.x;
    for (uint32_t t = threadIdx.x; t < N; t += MG_T) {
        const uint32_t sh = t / kin, j = t % kin;
        const uint64_t src = ((uint64_t)sh * nq + q) * kin + j;
        uint64_t id = ids[src];
        if (id != ~(uint64_t)0 && id_off) id += id_off[sh];
        sid[t] = id; sd[t] = __float_as_uint(dist[src]);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < N; t += MG_T) {
        const uint64_t id = sid[t]; const uint32_t d = sd[t];
        uint32_t rank = 0;
        for (uint32_t u = 0; u < N; u++) rank += (sd[u] < d) || (sd[u] == d && (sid[u] < id || (sid[u] == id && u < t)));
        if (rank < kout) { out_ids[q * kout + rank] = id; out_dist[q * kout + rank] = __uint_as_float(d); }
    }
    for (uint32_t t = N + threadIdx.x; t < kout; t += MG_T) { out_ids[q * kout + t] = ~(uint64_t)0; out_dist[q * kout + t] = INFINITY; }
}
}  // namespace gs

extern "C" {

/* Ranks with DIFFERENT numbers of queries (round 5): rank r contributes nq_local rows (its block of a contiguous sharding: sizes differ by one, none is
 * allowed), every rank passes the same nq_max >= all of them, and receives the compact concatenation in rank order - (sum of the counts) x knbn - plus
 * every rank's count (counts_out: HOST, n_ranks entries, optional). ONE ncclAllGather of the fixed-size packed blocks (gs_topk_block_bytes) on the
 * context's stream between a pack and an unpack kernel. all_*_dev must hold n_ranks * nq_max rows. */
static int allgatherv_launch(gs_comm *m, const uint64_t *ids_dev, const float *dist_dev, uint64_t nq_local, uint64_t nq_max, uint32_t knbn, uint64_t *all_ids_dev, float *all_dist_dev,
                             uint64_t *counts_dev)
{
    gs_ctx *c = m->ctx;
    GS_HIP_CHECK(hipSetDevice(c->device));
    const uint64_t block = gs_topk_block_bytes(nq_max, knbn);
    int rc;
    if ((rc = m->send.ensure(block))) return rc;
    if ((rc = m->recv.ensure(block * (size_t)m->n_ranks + 8 * (size_t)m->n_ranks + 64))) return rc;
    uint64_t *d_counts = (uint64_t *)((uint8_t *)m->recv.p + block * (size_t)m->n_ranks);
    uint32_t *d_bad = (uint32_t *)(d_counts + m->n_ranks);
    m->pending_counts = d_counts;
    GS_HIP_CHECK(hipMemsetAsync(d_counts, 0, 8 * (size_t)m->n_ranks + 8, c->stream));
    const uint64_t n = std::max<uint64_t>(nq_local * knbn, 1);
    hipLaunchKernelGGL(gs::k_topk_pack, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, ids_dev, dist_dev, nq_local, nq_max, knbn, (uint8_t *)m->send.p);
    GS_HIP_CHECK(hipGetLastError());
    if (m->n_ranks > 1) GS_NCCL_CHECK(gs::g_rccl.allgather(m->send.p, m->recv.p, block, /*ncclChar*/ 0, m->comm, c->stream));
    else GS_HIP_CHECK(hipMemcpyAsync(m->recv.p, m->send.p, block, hipMemcpyDeviceToDevice, c->stream));
    const uint32_t gx = (uint32_t)std::min<uint64_t>((nq_max * knbn + 255) / 256, 256);
    hipLaunchKernelGGL(gs::k_topk_unpack, dim3(std::max(gx, 1u), (uint32_t)m->n_ranks), dim3(256), 0, c->stream, (const uint8_t *)m->recv.p, block, m->n_ranks, nq_max, knbn, all_ids_dev, all_dist_dev,
                       d_counts, d_bad);
    GS_HIP_CHECK(hipGetLastError());
    if (counts_dev) GS_HIP_CHECK(hipMemcpyAsync(counts_dev, d_counts, 8 * (size_t)m->n_ranks + 8, hipMemcpyDeviceToDevice, c->stream));
    return GS_OK;
}
/* The exchange WITHOUT the host round trip (round 6): pack -> ncclAllGather -> unpack are queued on the context's stream and the call returns; what follows on that
 * stream (the next step's sketch, a merge kernel) sees the gathered answers in order. counts_dev (DEVICE, optional): n_ranks counts + one word that is non-zero when
 * a rank sent a block of another shape. gs_comm_wait() is the synchronising half: it waits for the stream, checks that word and hands the counts to the host. */
int gs_comm_allgatherv_topk_async_dev(gs_comm *m, const uint64_t *ids_dev, const float *dist_dev, uint64_t nq_local, uint64_t nq_max, uint32_t knbn, uint64_t *all_ids_dev,
                                      float *all_dist_dev, uint64_t *counts_dev)
{
    GS_REQUIRE(m && all_ids_dev && all_dist_dev && nq_local <= nq_max && nq_max > 0 && knbn > 0 && (nq_local == 0 || (ids_dev && dist_dev)), GS_ERR_INVALID,
               "gs_comm_allgatherv_topk_async_dev: bad argument");
    std::lock_guard<std::recursive_mutex> lk(m->ctx->mu);
    return allgatherv_launch(m, ids_dev, dist_dev, nq_local, nq_max, knbn, all_ids_dev, all_dist_dev, counts_dev);
}
int gs_comm_wait(gs_comm *m, uint64_t *counts_out)
{
    GS_REQUIRE(m, GS_ERR_INVALID, "gs_comm_wait: null communicator");
    gs_ctx *c = m->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    GS_HIP_CHECK(hipSetDevice(c->device));
    if (!m->pending_counts) { GS_HIP_CHECK(hipStreamSynchronize(c->stream)); if (counts_out) for (int r = 0; r < m->n_ranks; r++) counts_out[r] = 0; return GS_OK; }
    std::vector<uint64_t> hc(m->n_ranks + 1, 0);
    GS_HIP_CHECK(hipMemcpyAsync(hc.data(), m->pending_counts, 8 * (size_t)m->n_ranks + 4, hipMemcpyDeviceToHost, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    GS_REQUIRE((uint32_t)hc[m->n_ranks] == 0, GS_ERR_INVALID, "gs_comm: a rank sent a block of another shape (nq_max / knbn must agree on every rank)");
    if (counts_out) for (int r = 0; r < m->n_ranks; r++) counts_out[r] = hc[r];
    return GS_OK;
}
int gs_comm_allgatherv_topk_dev(gs_comm *m, const uint64_t *ids_dev, const float *dist_dev, uint64_t nq_local, uint64_t nq_max, uint32_t knbn, uint64_t *all_ids_dev,
                                float *all_dist_dev, uint64_t *counts_out)
{
    GS_REQUIRE(m && all_ids_dev && all_dist_dev && nq_local <= nq_max && (nq_local == 0 || (ids_dev && dist_dev)), GS_ERR_INVALID, "gs_comm_allgatherv_topk_dev: bad argument");
    std::lock_guard<std::recursive_mutex> lk(m->ctx->mu);
    if (nq_max == 0 || knbn == 0) { if (counts_out) for (int r = 0; r < m->n_ranks; r++) counts_out[r] = 0; return GS_OK; }
    const int rc = allgatherv_launch(m, ids_dev, dist_dev, nq_local, nq_max, knbn, all_ids_dev, all_dist_dev, nullptr);
    return rc ? rc : gs_comm_wait(m, counts_out);
}
/* the equal-shape form: same nq_local and knbn on every rank; all_*_dev: n_ranks x nq_local x knbn, rank order */
int gs_comm_allgather_topk_dev(gs_comm *m, const uint64_t *ids_dev, const float *dist_dev, uint64_t nq_local, uint32_t knbn, uint64_t *all_ids_dev, float *all_dist_dev)
{
    GS_REQUIRE(m && ids_dev && dist_dev && all_ids_dev && all_dist_dev, GS_ERR_INVALID, "null argument");
    std::vector<uint64_t> counts(m->n_ranks);
    const int rc = gs_comm_allgatherv_topk_dev(m, ids_dev, dist_dev, nq_local, nq_local, knbn, all_ids_dev, all_dist_dev, counts.data());
    if (rc) return rc;
    for (int r = 0; r < m->n_ranks; r++) GS_REQUIRE(counts[r] == nq_local || nq_local == 0, GS_ERR_INVALID, "gs_comm_allgather_topk_dev: rank %d holds %llu queries, this rank %llu (use gs_comm_allgatherv_topk_dev for unequal shards)", r, (unsigned long long)counts[r], (unsigned long long)nq_local);
    return GS_OK;
}
/* DB-sharded alternative: ids_dev / dist_dev hold the answers of n_shards shards for the SAME nq queries, shard-major (n_shards x nq x knbn_in, what the
 * all-gather above returns when every rank answers all queries on its shard); id_offset (HOST, optional): added to the ids of shard s (local -> global
 * numbering). out_*: nq x knbn_out, the best under (distance, id), unused slots UINT64_MAX / +inf. No communicator needed. */
int gs_topk_merge_dev(gs_ctx *c, const uint64_t *ids_dev, const float *dist_dev, uint32_t n_shards, uint64_t nq, uint32_t knbn_in, const uint64_t *id_offset, uint32_t knbn_out,
                      uint64_t *out_ids_dev, float *out_dist_dev)
{
    GS_REQUIRE(c && ids_dev && dist_dev && out_ids_dev && out_dist_dev && n_shards >= 1 && knbn_in >= 1 && knbn_out >= 1, GS_ERR_INVALID, "gs_topk_merge_dev: bad argument");
    const uint64_t N = (uint64_t)n_shards * knbn_in;
    GS_REQUIRE(N * 12 <= 96 * 1024, GS_ERR_UNSUPPORTED, "gs_topk_merge_dev: %llu keys per query do not fit the LDS", (unsigned long long)N);
    if (nq == 0) return GS_OK;
    GS_CTX_LOCK(c);
    gs::PoolBuf off(c, 47);
    const uint64_t *d_off = nullptr;
    if (id_offset) {
        int rc = off.alloc(8 * (size_t)n_shards); if (rc) return rc;
        GS_HIP_CHECK(hipMemcpyAsync(off.p, id_offset, 8 * (size_t)n_shards, hipMemcpyHostToDevice, c->stream));
        d_off = off.as<uint64_t>();
    }
    auto kern = gs::k_topk_merge;
    const size_t lds = (size_t)N * 12;
    if (lds > 48 * 1024) GS_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((uint32_t)nq), dim3(gs::MG_T), lds, c->stream, ids_dev, dist_dev, n_shards, nq, knbn_in, d_off, knbn_out, out_ids_dev, out_dist_dev);
    GS_HIP_CHECK(hipGetLastError());
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

}  // extern "C"
