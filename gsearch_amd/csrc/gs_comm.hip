// gs_comm.hip — multi-GPU exchange of the path behind the C ABI, for hosts without torch (the reference's host is Rust):
// one process per GPU, one RCCL communicator over xGMI, and the only collective the path has - the all-gather of the per-rank
// top-k blocks (conceptual ancestor: the per-shard loop + merge of /root/reference/scripts/multiple_search.sh:71-107).
// RCCL is loaded lazily (dlopen) when the first communicator is made: the sketch / distance / index entry points do not depend on
// it, and a process that already carries an RCCL (e.g. through torch.distributed) is not disturbed unless it asks for this one.
#include <dlfcn.h>
#include <mutex>
#include <string.h>
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
    for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    GS_REQUIRE(h, GS_ERR_UNSUPPORTED, "RCCL (librccl.so) cannot be loaded: %s", dlerror());
    Rccl r; r.h = h;
    r.get_id = (fn_get_id)dlsym(h, "ncclGetUniqueId"); r.init_rank = (fn_init_rank)dlsym(h, "ncclCommInitRank");
    r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy"); r.allgather = (fn_allgather)dlsym(h, "ncclAllGather");
    r.errstr = (fn_errstr)dlsym(h, "ncclGetErrorString");
    GS_REQUIRE(r.get_id && r.init_rank && r.destroy && r.allgather, GS_ERR_UNSUPPORTED, "librccl.so lacks an expected entry point");
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

/* Every rank contributes nq_local x knbn neighbour ids (u64) and distances (f32) - identical shapes on all ranks - and receives the
 * concatenation in rank order: ONE ncclAllGather of the packed per-rank block (12 bytes per neighbour; latency bound).
 * All pointers are device memory of the communicator's context; asynchronous on the context's stream until the final sync. */
int gs_comm_allgather_topk_dev(gs_comm *m, const uint64_t *ids_dev, const float *dist_dev, uint64_t nq_local, uint32_t knbn, uint64_t *all_ids_dev, float *all_dist_dev)
{
    GS_REQUIRE(m && ids_dev && dist_dev && all_ids_dev && all_dist_dev, GS_ERR_INVALID, "null argument");
    gs_ctx *c = m->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    GS_HIP_CHECK(hipSetDevice(c->device));
    const size_t nb_ids = (size_t)nq_local * knbn * 8, nb_dist = (size_t)nq_local * knbn * 4, block = nb_ids + nb_dist;
    if (block == 0) return GS_OK;
    int rc;
    if ((rc = m->send.ensure(block))) return rc;
    if ((rc = m->recv.ensure(block * (size_t)m->n_ranks))) return rc;
    GS_HIP_CHECK(hipMemcpyAsync(m->send.p, ids_dev, nb_ids, hipMemcpyDeviceToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpyAsync((uint8_t *)m->send.p + nb_ids, dist_dev, nb_dist, hipMemcpyDeviceToDevice, c->stream));
    GS_NCCL_CHECK(gs::g_rccl.allgather(m->send.p, m->recv.p, block, /*ncclChar*/ 0, m->comm, c->stream));
    GS_HIP_CHECK(hipMemcpy2DAsync(all_ids_dev, nb_ids, m->recv.p, block, nb_ids, (size_t)m->n_ranks, hipMemcpyDeviceToDevice, c->stream));
    GS_HIP_CHECK(hipMemcpy2DAsync(all_dist_dev, nb_dist, (uint8_t *)m->recv.p + nb_ids, block, nb_dist, (size_t)m->n_ranks, hipMemcpyDeviceToDevice, c->stream));
    GS_HIP_CHECK(hipStreamSynchronize(c->stream));
    return GS_OK;
}

}  // extern "C"
