// gs_internal.hpp — shared plumbing of the C-ABI implementation (not part of the public interface).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include "../../include/gsearch_amd.h"

namespace gs {

void set_error(const char *fmt, ...);

#define GS_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            gs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return GS_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

#define GS_REQUIRE(cond, code, ...)                \
    do {                                           \
        if (!(cond)) {                             \
            gs::set_error(__VA_ARGS__);            \
            return (code);                         \
        }                                          \
    } while (0)

enum { FAM_SKETCH = 0, FAM_HAMMING = 1, FAM_SEARCH = 2, FAM_INSERT = 3, FAM_COUNT = 4 };

struct ProfSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    double total_ms = 0;
    uint64_t launches = 0;
};

}  // namespace gs

struct gs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 0;
    uint64_t hbm_bytes = 0;
    // this device against the one the cost model's rates were measured on (MI355X: 256 CUs at 2.4 GHz, 8 TB/s): dense_pays() scales its compute-side rates by
    // `rel_compute` and its streaming rates by `rel_hbm` instead of assuming that device (1.0 / 1.0 there; gs_ctx_create fills them from hipDeviceProp_t)
    double rel_compute = 1.0, rel_hbm = 1.0;
    char name[128] = {0};
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool profile = false;
    gs::ProfSlot prof[gs::FAM_COUNT];
    void *scratch_pool = nullptr;      // gs::ScratchPool: grow-only device buffers reused by the calls of this context
    bool wait_sleeping = false;        // gs::stream_wait sleeps (few long waits: the device pipeline of gs_sketch_files) instead of spinning (many short ones)
    hipEvent_t sync_ev = nullptr;      // blocking-sync event of gs::stream_wait (a sleeping wait: the file pipelines leave the cores to the decoders)
    gs_ctx *child = nullptr;           // second context (own stream, pools) on the same device: gs_sketch_files runs its host-decoded files on it
    void *pinned_pool = nullptr;       // gs::PinnedPool: grow-only pinned host staging buffers of gs_sketch_files (hipHostMalloc costs ~0.3 s per GB)
    // Worker contexts (gs::worker_ctx): the reference clones its sketcher into --nbthreads workers that all call it through &self
    // (dnasketch.rs:252,305,322). The synchronous host-pointer entry points (gs_sketch_batch, gs_hamming_qxc, gs_hamming_pairs) give every calling
    // thread other than the first its own stream + scratch on the same device, so such calls run side by side instead of queueing on one lock.
    std::thread::id owner_thread{};
    std::map<std::thread::id, gs_ctx *> workers; std::mutex workers_mu;
    gs_ctx *parent = nullptr;
    uint64_t last_use = 0, use_tick = 0;   // LRU stamps of the worker table (under workers_mu)
    // A worker is PINNED (under the parent's workers_mu) from the moment worker_ctx() hands it out, or a snapshot of the table copies its pointer, until
    // worker_unpin(): eviction only takes workers with pins == 0, and a worker whose thread exits while a snapshot still holds it is only marked
    // `orphan` - the last unpin destroys it (the LRU eviction could free a worker between worker_ctx() and the body's lock).
    int pins = 0; bool orphan = false;
    uint32_t sketch_min_lds = 0;             // the slot-min sketch kernel asks for at least this much LDS per workgroup (co-residency shaping of the request pipeline; 0 = what it needs)
    uint32_t last_sketch[4] = {0, 0, 0, 0};   // gs_ctx_last_sketch_info: {filtered emitter, slot table in LDS, workgroups per genome, launches} of the last slot-min sketch call
    // One context = one stream and one scratch pool. The reference clones its sketcher into --nbthreads workers and calls it, DistHamming
    // and parallel_search through &self from many threads (dnasketch.rs:252,305,322): every entry point that touches the stream or the
    // pool takes this lock, so concurrent calls on one context are safe (they queue on the GPU anyway); recursive because entry
    // points call one another (save -> export / get_data, bruteforce -> hamming).
    std::recursive_mutex mu;
};
// Taking the lock also binds the calling thread to the context's device: a worker thread's current device is 0 until it says otherwise, and
// allocations / launches made from it must land on the device that owns the stream.
#define GS_CTX_LOCK(c) std::lock_guard<std::recursive_mutex> gs_ctx_lock_((c)->mu); (void)hipSetDevice((c)->device)

namespace gs {

// the context a host-pointer call of the current thread runs on: `c` itself for the thread that used it first, a per-thread worker context otherwise
// (GS_THREAD_CONTEXTS=0: always `c` - every call of every thread queues on the one context, as before round 4)
gs_ctx *worker_ctx(gs_ctx *c);
void worker_done(gs_ctx *parent, gs_ctx *worker);
void worker_unpin(gs_ctx *parent, gs_ctx *worker);
void on_worker_failed(gs_ctx *parent);
// run a synchronous host-pointer call on the calling thread's worker context; a device failure there (a worker's scratch pool beside 63 others)
// releases the idle workers' pools and repeats the call on the parent, where it queues as every call did before worker contexts existed
template <class F> inline int on_worker(gs_ctx *c, F &&body)
{
    gs_ctx *w = worker_ctx(c);
    int rc = body(w);
    if (w != c) {
        worker_done(c, w);
        if (rc == GS_ERR_HIP) { on_worker_failed(c); rc = body(c); }
    }
    return rc;
}

// Wait for the context's stream WITHOUT spinning: hipStreamSynchronize busy-waits on a core, and the two pipelines of gs_sketch_files wait
// for hundreds of milliseconds at a time (a k_inflate launch) while the host decoders want every core of the cgroup's quota.
inline hipError_t stream_wait(gs_ctx *c)
{
    hipError_t e;
    // a sleeping waiter has to be scheduled again when the event fires: with every core of the quota busy decoding that takes milliseconds, and
    // the host pipeline waits a few times per 64-file group (16 ms per group instead of 1.7) - it spins
    if (!c->wait_sleeping) return hipStreamSynchronize(c->stream);
    if (!c->sync_ev && (e = hipEventCreateWithFlags(&c->sync_ev, hipEventBlockingSync | hipEventDisableTiming)) != hipSuccess) return e;
    if ((e = hipEventRecord(c->sync_ev, c->stream)) != hipSuccess) return e;
    return hipEventSynchronize(c->sync_ev);
}

// RAII-ish helpers -----------------------------------------------------------------------------
struct ProfScope {   // brackets one kernel launch with events when profiling is on
    gs_ctx *c; int fam; hipEvent_t a = nullptr, b = nullptr;
    ProfScope(gs_ctx *ctx, int family) : c(ctx), fam(family)
    {
        if (c->profile) {
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a, c->stream);
        }
    }
    ~ProfScope()
    {
        if (c->profile) {
            (void)hipEventRecord(b, c->stream);
            c->prof[fam].pending.emplace_back(a, b);
        }
    }
};

// dense Q x C DistHamming tile kernel with arbitrary row strides (bytes); writes exactly one of: f32 distances (out),
// 32-bit mismatch counts (out_cnt), 16-bit mismatch counts (out_cnt16); ld_out = output row pitch in elements (0 -> nc).
int hamming_qxc_strided(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t nq, uint64_t strideQ_bytes, const void *C, uint64_t nc,
                        uint64_t strideC_bytes, float *out, uint32_t *out_cnt, uint16_t *out_cnt16, uint64_t ld_out);

int widen_u16_rows(gs_ctx *c, const void *src_dev, uint64_t nrows, uint32_t m, void *dst_dev, uint64_t dst_stride_bytes);
int narrow_u16_rows(gs_ctx *c, const void *src_dev, uint64_t src_stride_bytes, uint64_t nrows, uint32_t m, void *dst_dev);

struct DevBuf;
// match-join form of the dense count matrix (gs_join.hip): counts of nq strided query rows against the first n nodes of the
// column-major database copy `cols` ([m][colcap]); out16[q * ld + e] = mismatch count. scratch: JOIN_SCRATCH reusable buffers. init = false: the
// counters already hold m (a sub-range of columns of rows initialised by an earlier call: the row-spanning memset would wipe their neighbours); col0: node e of the range is column col0 + e of the matrix.
uint64_t match_join_max_queries();
enum { JOIN_SCRATCH = 16 };
// rows / rstride (optional): the row-major signatures of the same nodes - with them a request batch's heavy (query, node) blocks are found and
// written by the compare tile kernel instead of being counted match by match (gs_join.hip "heavy blocks")
int match_join_counts(gs_ctx *c, int kind, uint32_t m, const void *qrows, uint64_t qstride, uint64_t nq, const void *cols, uint64_t colcap, uint64_t n,
                      uint16_t *out16, uint64_t ld, DevBuf *scratch, int *declined = nullptr, unsigned long long *stats = nullptr, bool init = true, uint64_t col0 = 0,
                      const void *rows = nullptr, uint64_t rstride = 0);
int hamming_blocks(gs_ctx *c, int kind, uint32_t m, const void *Q, uint64_t strideQ_bytes, const void *C, uint64_t strideC_bytes, const void *items_dev,
                   uint32_t n_items, const uint32_t *qlist_dev, const uint32_t *clist_dev, uint16_t *out_cnt16, uint64_t ld_out, uint32_t n_thin = 0);
int rows_to_cols(gs_ctx *c, int kind, uint32_t m, const void *rows, uint64_t stride, uint64_t nrows, void *cols, uint64_t colcap, uint64_t first);

// gs_sketch.hip: the device sketch of a batch on context c's stream; sync_at_end = false leaves the results in flight (optdens / revoptdens only)
int sketch_dev_impl(gs_ctx *c, const gs_sketch_params *p, const void *seq, uint64_t seq_bytes, const uint64_t *rec_start, const uint64_t *rec_len, uint64_t n_rec,
                    const uint64_t *genome_rec_off, uint64_t n_genomes, void *sig_out, bool sync_at_end);
// gs_radix.hip: stable LSD radix sort of 64-bit keys (bits [0, endbit)) between two buffers, and run-length encoding of a sorted array
size_t radix_scratch_bytes(uint64_t n);
int radix_sort_u64(gs_ctx *c, uint64_t *keys, uint64_t *alt, uint64_t n, int endbit, void *scratch, uint64_t **sorted_out);
int run_length_encode_u64(gs_ctx *c, const uint64_t *sorted, uint64_t n, uint64_t *uniq, uint32_t *len, uint32_t *nruns_dev, uint32_t *pos, void *scratch);

inline size_t kind_bytes(int kind) { return kind == GS_KIND_U16 ? 2 : (kind == GS_KIND_U64 ? 8 : 4); }
inline uint64_t round_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

inline bool mem_verbose() { static const bool v = getenv("GS_MEM_VERBOSE") != nullptr; return v; }   // trace of the allocations of 256 MB and more (stderr)
struct DevBuf {   // owning device allocation
    void *p = nullptr; size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    int alloc(size_t n)
    {
        release();
        if (n == 0) n = 16;
        const hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess || mem_verbose()) {
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            if (e != hipSuccess) {     // say how much was asked for and what the device had left: "out of memory" alone does not tell a 54 GB block from a 1 MB one
                p = nullptr;
                gs::set_error("hipMalloc of %zu bytes failed: %s (%zu of %zu bytes free on the device)", n, hipGetErrorString(e), fr, tot);
                return GS_ERR_HIP;
            }
            if (n >= ((size_t)256 << 20)) fprintf(stderr, "[GS_MEM] + %.2f GB (free after %.2f of %.2f GB)\n", n / 1e9, fr / 1e9, tot / 1e9);
        }
        bytes = n;
        return GS_OK;
    }
    int ensure(size_t n) { return (n <= bytes && p) ? GS_OK : alloc(n); }
    template <class T> T *as() const { return (T *)p; }
};

// A device buffer that GROWS IN PLACE (round 5): one reserved range of virtual addresses, physical memory mapped into it 1 GB at a time
// (hipMemAddressReserve / hipMemCreate / hipMemMap; the chunk size is a parameter so that tests can fill an arena of a few MB). What is already there keeps its address and its content, growing costs neither a second
// copy next to the first nor a memcpy - the sparse pair rows of an index that is still being built live here (gs_index.hip), addressed by
// 32-bit offsets from base(). Chunks of more than 2 GB are refused by hipMemSetAccess on this runtime (tools/ubench_vmm.hip).
struct VmArena {
    size_t chunk = (size_t)1 << 30;
    void *va = nullptr; size_t va_bytes = 0, mapped = 0; int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    VmArena() = default;
    VmArena(const VmArena &) = delete;
    VmArena &operator=(const VmArena &) = delete;
    ~VmArena() { release(); }
    static bool supported(int dev)
    {
        int sup = 0;
        return hipDeviceGetAttribute(&sup, hipDeviceAttributeVirtualMemoryManagementSupported, dev) == hipSuccess && sup != 0;
    }
    // address space only (no memory yet); bytes is rounded up to whole chunks. false: not available
    bool reserve(int dev, size_t bytes, size_t chunk_bytes = (size_t)1 << 30)
    {
        release();
        if (!supported(dev)) return false;
        chunk = std::max<size_t>(chunk_bytes / 65536 * 65536, 65536);
        bytes = (bytes + chunk - 1) / chunk * chunk;
        if (hipMemAddressReserve(&va, bytes, 0, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); va = nullptr; return false; }
        va_bytes = bytes; device = dev; mapped = 0;
        return true;
    }
    // at least `bytes` of the range backed by memory. false: the range or the device is full (what is mapped stays)
    bool map_to(size_t bytes)
    {
        if (bytes > va_bytes) return false;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice; acc.location.id = device; acc.flags = hipMemAccessFlagsProtReadWrite;
        while (mapped < bytes) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
            if (hipMemMap((char *)va + mapped, chunk, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); return false; }
            if (hipMemSetAccess((char *)va + mapped, chunk, &acc, 1) != hipSuccess) { (void)hipGetLastError(); (void)hipMemUnmap((char *)va + mapped, chunk); (void)hipMemRelease(h); return false; }
            handles.push_back(h); mapped += chunk;
            if (mem_verbose()) fprintf(stderr, "[GS_MEM] arena: %.2f GB mapped of %.2f GB reserved\n", mapped / 1e9, va_bytes / 1e9);
        }
        return true;
    }
    void release()
    {
        for (size_t i = 0; i < handles.size(); i++) { (void)hipMemUnmap((char *)va + i * chunk, chunk); (void)hipMemRelease(handles[i]); }
        handles.clear();
        if (va) (void)hipMemAddressFree(va, va_bytes);
        va = nullptr; va_bytes = 0; mapped = 0;
    }
};

// Per-context scratch: a call's temporaries come from numbered grow-only slots instead of hipMalloc/hipFree (allocating and freeing
// multi-GB buffers costs more than the kernels that use them). A context serves one call at a time (one stream), so slots are never
// shared; gs_ctx_release_scratch / gs_ctx_destroy give the memory back.
enum { SCRATCH_SLOTS = 64 };        // 48-52: staging of gs_sketch_batch, 53-57: of gs_hamming_qxc / gs_hamming_pairs (host-pointer calls)
struct ScratchPool { DevBuf b[SCRATCH_SLOTS]; };
enum { PINNED_SLOTS = 36 };      // 0-15 text, 16-31 compressed members, 32-33 inflate descriptors / results
struct PinnedPool {
    void *p[PINNED_SLOTS] = {}; size_t cap[PINNED_SLOTS] = {};
    ~PinnedPool() { for (int i = 0; i < PINNED_SLOTS; i++) if (p[i]) (void)hipHostFree(p[i]); }
    // at least n bytes in slot i; the previous content is NOT kept. nullptr when the allocation fails
    void *ensure(int i, size_t n)
    {
        if (p[i] && cap[i] >= n) return p[i];
        if (p[i]) (void)hipHostFree(p[i]);
        p[i] = nullptr; cap[i] = 0;
        if (hipHostMalloc(&p[i], n, hipHostMallocDefault) != hipSuccess) { p[i] = nullptr; return nullptr; }
        cap[i] = n;
        return p[i];
    }
};
inline PinnedPool *pinned_pool(gs_ctx *c)
{
    if (!c->pinned_pool) c->pinned_pool = new PinnedPool();
    return (PinnedPool *)c->pinned_pool;
}
struct PoolBuf {    // same surface as DevBuf for the code that uses it
    gs_ctx *c; int slot; void *p = nullptr; size_t bytes = 0;
    PoolBuf(gs_ctx *ctx, int s) : c(ctx), slot(s) {}
    int alloc(size_t n)
    {
        if (!c->scratch_pool) c->scratch_pool = new ScratchPool();
        DevBuf &d = ((ScratchPool *)c->scratch_pool)->b[slot];
        if (n == 0) n = 16;
        int rc = d.ensure(n);
        if (rc) return rc;
        p = d.p; bytes = n;
        return GS_OK;
    }
    void release() { p = nullptr; bytes = 0; }          // the slot keeps its memory for the next call
    template <class T> T *as() const { return (T *)p; }
};

}  // namespace gs
